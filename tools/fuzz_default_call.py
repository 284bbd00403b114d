#!/usr/bin/env python3
"""Randomised check of the reference's default call against the CPU oracle (tests/oracle_lib.py): kmg_reduce and kmg_palette of
random images (noise / few colours / blobs / gradients, 1 x 1 ... ~900 x 700, so the shrink to <= 256 is exercised both ways)
for random k and modes must equal oracle.reduce / oracle.palette byte for byte.   usage: fuzz_default_call.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kmeans_gpu_amd as kg
import oracle_lib as oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
p = kg.ImageProcessor()


def image(kind, w, h):
    n = w * h
    if kind == "noise":
        a = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    elif kind == "few":
        pal = rng.integers(0, 256, (int(rng.integers(1, 9)), 4), dtype=np.uint8)
        a = pal[rng.integers(0, pal.shape[0], n)]
    elif kind == "blobs":
        c = rng.integers(0, 256, (int(rng.integers(2, 30)), 3))
        a = np.zeros((n, 4), np.uint8)
        a[:, :3] = np.clip(c[rng.integers(0, c.shape[0], n)] + rng.normal(0, rng.uniform(2, 30), (n, 3)), 0, 255).astype(np.uint8)
    else:
        i = np.arange(n)
        a = np.stack([(i % w) * 255 // max(w - 1, 1), (i // w) * 255 // max(h - 1, 1), (i * 7) % 256, i % 256], 1).astype(np.uint8)
    # alpha is ignored on input (rgb_to_lab.wgsl:78) and 255 on output (lab_to_rgb.wgsl:37): every image carries random alpha
    a[:, 3] = rng.integers(0, 256, n, dtype=np.uint8)
    return a.reshape(h, w, 4)


bad = 0
for case in range(cases):
    kind = ["noise", "few", "blobs", "gradient"][int(rng.integers(0, 4))]
    w = int(rng.integers(1, 900)); h = int(rng.integers(1, 700)) if rng.random() < 0.85 else 1
    k = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 32, 33, 64, 100, 256, 300]))
    mode = int(rng.integers(0, 3))
    img = image(kind, w, h)
    got = p.reduce(k, img, reduce_mode=mode)
    want = oracle.reduce(img, k, mode)
    ok = np.array_equal(got, want)
    gp, wp = p.palette(k, img), oracle.palette(img, k)
    ok = ok and np.array_equal(gp, wp)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {kind} {w}x{h} k={k} mode={mode}", flush=True)
print(f"{cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
