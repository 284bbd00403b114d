#!/usr/bin/env python3
"""Randomised check of the multi-device layer of the C ABI (kmg_group_*): for random images, k, modes, algorithms and device
lists -- one rank with every RCCL collective forced, two to five ranks sharing device 0 through the loopback exchange -- and for
both working resolutions (the reference's shrink to 256, and full resolution = the sharded k-means), palette / find / reduce
of the group must equal the single processor's, byte for byte.   usage: fuzz_group.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np
import kmeans_gpu_amd as kg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
procs, groups = {}, {}


def processor(shrink):
    if shrink not in procs:
        procs[shrink] = kg.ImageProcessor(shrink_max_dim=shrink)
    return procs[shrink]


def group(ranks, shrink):
    key = (ranks, shrink)
    if key not in groups:
        flags = kg.GROUP_FORCE_COLLECTIVES if ranks == 1 else kg.GROUP_LOOPBACK
        groups[key] = kg.Group(devices=[0] * ranks, flags=flags, shrink_max_dim=shrink)
    return groups[key]


def image(kind, w, h):
    n = w * h
    if kind == "noise":
        a = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    elif kind == "few":
        pal = rng.integers(0, 256, (int(rng.integers(1, 9)), 4), dtype=np.uint8)
        a = pal[rng.integers(0, pal.shape[0], n)]
    else:
        i = np.arange(n)
        a = np.stack([(i % w) * 255 // max(w - 1, 1), (i // w) * 255 // max(h - 1, 1), (i * 7) % 256, i % 256], 1).astype(np.uint8)
    return np.ascontiguousarray(a.reshape(h, w, 4))


bad = 0
for case in range(cases):
    kind = ["noise", "few", "gradient"][int(rng.integers(0, 3))]
    ranks = int(rng.choice([1, 2, 3, 4, 5]))
    shrink = int(rng.choice([256, 0]))
    if shrink == 0 and rng.random() < 0.5:
        w, h = int(rng.integers(1024, 1400)), int(rng.integers(1024, 1100))      # >= 2^20 pixels: the k-means itself is sharded
    else:
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500)) if rng.random() < 0.9 else 1
    k = int(rng.choice([1, 2, 3, 8, 16, 33, 64]))
    mode = int(rng.integers(0, 3))
    algo = int(rng.random() < 0.25)
    img = image(kind, w, h)
    p, g = processor(shrink), group(ranks, shrink)
    pal = rng.integers(0, 256, (int(rng.integers(1, 40)), 4), dtype=np.uint8)
    ok = np.array_equal(g.reduce(k, img, algo, mode), p.reduce(k, img, algo, mode))
    ok = ok and np.array_equal(g.palette(k, img, algo), p.palette(k, img, algo))
    ok = ok and np.array_equal(g.find(img, pal, mode), p.find(img, pal, mode))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {kind} {w}x{h} k={k} mode={mode} algo={algo} ranks={ranks} shrink={shrink}", flush=True)
for g in groups.values():
    g.close()
print(f"{cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
