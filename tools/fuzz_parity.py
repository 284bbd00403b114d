#!/usr/bin/env python3
"""Randomised cross-check of the two strategy families through the C ABI: for random images, sizes, k and
centroid tables, the colour-table / pruned paths (kmg_options.strategy = table) must return exactly what the per-pixel
scans (strategy = scan) return -- initialisation, Lloyd run (labels, centroids, iteration count) and the
three output modes.  usage: fuzz_parity.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.current_stream().cuda_stream


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
    else:  # gradient
        i = np.arange(n)
        a = np.stack([(i % w) * 255 // max(w - 1, 1), (i // w) * 255 // max(h - 1, 1), (i * 7) % 256, i % 256], 1).astype(np.uint8)
    # alpha is ignored on input (rgb_to_lab.wgsl:78) and 255 on output (lab_to_rgb.wgsl:37): every image carries random alpha
    a[:, 3] = rng.integers(0, 256, n, dtype=np.uint8)
    return a


def run(strategy, rgba, w, h, k, cent_fixed):
    kg.set_strategy(strategy)
    n = w * h
    p = kg.ImageProcessor(shrink_max_dim=0, max_iterations=12)
    d = torch.from_numpy(rgba).cuda()
    s = kg.Lloyd(p, k)
    s.init_centroids(d.data_ptr(), w, h, st)
    c_init = s.get_centroids(st).copy()
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    it = s.run(d.data_ptr(), n, labels.data_ptr(), st)
    c_run = s.get_centroids(st).copy()
    outs = []
    for mode in (kg.ReduceMode.Replace, kg.ReduceMode.Dither, kg.ReduceMode.Meld):
        out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent_fixed, mode, out.data_ptr(), st)
        outs.append(out.cpu().numpy())
    torch.cuda.synchronize()
    res = (c_init.view(np.uint32), it, c_run.view(np.uint32), labels.cpu().numpy(), *outs)
    s.close(); p.close()
    return res


bad = 0
for case in range(cases):
    kind = ["noise", "few", "blobs", "gradient"][int(rng.integers(0, 4))]
    w = int(rng.integers(1, 1500)); h = int(rng.integers(1, 1500)) if rng.random() < 0.8 else 1
    if rng.random() < 0.1:
        w, h = int(rng.integers(1500, 2600)), int(rng.integers(900, 1200))     # beyond 2^21 pixels: partitioned histogram
    k = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 65, 130, 256, 257, 300]))
    rgba = image(kind, w, h)
    pal = rng.integers(0, 256, (k, 4), dtype=np.uint8); pal[:, 3] = 255
    cent_fixed = kg.palette_to_centroids(pal)
    a = run("brute", rgba, w, h, k, cent_fixed)
    b = run("table", rgba, w, h, k, cent_fixed)
    names = ["init centroids", "iterations", "final centroids", "labels", "replace", "dither", "meld"]
    diff = [nm for nm, x, y in zip(names, a, b) if not np.array_equal(x, y)]
    if diff:
        bad += 1
        print(f"MISMATCH case {case}: {kind} {w}x{h} k={k}: {diff}", flush=True)
print(f"{cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
