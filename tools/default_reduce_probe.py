#!/usr/bin/env python3
"""The reference's DEFAULT call (lib.rs:116-164: shrink to <= 256, farthest-point init, Lloyd loop, output pass) from and to
host buffers: wall time per call of kmg_reduce / kmg_palette / kmg_find on the reference's test image (tests/golden/tokyo.png)
and on a 3840 x 2160 tile of it, for a few k -- the latency a user of the CLI sees.  KMG_LOG=debug prints the stage times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np
from PIL import Image
import kmeans_gpu_amd as kg

tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
big = np.tile(tokyo, (2160 // tokyo.shape[0] + 1, 3840 // tokyo.shape[1] + 1, 1))[:2160, :3840].copy()
p = kg.ImageProcessor()                       # defaults: shrink_max_dim = 256, 128 iterations, check every 8


def best(fn, reps=6):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts[1:]), ts[0]


for name, img in (("tokyo %dx%d" % tokyo.shape[1::-1], tokyo), ("4K tile 3840x2160", big)):
    for k in (8, 64, 256):
        for mode, mname in ((0, "replace"), (1, "dither")):
            warm, cold = best(lambda: p.reduce(k, img, reduce_mode=mode))
            print(f"{name:22s} reduce  k={k:3d} {mname:8s} {warm:8.2f} ms (first call {cold:8.2f})", flush=True)
        warm, cold = best(lambda: p.palette(k, img))
        print(f"{name:22s} palette k={k:3d}          {warm:8.2f} ms (first call {cold:8.2f})", flush=True)
    pal = p.palette(64, img)
    warm, cold = best(lambda: p.find(img, pal, reduce_mode=1))
    print(f"{name:22s} find    k= 64 dither   {warm:8.2f} ms (first call {cold:8.2f})", flush=True)
