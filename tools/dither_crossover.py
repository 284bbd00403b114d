#!/usr/bin/env python3
"""Ordered-dither output pass: scan of all centroids (k_apply) against the pruned pass (lists over Lab cells for k <= 256, mask
words above) over a grid of (pixels, k) on noise -- the data behind dither_pruning_pays() in csrc/kmg_apply.hip.
python tools/dither_crossover.py > gpurun_out/dither_crossover.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
W = 8192
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, W * W, device="cuda")
out = torch.empty((W * W, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)
for k in (4, 8, 12, 16, 24, 32, 64, 128, 256, 512):
    pal = rng.integers(0, 256, (k, 4), dtype=np.uint8); pal[:, 3] = 255
    cent = kg.palette_to_centroids(pal)
    for side in (128, 256, 512, 1024, 2048, 4096, 8192):
        res = {}
        for strat in ("brute", "table"):
            kg.set_strategy(strat)
            proc = kg.ImageProcessor(shrink_max_dim=0)
            reps = 3 if side >= 2048 else 20
            proc.apply(rgba.data_ptr(), side, side, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps):
                proc.apply(rgba.data_ptr(), side, side, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
            torch.cuda.synchronize()
            res[strat] = ((time.perf_counter() - t) / reps * 1e3, out[: side * side].clone())
            proc.close()
        same = bool(torch.equal(res["brute"][1], res["table"][1]))
        print(f"k={k:4d} {side:5d}^2: scan {res['brute'][0]:8.3f} ms, pruned {res['table'][0]:8.3f} ms, identical {same}", flush=True)
