#!/usr/bin/env python3
"""Farthest-point initialisation: per-pixel passes (k >= 32: several centroids per launch) against the passes over the image's
colours, for a grid of (pixels, k) on noise -- the data behind init_table_pays() in csrc/kmg_lloyd.hip.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
st = torch.cuda.current_stream().cuda_stream
p = kg.ImageProcessor(shrink_max_dim=0)
print("pixels      k   pixels-path ms   colours-path ms   auto")
for mpx in (0.25, 0.5, 1, 2, 4, 8):
    n = int(mpx * (1 << 20))
    w = 1024
    h = n // w
    rgba = synth.uniform_rgba_torch(0x1717, n, device="cuda")
    for k in (16, 64, 256):
        res = {}
        for name, env in (("pixels", "brute"), ("colours", "table"), ("auto", None)):
            if env is None:
                kg.set_strategy("auto")
            else:
                kg.set_strategy(env)
            s = kg.Lloyd(p, k)
            s.init_centroids(rgba.data_ptr(), w, h, st)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                s.init_centroids(rgba.data_ptr(), w, h, st)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t)
            res[name] = best * 1e3
            s.close()
        auto = "pixels" if abs(res["auto"] - res["pixels"]) < abs(res["auto"] - res["colours"]) else "colours"
        print(f"{n:9d} {k:4d}   {res['pixels']:10.3f}   {res['colours']:12.3f}      {auto}{'  <-- slower choice' if res[auto] > 1.15 * min(res['pixels'], res['colours']) else ''}", flush=True)
