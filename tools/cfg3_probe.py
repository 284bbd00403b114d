#!/usr/bin/env python3
"""BASELINE config 3 end to end on one GPU: reference init at full resolution, Lloyd to convergence
(or the iteration cap), label map, dither output pass -- wall time of each stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
W = 8192; n = W * W
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
p = kg.ImageProcessor(shrink_max_dim=0)
def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t) * 1e3:9.2f} ms"); return r
for rep in range(2):
    s = kg.Lloyd(p, k)
    timed("init (bind + k-1 passes)", lambda: s.init_centroids(rgba.data_ptr(), W, W, st))
    timed("init again (passes only)", lambda: s.init_centroids(rgba.data_ptr(), W, W, st))
    it = timed("lloyd run + label map", lambda: s.run(rgba.data_ptr(), n, labels.data_ptr(), st))
    cent = s.get_centroids(st)
    timed("dither output pass", lambda: p.apply(rgba.data_ptr(), W, W, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st))
    print("iterations", it)
    s.close()
