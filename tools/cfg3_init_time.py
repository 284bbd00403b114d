#!/usr/bin/env python3
"""BASELINE config 3's initialisation alone (8192 x 8192 noise, k = 256, full resolution): wall time of kmg_lloyd_init_centroids,
warm (the second and third call), with KMG_LOG=debug the number of launches.   python tools/cfg3_init_time.py [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 8192 * 8192
st = torch.cuda.current_stream().cuda_stream
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
p = kg.ImageProcessor(shrink_max_dim=0)
s = kg.Lloyd(p, k)
for i in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    s.init_centroids(rgba.data_ptr(), 8192, 8192, st)
    torch.cuda.synchronize()
    print(f"init k={k} call {i}: {(time.perf_counter() - t) * 1e3:.3f} ms", flush=True)
