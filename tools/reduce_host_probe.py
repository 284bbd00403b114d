#!/usr/bin/env python3
"""kmg_reduce of the 8192 x 8192 image from and to host buffers, k = 256, dither: wall time per call (run under
`rocprofv3 --hip-trace --kernel-trace --stats` to see where the host time goes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
W = 8192
img = synth.uniform_rgba(synth.SEED_CFG3, W * W).reshape(W, W, 4) if hasattr(synth, "uniform_rgba") else None
if img is None:
    import torch
    img = synth.uniform_rgba_torch(synth.SEED_CFG3, W * W, device="cuda").cpu().numpy().reshape(W, W, 4)
p = kg.ImageProcessor(shrink_max_dim=0)
for i in range(4):
    t = time.perf_counter()
    out = p.reduce(256, img, reduce_mode=1)
    print(f"call {i}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
