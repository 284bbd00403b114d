#!/usr/bin/env python3
"""Time of binding an 8192x8192 image (histogram build + cell sums + work list): noise, a tiled photograph,
a flat image -- the partitioned build must not degrade on skewed colour distributions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
W = 8192; n = W * W
st = torch.cuda.current_stream().cuda_stream
tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
big = np.tile(tokyo, (16, 11, 1))[:W, :W].copy()
imgs = {"noise": synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda"),
        "photo": torch.from_numpy(big.reshape(-1, 4)).cuda(),
        "flat": torch.full((n, 4), 77, dtype=torch.uint8, device="cuda"),
        "two colours": torch.from_numpy(np.where(np.arange(n)[:, None] % 3 == 0, np.uint8(10), np.uint8(200)).repeat(4, 1).astype(np.uint8)).cuda()}
p = kg.ImageProcessor(shrink_max_dim=0)
s = kg.Lloyd(p, 16)
for name, img in imgs.items():
    s.bind_image(img.data_ptr(), n, st); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): s.bind_image(img.data_ptr(), n, st)
    torch.cuda.synchronize()
    print(f"{name:12s} bind {(time.perf_counter() - t) / 5 * 1e3:.3f} ms")
