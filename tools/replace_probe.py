#!/usr/bin/env python3
"""find -m replace with the 64-entry palette on 8192x8192 through the colour table (kernel times: run under rocprofv3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
n = 8192 * 8192
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal = np.array(sorted(set(map(tuple, px))), np.uint8)
cent = kg.palette_to_centroids(pal)
proc = kg.ImageProcessor()
rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode in (kg.ReduceMode.Replace, kg.ReduceMode.Dither):
    for _ in range(4):
        proc.apply(rgba.data_ptr(), 8192, 8192, 0, cent, mode, out.data_ptr(), st)
torch.cuda.synchronize()
