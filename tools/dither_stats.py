#!/usr/bin/env python3
"""Candidate counts of the dither output pass per (cell, Bayer index) slot and per cell (KMG_DITHER_STATS makes
kmg_debug_check_dither_masks print both distributions to stderr); run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
use_tools_library()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
os.environ["KMG_DITHER_STATS"] = "1"
import numpy as np, torch
import kmeans_gpu_amd as kg
from PIL import Image
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal = np.array(sorted(set(map(tuple, px))), np.uint8)
proc = kg.ImageProcessor(shrink_max_dim=0)
print("resurrect64", proc.debug_check_dither_masks(kg.palette_to_centroids(pal)))
rng = np.random.default_rng(3)
for k in (16, 256):
    p = rng.integers(0, 256, (k, 4), dtype=np.uint8); p[:, 3] = 255
    print("random", k, proc.debug_check_dither_masks(kg.palette_to_centroids(p)))
