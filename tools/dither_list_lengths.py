#!/usr/bin/env python3
"""cfg5 (find -m dither, 8192^2, resurrect_64): how long are the lanes' candidate lists against the wave's longest -- the share of
k_dither_lists' list walk that is padding (tools build: kmg_tools_dither_list_stats).   python tools/dither_list_lengths.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
use_tools_library()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
import bench
from PIL import Image
proc = kg.ImageProcessor(shrink_max_dim=0)
L = kg.lib()
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal = np.array(sorted(set(map(tuple, px))), np.uint8)
rng = np.random.default_rng(5)
pal256 = rng.integers(0, 256, (256, 4), dtype=np.uint8); pal256[:, 3] = 255
n = 8192 * 8192
for kind in ("uniform", "photo"):
    rgba = bench.synthetic_image(kind, n, 0, 64, 0x5EED0005)
    for name, p in (("resurrect_64", pal), ("random 256", pal256)):
        cent = kg.palette_to_centroids(p)
        out = np.zeros(68, np.uint64)
        rc = L.kmg_tools_dither_list_stats(proc.handle, C.c_void_p(rgba.data_ptr()), 8192, 8192, C.c_void_p(cent.ctypes.data), len(p), C.c_void_p(out.ctypes.data))
        assert rc == 0, L.kmg_last_error()
        lanes, wave, pixels, nolist = (int(v) for v in out[:4])
        hist = out[4:].astype(float) / max(pixels, 1)
        mean = float((hist * np.arange(64)).sum())
        cum = np.cumsum(hist)
        print(f"{kind:8s} {name:13s}: lanes' words / (64 x the wave's longest) = {lanes / max(wave, 1):.3f}; mean list {mean:.2f} entries, "
              f"median {int(np.searchsorted(cum, 0.5))}, 90 % <= {int(np.searchsorted(cum, 0.9))}, 99 % <= {int(np.searchsorted(cum, 0.99))}; no list: {nolist / pixels:.5f}")
        print("   entries:share " + " ".join(f"{c}:{hist[c]:.3f}" for c in range(64) if hist[c] >= 0.002))
