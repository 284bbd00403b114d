#!/usr/bin/env python3
"""cfg5 timing (find -m dither / replace / meld at 8192^2, resurrect_64 and 256 random colours) of whatever library KMG_LIBRARY names.
    python tools/dither_time.py [uniform|photo]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
import bench
from PIL import Image
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal = np.array(sorted(set(map(tuple, px))), np.uint8)
rng = np.random.default_rng(5)
pal256 = rng.integers(0, 256, (256, 4), dtype=np.uint8); pal256[:, 3] = 255
n = 8192 * 8192
for kind in sys.argv[1:] or ["uniform"]:
    rgba = bench.synthetic_image(kind, n, 0, 64, 0x5EED0005)
    out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
    for name, p in (("resurrect_64", pal), ("random 256", pal256)):
        cent = kg.palette_to_centroids(p)
        for mode in (kg.ReduceMode.Dither,):
            proc.apply(rgba.data_ptr(), 8192, 8192, 0, cent, mode, out.data_ptr(), st)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5):
                proc.apply(rgba.data_ptr(), 8192, 8192, 0, cent, mode, out.data_ptr(), st)
            torch.cuda.synchronize()
            print(f"{kind:8s} {name:13s} {mode.name:8s} {(time.perf_counter() - t) / 5 * 1e3:.3f} ms  digest {int(out.view(torch.int32).sum().item()) & 0xFFFFFFFF:08x}", flush=True)
