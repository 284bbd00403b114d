#!/usr/bin/env python3
"""The output passes of the path on 8192 x 8192 (run under rocprofv3 by tools/profile_apply.sh, or alone for host timings):
   find -m dither / replace / meld with the 64-entry resurrect_64 palette (BASELINE config 5) and the dither pass with k = 256
   centroids (BASELINE config 3's output pass: farthest-point init + 8 Lloyd iterations on the same image)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = 8192; n = W * W
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal = np.array(sorted(set(map(tuple, px))), np.uint8)
cent64 = kg.palette_to_centroids(pal)
proc = kg.ImageProcessor(shrink_max_dim=0, max_iterations=9)
rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
s = kg.Lloyd(proc, 256)
s.init_centroids(rgba.data_ptr(), W, W, st)
s.run(rgba.data_ptr(), n, 0, st)
cent256 = s.get_centroids(st)
s.close()
for name, cent, mode in (("dither k=64", cent64, kg.ReduceMode.Dither), ("dither k=256", cent256, kg.ReduceMode.Dither),
                         ("replace k=64", cent64, kg.ReduceMode.Replace), ("meld k=64", cent64, kg.ReduceMode.Meld)):
    proc.apply(rgba.data_ptr(), W, W, 0, cent, mode, out.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        proc.apply(rgba.data_ptr(), W, W, 0, cent, mode, out.data_ptr(), st)
    torch.cuda.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t) / reps * 1e3:8.3f} ms per call", flush=True)
