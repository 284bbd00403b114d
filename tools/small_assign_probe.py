#!/usr/bin/env python3
"""k_assign on the reference's default working size (an image shrunk to <= 256 x 256): time per launch by pixels per thread
(KMG_ASSIGN_PPT, set per run), labels only / sums only / both, k = 8, 64, 256.
    for P in 1 2 4 8; do KMG_ASSIGN_PPT=$P python tools/small_assign_probe.py; done"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
use_tools_library()
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg

tokyo = Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA").resize((256, 171))
rgba = np.array(tokyo).reshape(-1, 4)
n = rgba.shape[0]
d = torch.from_numpy(rgba).cuda()
st = torch.cuda.current_stream().cuda_stream
p = kg.ImageProcessor()
print("KMG_ASSIGN_PPT =", os.environ.get("KMG_ASSIGN_PPT", "auto"), " pixels", n)
for k in (8, 64, 256):
    s = kg.Lloyd(p, k)
    s.init_centroids(d.data_ptr(), 256, 171, st)
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for name, (l, a) in (("labels", (labels.data_ptr(), 0)), ("sums", (0, acc.data_ptr())), ("both", (labels.data_ptr(), acc.data_ptr()))):
        for _ in range(5):
            s.assign_accumulate(d.data_ptr(), n, l, a, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            s.assign_accumulate(d.data_ptr(), n, l, a, st)
        e1.record(); torch.cuda.synchronize()
        print(f"  k={k:3d} {name:7s} {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per call")
    s.close()
