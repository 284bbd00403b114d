#!/usr/bin/env python3
"""kmg_palette of tests/golden/tokyo.png at the reference's defaults, k from argv (default 256), five calls -- for
rocprofv3 --kernel-trace: which launches a default call is made of, and how much of its wall time they fill
(tools/default_palette_trace.sh prints that)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np
from PIL import Image
import kmeans_gpu_amd as kg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
p = kg.ImageProcessor()
for i in range(5):
    t = time.perf_counter()
    pal = p.palette(k, tokyo)
    print(f"call {i}: {(time.perf_counter() - t) * 1e3:.2f} ms, {len(pal)} colours", flush=True)
