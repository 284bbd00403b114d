#!/usr/bin/env python3
"""Knock-out timing of the cfg5 output pass (k_dither_pruned), run on the GPU box; results of knocked-out runs are wrong by design."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    import time, numpy as np, torch
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    from PIL import Image
    W = 8192; n = W * W
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
    pal = np.array(sorted(set(map(tuple, px))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    proc = kg.ImageProcessor(shrink_max_dim=0)
    for _ in range(2):
        proc.apply(rgba.data_ptr(), W, W, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        proc.apply(rgba.data_ptr(), W, W, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t) / 5 * 1e3:.3f} ms")
    sys.exit(0)
for name, kn in (("baseline", 0), ("no mask gather", 1), ("no candidates", 2), ("no Lab conversion", 4), ("no palette gather", 8),
                 ("no gather, no candidates", 3), ("none of the four", 15)):
    env = dict(os.environ, KMG_DITHER_KNOCK=str(kn))
    use_tools_library(env)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(f"{name:30s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
