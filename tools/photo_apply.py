#!/usr/bin/env python3
"""The output passes on the tiled photograph with the k-means centroids of that photograph (k = 256 and 16): init + run +
dither / replace / meld timings (host clock, warm)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
import bench
W = 8192; n = W * W
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
rgba = bench.synthetic_image("photo", n, 0, 256, 0x5EED0B10)
out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
for k in (256, 16):
    s = kg.Lloyd(proc, k)
    def timed(fn, reps=1):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
    s.init_centroids(rgba.data_ptr(), W, W, st)
    t_init, _ = timed(lambda: s.init_centroids(rgba.data_ptr(), W, W, st))
    t_run, it = timed(lambda: s.run(rgba.data_ptr(), n, labels.data_ptr(), st))
    cent = s.get_centroids(st)
    s.close()
    line = f"k={k}: init {t_init:.2f} ms, run {t_run:.2f} ms ({it} iterations)"
    for name, mode in (("dither", kg.ReduceMode.Dither), ("replace", kg.ReduceMode.Replace), ("meld", kg.ReduceMode.Meld)):
        proc.apply(rgba.data_ptr(), W, W, 0, cent, mode, out.data_ptr(), st)
        t, _ = timed(lambda: proc.apply(rgba.data_ptr(), W, W, 0, cent, mode, out.data_ptr(), st), 3)
        line += f", {name} {t:.2f} ms"
    print(line, flush=True)
