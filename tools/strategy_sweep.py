#!/usr/bin/env python3
"""Times one Lloyd iteration (update + assign/accumulate, labels written) with the per-pixel scan and
with the colour table over a grid of (pixels, k): the data behind table_pays() in csrc/kmg_lloyd.hip."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np
import torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth

proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
rows = []
for logn in (18, 20, 22, 24, 26):
    n = 1 << logn
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    for k in (4, 8, 16, 32, 64, 128, 256, 512):
        lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
        proc.rgb_to_lab(rgba.data_ptr(), k, lab.data_ptr(), st)
        cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        res = {}
        for strat in ("scan", "table"):
            for want_labels in (True, False):
                s = kg.Lloyd(proc, k)
                s.set_centroids(cent, st)
                t_bind = 0.0
                if strat == "table":
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    s.bind_image(rgba.data_ptr(), n, st)
                    torch.cuda.synchronize(); t_bind = time.perf_counter() - t0
                lp = labels.data_ptr() if want_labels else 0
                for _ in range(2):
                    s.assign_accumulate(rgba.data_ptr(), n, lp, acc.data_ptr(), st); s.update(acc.data_ptr(), st)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    s.assign_accumulate(rgba.data_ptr(), n, lp, acc.data_ptr(), st); s.update(acc.data_ptr(), st)
                torch.cuda.synchronize()
                res[(strat, want_labels)] = ((time.perf_counter() - t0) / reps * 1e3, t_bind * 1e3)
                s.close()
        print(f"n=2^{logn} k={k:4d}  scan {res[('scan',True)][0]:8.3f} ms | table+labels {res[('table',True)][0]:8.3f} ms, "
              f"table sums-only {res[('table',False)][0]:8.3f} ms (bind {res[('table',True)][1]:.2f} ms) | scan sums-only {res[('scan',False)][0]:8.3f}", flush=True)
proc.close()
