#!/usr/bin/env python3
"""The data behind table_pays() (csrc/kmg_lloyd.hip), round 6: one Lloyd iteration with its label map (kmg_lloyd_assign_update) by the
per-pixel scan and by the colour table, on noise, Gaussian blobs and the tiled photograph, with what the binding finds in the
image (occupied cells, hot cells) and the library's own choice.    python tools/costmodel_sweep.py > gpurun_out/costmodel_sweep.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
import bench

proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
KS = [int(v) for v in os.environ.get("KS", "8,16,32,64,128,256").split(",")]
for kind in os.environ.get("KINDS", "uniform,photo,blobs").split(","):
    for logn in [int(v) for v in os.environ.get("LOGN", "18,19,20,21,22,24").split(",")]:
        n = 1 << logn
        rgba = bench.synthetic_image(kind, n, 0, 64, 0x5EED0B10)
        labels = torch.empty(n, dtype=torch.int32, device="cuda")
        for k in KS:
            sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
            lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
            proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
            torch.cuda.synchronize()
            cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
            acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            kg.set_strategy("auto")
            s = kg.Lloyd(proc, k); s.set_centroids(cent, st)
            auto = s.prepare(rgba.data_ptr(), n, True, st); s.close()
            res, info, bind = {}, (0, 0), 0.0
            for strat in ("scan", "table"):
                kg.set_strategy(strat)
                s = kg.Lloyd(proc, k); s.set_centroids(cent, st)
                s.prepare(rgba.data_ptr(), n, True, st)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                s.prepare(rgba.data_ptr(), n, True, st)
                torch.cuda.synchronize()
                if strat == "table":
                    bind = time.perf_counter() - t0
                    info = s.debug_bound_image()
                best = 1e9
                for _ in range(3):
                    s.set_centroids(cent, st)
                    for _ in range(3):
                        s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20):
                        s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 20)
                res[strat] = best
                s.close()
            kg.set_strategy("auto")
            # (the per-iteration times decide; the binding -- spread over 16 passes -- is shown beside them: the model counts it
            # before it binds and not after)
            regret = res[auto] / min(res.values()) - 1.0
            print(f"{kind:8s} n=2^{logn} k={k:3d} occ {info[0]:5d} hot {info[1]:2d} | scan {res['scan'] * 1e6:8.1f} us  table {res['table'] * 1e6:8.1f} us "
                  f"(+ bind {bind * 1e6:6.0f} / 16) | auto = {auto:5s} {'ok' if regret < 0.02 else 'slower by %.0f %%' % (regret * 100)}", flush=True)
proc.close()
