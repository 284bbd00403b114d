#!/usr/bin/env python3
"""Candidate-set / label-table statistics of the colour-table strategy on the benchmark workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n = 8192 * 8192
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
for k in (16, 64, 256):
    lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
    sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
    proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
    cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
    s = kg.Lloyd(proc, k); s.set_centroids(cent, st); s.bind_image(rgba.data_ptr(), n, st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for it in range(12):
        s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st); 
        if it in (0, 3, 11):
            d = s.debug_table_stats(st)
            print(f"k={k} it={it}: mean candidates/cell {d['candidates_total']/d['occupied_cells']:.2f} max {d['max_candidates']} "
                  f"one-candidate cells {d['cells_one_candidate']/d['occupied_cells']:.3f} one-label cells {d['cells_one_label']/d['occupied_cells']:.3f} "
                  f"one-label sub-cells {d['sub_cells_one_label']/d['occupied_sub_cells']:.3f} colours {d['distinct_colours']} | "
                  f"sub-cells decided by bounds {d['sub_cells_decided']} scanned {d['sub_cells_scanned']} "
                  f"candidates/scanned sub-cell {d['scan_candidates']/max(d['sub_cells_scanned'],1):.2f} unlisted cells {d['cells_unlisted']}")
        s.update(acc.data_ptr(), st)
    s.close()
