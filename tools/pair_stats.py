#!/usr/bin/env python3
"""Share of the pixels the label pass resolves from the LDS pair entries alone (benchmark workload)."""
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
        s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
        if it in (0, 3, 11):
            bad, resolved, total = s.debug_check_pairs(st)
            print(f"k={k} it={it}: mismatches {bad}  per-colour gathers {1 - resolved / total:.4f} of the pixels")
        s.update(acc.data_ptr(), st)
    s.close()
