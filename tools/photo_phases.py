#!/usr/bin/env python3
"""The Lloyd iteration on the tiled photograph of bench.py (k = 256, colour table), 12 + 10 iterations -- run under
rocprofv3 --kernel-trace --stats for the per-kernel times (tools/photo_phases.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
import bench
n, k = 8192 * 8192, 256
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
rgba = bench.synthetic_image("photo", n, 0, k, 0x5EED0B10)
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st); torch.cuda.synchronize()
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, st)
s.prepare(rgba.data_ptr(), n, True, st)
labels = torch.empty(n, dtype=torch.int32, device="cuda"); acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
for _ in range(22):
    s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
torch.cuda.synchronize()
s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
print(s.debug_table_stats(st))
