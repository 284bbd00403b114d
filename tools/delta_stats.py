#!/usr/bin/env python3
"""How many pixels / 8x8x8 colour cells change label from one Lloyd iteration to the next on the
benchmark workload?  (Decides whether an incremental label pass could pay.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n, k = 8192 * 8192, 256
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
px = rgba.view(torch.int32).view(-1)
cell = (((px & 255) >> 3) << 10) | ((((px >> 8) & 255) >> 3) << 5) | (((px >> 16) & 255) >> 3)
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, st); s.bind_image(rgba.data_ptr(), n, st)
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
labels = torch.zeros(n, dtype=torch.int32, device="cuda")
prev = None
for it in range(28):
    s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    if prev is not None:
        ch = labels != prev
        cells = torch.unique(cell[ch]).numel()
        print(f"iteration {it}: pixels changed {ch.float().mean().item()*100:.2f} %  cells with a change {cells/32768*100:.1f} %")
    prev = labels.clone()
    s.update(acc.data_ptr(), st)
