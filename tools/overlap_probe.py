#!/usr/bin/env python3
"""Timing-only probe: how much does running the label pass of iteration i on a second stream, beside the
candidates + cube pass of iteration i+1, shorten the iteration?  (No double buffering here, so the
labels written during the probe are not meaningful -- this only measures the overlap.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
proc = kg.ImageProcessor(shrink_max_dim=0)
n, k = 8192 * 8192, 256
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), main.cuda_stream)
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, main.cuda_stream); s.bind_image(rgba.data_ptr(), n, main.cuda_stream)
def serial(iters):
    for _ in range(iters):
        s.update(acc.data_ptr(), main.cuda_stream)
        s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), main.cuda_stream)
def overlapped(iters):
    ev = None
    for _ in range(iters):
        s.update(acc.data_ptr(), main.cuda_stream)
        s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), main.cuda_stream)
        e = torch.cuda.Event(); e.record(main)
        side.wait_event(e)
        s.labels(rgba.data_ptr(), n, labels.data_ptr(), side.cuda_stream)
s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), main.cuda_stream)
for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    fn(3); torch.cuda.synchronize(); t = time.perf_counter(); fn(20); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per iteration")
