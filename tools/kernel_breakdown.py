#!/usr/bin/env python3
"""Per-kernel HIP-event times of one Lloyd iteration on the benchmark workload (all kernels timed, so the
iteration itself runs a little slower than in bench.py, which times only the heavy ones)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n = 8192 * 8192
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, st)
print("strategy:", s.prepare(rgba.data_ptr(), n, True, st))
def it():
    s.update(acc.data_ptr(), st)
    s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
for _ in range(3): it()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): it()
torch.cuda.synchronize(); print(f"iteration, untimed kernels: {(time.perf_counter() - t) / 20 * 1e3:.4f} ms")
s.profile(True)
for _ in range(20): it()
torch.cuda.synchronize()
for name, (ms, cnt) in s.profile_read().items():
    print(f"  {name:20s} {ms / cnt * 1e3:8.1f} us  x{cnt}")
