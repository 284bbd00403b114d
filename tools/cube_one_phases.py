#!/usr/bin/env python3
"""Where the one-launch cube pass (k_cube_one, 32 < k <= 256) spends its time, on the GPU box: the tools build stamps
s_memrealtime (100 MHz) at the phase boundaries of every workgroup (thread 0).  Prints, over the 512 workgroups of one
pass on the benchmark image (8192^2 noise, k = 256): mean / max duration of each phase and the pass's span.
    python tools/cube_one_phases.py > gpurun_out/cube_one_phases.txt
    SHARE=4/8 [FUSED=1] python tools/cube_one_phases.py      the same for one rank's share of the cube (cell-sharded loop)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
use_tools_library()
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n = 8192 * 8192
k = int(os.environ.get("K", "256"))
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, st); s.bind_image(rgba.data_ptr(), n, st)
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(int(os.environ.get("PASSES", "4"))):     # (the pass re-deals its tasks between workgroups from pass to pass: CubeBalance)
    s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
share = os.environ.get("SHARE")            # "R/N": the pass over rank R's share of the cube (the strong-scaling loop's per-rank launch)
if share:
    R, N = (int(x) for x in share.split("/"))
    s.set_cell_share(R, N, st)
    for _ in range(4):
        if os.environ.get("FUSED"):
            s.accumulate_into(rgba.data_ptr(), n, acc.data_ptr(), st)
        else:
            s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
torch.cuda.synchronize()
L = kg.lib()
buf = np.zeros((512, 12), np.uint64)  # (only the first grid-size rows are written)
rc = L.kmg_tools_cube_one_stamps(C.c_void_p(buf.ctypes.data), C.c_uint32(buf.size))
assert rc == 0, rc
t = buf.astype(np.int64)
t = t[t[:, 0] > 0]            # (rows of workgroups that ran)
t0 = t[:, 0].min()
counts = t[:, 8:11]
t = t[:, :8]
us = (t - t0) / 100.0
names = ["start", "1a candidates (wave 0)", "1b sweeps (wave 0)", "tests (wave 0)", "1c decisions (wave 0)", "scan + entries (wave 0)", "-", "own cells without items (wave 0)"]
t[:, 6] = t[:, 5]
print(f"{len(t)} workgroups")
print(f"k={k}: pass span {us[:, 7].max():.1f} us (first start -> last workgroup's wave 0 done); starts spread over {us[:, 0].max():.1f} us")
for i in range(1, 8):
    d = us[:, i] - us[:, i - 1]
    print(f"  {names[i]:28s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us   (done at: mean {us[:, i].mean():6.1f}, max {us[:, i].max():6.1f})")
total = us[:, 7] - us[:, 0]
print("per workgroup: items mean %.1f max %d; pending cells mean %.1f max %d; tests mean %.1f max %d" % (counts[:, 0].mean(), counts[:, 0].max(), counts[:, 1].mean(), counts[:, 1].max(), counts[:, 2].mean(), counts[:, 2].max()))
for nm, col in (("items", 0), ("pending", 1), ("tests", 2)):
    print("  corr(total time, %s) = %.2f" % (nm, np.corrcoef(total, counts[:, col])[0, 1]))
order = np.argsort(total)
print("slowest workgroups:", [(int(i), round(float(total[i]), 1), counts[i].tolist()) for i in order[-6:]])
print("fastest workgroups:", [(int(i), round(float(total[i]), 1), counts[i].tolist()) for i in order[:6]])
s.close()
