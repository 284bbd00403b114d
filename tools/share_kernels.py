#!/usr/bin/env python3
"""One rank's iteration of the cell-sharded strong-scaling loop (rank R of N on ONE image, tools/strong_cells_per_rank.py) as a plain
loop for rocprofv3 --kernel-trace --stats: update, the rank's share of the cube pass, the label map of its band.  The other ranks'
shares are NOT run (their tables stay stale: the kernel times do not depend on them).
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/share_kernels.py 8 4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
N, R = int(sys.argv[1]), int(sys.argv[2])
W = H = 8192; n = W * H; k = 256
kg.set_strategy("table")
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
sel = synth.uniform_rgba_at(synth.SEED_CFG3, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
proc.rgb_to_lab(torch.from_numpy(sel).cuda().data_ptr(), k, lab.data_ptr(), st)
torch.cuda.synchronize()
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
s = kg.Lloyd(proc, k); s.set_centroids(cent, st); s.prepare(rgba.data_ptr(), n, True, st)
rows = H // N
band = rgba[R * rows * W:(R + 1) * rows * W]
labels = torch.empty(rows * W, dtype=torch.int32, device="cuda")
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)       # the whole cube once: real sums for the updates below
s.set_cell_share(R, N, st)
FUSED = len(sys.argv) > 3 and sys.argv[3] == "fused"      # kmg_group_lloyd_step with KMG_GROUP_CELLS | KMG_GROUP_FUSED_UPDATE
keep = acc.clone()
for it in range(30):
    if FUSED:
        s.accumulate_into(rgba.data_ptr(), n, acc.data_ptr(), st)
        acc.copy_(keep)                                           # (the other ranks' sums are not run: the whole image's, for sane centroids)
        s.labels_from_tables_update(band.data_ptr(), rows * W, labels.data_ptr(), acc.data_ptr(), st)
    else:
        s.update(acc.data_ptr(), st)
        s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
        s.labels_from_tables(band.data_ptr(), rows * W, labels.data_ptr(), st)
torch.cuda.synchronize()
s.close()
print("done")
