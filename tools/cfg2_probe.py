#!/usr/bin/env python3
"""BASELINE config 2 (4096 x 4096, k = 16): table statistics of the colour-table pass (how much of the cube the bounds
decide, how many pixels the pair entries resolve) -- the numbers the cfg2 design in DESIGN.md section 4 quotes.
    python tools/cfg2_probe.py [iterations]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    st = torch.cuda.current_stream().cuda_stream
    proc = kg.ImageProcessor(shrink_max_dim=0)
    n, k = bench.CFG2_WIDTH * bench.CFG2_HEIGHT, bench.CFG2_K
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG2, n, device="cuda")
    sel = synth.uniform_rgba_at(synth.SEED_CFG2, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
    d_sel = torch.from_numpy(sel).cuda()
    lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
    proc.rgb_to_lab(d_sel.data_ptr(), k, lab.data_ptr(), st)
    torch.cuda.synchronize()
    cent = np.ones((k, 4), np.float32)
    cent[:, :3] = lab.cpu().numpy()
    kg.set_strategy("table")
    s = kg.Lloyd(proc, k)
    s.set_centroids(cent, st)
    s.prepare(rgba.data_ptr(), n, True, st)
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for it in range(iters):
        s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        stats = s.debug_table_stats(st)
        _, resolved, total = s.debug_check_pairs(st)
        stats["pixels_resolved_in_lds"] = resolved / max(total, 1)
        print(json.dumps({"iteration": it, **stats}), flush=True)
        s.update(acc.data_ptr(), st)
    s.close()
    proc.close()


if __name__ == "__main__":
    main()
