#!/usr/bin/env python3
"""The dominance phase of the general cube pass (k_cube_prune, KMG_CUBE_PRUNE=1 in the tools build), on the GPU box:
results against the pass without it (per-colour labels, sums, the exhaustive check over all 2^24 colours) and what it removes.
    python tools/cube_prune_probe.py > gpurun_out/prune_probe.txt"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    import hashlib, numpy as np, torch
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    proc = kg.ImageProcessor(shrink_max_dim=0)
    st = torch.cuda.current_stream().cuda_stream
    n = 8192 * 8192
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    for k in (64, 256):
        lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
        sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
        proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
        cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
        s = kg.Lloyd(proc, k); s.set_centroids(cent, st); s.bind_image(rgba.data_ptr(), n, st)
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        labels = torch.empty(n, dtype=torch.int32, device="cuda")
        h = hashlib.sha256()
        for it in range(8):
            s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
            torch.cuda.synchronize()
            h.update(acc.cpu().numpy().tobytes())
            if it in (0, 7):
                h.update(labels.cpu().numpy().tobytes())
                d = s.debug_table_stats(st)
                print(f"k={k} it={it}: check {s.debug_check_table(st)} pairs {s.debug_check_pairs(st)[0]} decided {d['sub_cells_decided']} "
                      f"scanned {d['sub_cells_scanned']} cands/scanned {d['scan_candidates']/max(d['sub_cells_scanned'],1):.2f} "
                      f"pruned cands {d['candidates_pruned']} pruned to one {d['sub_cells_pruned_to_one']} "
                      f"one-label sub-cells {d['sub_cells_one_label']} of {d['occupied_sub_cells']}", flush=True)
            s.update(acc.data_ptr(), st)
        print(f"k={k} digest {h.hexdigest()[:16]}", flush=True)
        s.close()
    sys.exit(0)
for prune in ("0", "1"):
    env = dict(os.environ, KMG_CUBE_PRUNE=prune)
    use_tools_library(env)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(f"--- KMG_CUBE_PRUNE={prune}\n{r.stdout}{r.stderr[-2000:] if r.returncode else ''}", flush=True)
