#!/usr/bin/env python3
"""Strong scaling of ONE 8192 x 8192 image with the cube pass sharded by CELLS, as far as one GPU can show it
(python tools/strong_cells_per_rank.py <tag>  ->  gpurun_out/<tag>_strong_cells_per_rank.json).

Row bands alone leave the cube pass (which works on the image's colours, not its pixels) at full size on every rank:
<= 1.83x at N = 8 (profiles/r02_strong_per_rank.json).  Here every rank binds the WHOLE image's histogram and per iteration
  1. updates the centroids from the all-reduced sums,
  2. runs the cube pass on ITS share of the occupied cells (kmg_lloyd_set_cell_share),
  3. exchanges label tables: receives the other ranks' shares of the per-colour labels (16 MiB (N-1)/N) and cell entries --
     stand-in: one device-to-device copy of that many bytes -- and all-reduces the k x 4 sums (stand-in: a tensor add),
  4. writes the label map of ITS row band from the complete tables (kmg_lloyd_labels_from_tables).
Since round 6 the loop is the fused form (kmg_lloyd_accumulate_into + kmg_lloyd_labels_from_tables_update, what kmg_group_lloyd_step
runs with KMG_GROUP_CELLS | KMG_GROUP_FUSED_UPDATE): the cube pass adds into accumulators the previous label pass left zero, and the
band's label pass performs step 1 of the NEXT iteration from the all-reduced sums -- two launches per iteration instead of four.
Steps 1-4 are timed with HIP events on the launch stream for one rank at a time; the other ranks' shares are run on the same
GPU between the timed segments (untimed) so that the tables and sums are those of a real N-rank run -- the centroids of
every N are bit-identical to the unsharded loop (asserted).  The once-per-image cost is reported separately: binding a band
(its histogram) plus an element-wise add of two 64 MiB histograms as the all-reduce stand-in."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth

tag = sys.argv[1] if len(sys.argv) > 1 else "strong_cells"
W = H = 8192
n = W * H
k = 256
ITERS, WARM = 8, 2
kg.set_strategy("table")
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
sel = synth.uniform_rgba_at(synth.SEED_CFG3, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
proc.rgb_to_lab(torch.from_numpy(sel).cuda().data_ptr(), k, lab.data_ptr(), st)
torch.cuda.synchronize()
cent0 = np.ones((k, 4), np.float32); cent0[:, :3] = lab.cpu().numpy()


def reference():
    s = kg.Lloyd(proc, k); s.set_centroids(cent0, st); s.prepare(rgba.data_ptr(), n, True, st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
    for _ in range(WARM + ITERS):
        s.update(acc.data_ptr(), st)
        s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
    c = s.get_centroids(st); s.close()
    return c


want = reference()
out = {"workload": f"synthetic uniform {W}x{H}, k={k}, one Lloyd iteration with the label map; per-rank times on one GPU",
       "per_rank": {}, "bind": {}}
table_bytes = (16 << 20) + (1 << 17)            # per-colour labels + pair entries
for N in (1, 2, 4, 8):
    rows = H // N
    worst = None
    for r in sorted({0, N // 2, N - 1}):
        s = kg.Lloyd(proc, k); s.set_centroids(cent0, st); s.prepare(rgba.data_ptr(), n, True, st)
        band = rgba[r * rows * W:(r + 1) * rows * W]
        labels = torch.empty(rows * W, dtype=torch.int32, device="cuda")
        acc_r = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        acc_q = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        total = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        recv_src = torch.empty(max(table_bytes * (N - 1) // N, 4), dtype=torch.uint8, device="cuda")
        recv_dst = torch.empty_like(recv_src)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(WARM + ITERS)]

        def others():
            for q in range(N):
                if q != r:
                    s.set_cell_share(q, N, st)
                    s.assign_accumulate(rgba.data_ptr(), n, 0, acc_q.data_ptr(), st)
                    total.add_(acc_q)
        # initial assignment
        total.zero_(); others()
        s.set_cell_share(r, N, st); s.assign_accumulate(rgba.data_ptr(), n, 0, acc_r.data_ptr(), st); total.add_(acc_r)
        # (the fused loop: `mine` holds this rank's sums, then -- the all-reduce's stand-in adds the other ranks' -- the image's; the
        # label pass updates from it and leaves it zero)
        mine = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.update(total.data_ptr(), st)                                          # centroids after the initial assignment
        for it in range(WARM + ITERS):
            e = ev[it]
            total.zero_(); others()                                             # (the other ranks' shares under the current centroids, untimed)
            s.set_cell_share(r, N, st)
            e[2].record()
            s.accumulate_into(rgba.data_ptr(), n, mine.data_ptr(), st)          # 2 (no hand-over launch)
            if N > 1:
                recv_dst.copy_(recv_src)                                        # 3: all-gather stand-in
            mine.add_(total)                                                    #    all-reduce stand-in
            s.labels_from_tables_update(band.data_ptr(), rows * W, labels.data_ptr(), mine.data_ptr(), st)   # 4 + 1 of the next iteration
            e[3].record()
        torch.cuda.synchronize()
        ms = [ev[i][2].elapsed_time(ev[i][3]) for i in range(WARM, WARM + ITERS)]
        assert int(mine.abs().sum().item()) == 0                                # the label pass left the accumulators zero
        got = s.get_centroids(st)
        # the loop above performed WARM + ITERS + 1 updates after the initial assignment; the reference WARM + ITERS
        s.close()
        rec = {"rank": r, "rows": rows, "ms_per_iteration": float(np.mean(ms)), "min": float(np.min(ms)), "max": float(np.max(ms))}
        if worst is None or rec["ms_per_iteration"] > worst["ms_per_iteration"]:
            worst = rec
        print(N, rec, flush=True)
    out["per_rank"][f"N{N}"] = worst
    # once per image: the band's histogram + the all-reduce of the 64 MiB histogram (stand-in: an add)
    b = kg.Lloyd(proc, k); b.set_centroids(cent0, st)
    band = rgba[:rows * W]
    b.prepare(band.data_ptr(), rows * W, True, st); torch.cuda.synchronize()
    t = time.perf_counter(); b.prepare(band.data_ptr(), rows * W, True, st); torch.cuda.synchronize()
    t_bind = (time.perf_counter() - t) * 1e3
    h0 = torch.zeros(1 << 24, dtype=torch.int32, device="cuda"); h1 = torch.ones_like(h0)
    h0.add_(h1); torch.cuda.synchronize()
    t = time.perf_counter(); h0.add_(h1); torch.cuda.synchronize()
    out["bind"][f"N{N}"] = {"band_bind_ms": t_bind, "histogram_add_64MiB_ms": (time.perf_counter() - t) * 1e3}
    b.close()

# the sharded loop is the same loop: its centroids after the same number of updates
s = kg.Lloyd(proc, k); s.set_centroids(cent0, st); s.prepare(rgba.data_ptr(), n, True, st)
acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda"); tot = torch.zeros_like(acc)
for it in range(WARM + ITERS + 1):
    if it:
        s.update(tot.data_ptr(), st)
    tot.zero_()
    for q in range(4):
        s.set_cell_share(q, 4, st); s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st); tot.add_(acc)
out["centroids_equal_unsharded"] = bool(np.array_equal(s.get_centroids(st).view(np.uint32), want.view(np.uint32)))
s.close()
t1 = out["per_rank"]["N1"]["ms_per_iteration"]
out["speedup"] = {f"N{N}": t1 / out["per_rank"][f"N{N}"]["ms_per_iteration"] for N in (1, 2, 4, 8)}
out["row_bands_only_r02"] = {"N1": 1.0, "N2": 1.35, "N4": 1.60, "N8": 1.83}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_strong_cells_per_rank.json"), "w"), indent=1)
print(json.dumps(out["speedup"]), out["centroids_equal_unsharded"])
