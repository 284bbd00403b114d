import os, sys, time
sys.path.insert(0, "kmeans-gpu_amd/python"); sys.path.insert(0, "tests")
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg
tokyo = np.array(Image.open("tests/golden/tokyo.png").convert("RGBA"))
big = np.tile(tokyo, (16, 11, 1))[:8192, :8192].copy()
# add a smooth gradient so tiles are not identical
yy, xx = np.mgrid[0:8192, 0:8192]
big[..., 0] = np.clip(big[..., 0].astype(int) + (xx >> 9), 0, 255).astype(np.uint8)
big[..., 1] = np.clip(big[..., 1].astype(int) + (yy >> 9), 0, 255).astype(np.uint8)
n = 8192 * 8192
d = torch.from_numpy(big.reshape(-1, 4)).cuda()
p = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
res = {}
for k in (16, 256):
    for strat in ("scan", "table"):
        kg.set_strategy(strat)
        s = kg.Lloyd(p, k)
        s.init_centroids(d.data_ptr(), 8192, 8192, st) if k == 16 else s.set_centroids(np.concatenate([np.random.default_rng(1).uniform([0,-60,-60],[100,60,60],(k,3)), np.ones((k,1))],1).astype(np.float32), st)
        if strat == "table": s.bind_image(d.data_ptr(), n, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda"); acc = torch.zeros((k,4), dtype=torch.int64, device="cuda")
        for _ in range(3):
            s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st); s.update(acc.data_ptr(), st)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st); s.update(acc.data_ptr(), st)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
        res[(k, strat)] = (labels.clone(), acc.clone(), s.get_centroids(st))
        extra = s.debug_table_stats(st) if strat == "table" else {}
        print(f"k={k} {strat}: {ms:.3f} ms/iteration", {kk: extra[kk] for kk in ("occupied_cells", "distinct_colours") if kk in extra})
        s.close()
    a, b = res[(k, "scan")], res[(k, "table")]
    print("  identical:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)))
