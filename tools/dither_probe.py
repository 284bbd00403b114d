#!/usr/bin/env python3
"""cfg5 output pass (find + ordered dither, 8192x8192): all-centroid scan vs candidate-pruned pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
from PIL import Image
W = 8192
n = W * W
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
out = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
pal64 = np.array(sorted(set(map(tuple, px))), np.uint8)
rng = np.random.default_rng(3)
tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
big = np.tile(tokyo, (16, 11, 1))[:W, :W].copy()
yy, xx = np.mgrid[0:W, 0:W]
big[..., 0] = np.clip(big[..., 0].astype(int) + (xx >> 9), 0, 255).astype(np.uint8)
big[..., 1] = np.clip(big[..., 1].astype(int) + (yy >> 9), 0, 255).astype(np.uint8)
photo = torch.from_numpy(big.reshape(-1, 4)).cuda()
images = {"noise": rgba, "photo": photo}
sizes = [(W, W), (4096, 4096), (2048, 2048), (1024, 1024)]
for name, pal in (("resurrect64", pal64), ("random16", rng.integers(0, 256, (16, 4), dtype=np.uint8)),
                  ("random256", rng.integers(0, 256, (256, 4), dtype=np.uint8))):
    pal[:, 3] = 255
    cent = kg.palette_to_centroids(pal)
    for iname, img in images.items():
      for (w, h) in sizes:
        res = {}
        for strat in ("brute", "table"):
            kg.set_strategy(strat)
            proc = kg.ImageProcessor(shrink_max_dim=0)
            proc.apply(img.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3):
                proc.apply(img.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
            torch.cuda.synchronize()
            res[strat] = ((time.perf_counter() - t) / 3 * 1e3, out[: w * h].clone())
            proc.close()
        same = bool(torch.equal(res["brute"][1], res["table"][1]))
        print(f"{name} {iname} {w}x{h}: scan {res['brute'][0]:.3f} ms, pruned {res['table'][0]:.3f} ms, identical {same}")
        if (w, h) in ((W, W), (2048, 2048)):
            meld = {}
            for strat in ("brute", "table"):
                kg.set_strategy(strat)
                proc = kg.ImageProcessor(shrink_max_dim=0)
                proc.apply(img.data_ptr(), w, h, 0, cent, kg.ReduceMode.Meld, out.data_ptr(), st)
                torch.cuda.synchronize(); t = time.perf_counter()
                proc.apply(img.data_ptr(), w, h, 0, cent, kg.ReduceMode.Meld, out.data_ptr(), st)
                torch.cuda.synchronize()
                meld[strat] = ((time.perf_counter() - t) * 1e3, out[: w * h].clone())
                proc.close()
            print(f"   meld {w}x{h}: scan {meld['brute'][0]:.3f} ms, pruned {meld['table'][0]:.3f} ms, identical {bool(torch.equal(meld['brute'][1], meld['table'][1]))}")
