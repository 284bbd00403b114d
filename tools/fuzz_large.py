#!/usr/bin/env python3
"""tools/fuzz_parity.py for LARGE images (1 ... 25 Mpx): the sizes where the colour-table strategy, the one-launch cube pass
with its pass-to-pass re-deal (32 < k <= 256), the three-launch pass of images with hot cells and the several-picks-per-launch
initialisation over the colours actually run.  strategy = table against strategy = scan through the C ABI: init centroids,
iterations, final centroids, labels, replace / dither / meld bytes.   usage: fuzz_large.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
import numpy as np, torch
import kmeans_gpu_amd as kg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.current_stream().cuda_stream
try:
    from PIL import Image
    PHOTO = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
except Exception:                                             # noqa: BLE001 -- without PIL the photograph kind is left out
    PHOTO = None


def image(kind, w, h):
    n = w * h
    if kind == "noise":
        a = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    elif kind == "few":
        pal = rng.integers(0, 256, (int(rng.integers(1, 3000)), 4), dtype=np.uint8)
        a = pal[rng.integers(0, pal.shape[0], n)]
    elif kind == "blobs":
        c = rng.integers(0, 256, (int(rng.integers(2, 400)), 3))
        a = np.zeros((n, 4), np.uint8)
        a[:, :3] = np.clip(c[rng.integers(0, c.shape[0], n)] + rng.normal(0, rng.uniform(1, 25), (n, 3)), 0, 255).astype(np.uint8)
    elif kind == "photo" and PHOTO is not None:
        ph, pw = PHOTO.shape[:2]
        oy, ox = int(rng.integers(0, ph)), int(rng.integers(0, pw))
        ys = (np.arange(h) + oy) % ph
        xs = (np.arange(w) + ox) % pw
        a = PHOTO[ys][:, xs].reshape(n, 4).copy()
    else:  # dark noise: crowded cells near black (hot cells and long candidate lists)
        a = (rng.integers(0, 256, (n, 4)) ** 2 // 700).astype(np.uint8)
    a[:, 3] = rng.integers(0, 256, n, dtype=np.uint8)
    return np.ascontiguousarray(a)


def run(strategy, rgba, w, h, k, cent_fixed):
    kg.set_strategy(strategy)
    n = w * h
    p = kg.ImageProcessor(shrink_max_dim=0, max_iterations=int(os.environ.get("ITER", "12")))
    d = torch.from_numpy(rgba).cuda()
    s = kg.Lloyd(p, k)
    s.init_centroids(d.data_ptr(), w, h, st)
    c_init = s.get_centroids(st).copy()
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    it = s.run(d.data_ptr(), n, labels.data_ptr(), st)
    c_run = s.get_centroids(st).copy()
    outs = []
    for mode in (kg.ReduceMode.Replace, kg.ReduceMode.Dither, kg.ReduceMode.Meld):
        out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent_fixed, mode, out.data_ptr(), st)
        outs.append(out.cpu().numpy())
    torch.cuda.synchronize()
    res = (c_init.view(np.uint32), it, c_run.view(np.uint32), labels.cpu().numpy(), *outs)
    s.close(); p.close()
    return res


bad = 0
for case in range(cases):
    kind = ["noise", "few", "blobs", "photo", "dark"][int(rng.integers(0, 5))]
    w = int(rng.integers(1024, 6145)); h = int(rng.integers(512, 4097))
    k = int(rng.choice([16, 32, 33, 40, 64, 100, 128, 200, 255, 256, 257, 512]))
    rgba = image(kind, w, h)
    pal = rng.integers(0, 256, (k, 4), dtype=np.uint8); pal[:, 3] = 255
    cent_fixed = kg.palette_to_centroids(pal)
    a = run("brute", rgba, w, h, k, cent_fixed)
    b = run("table", rgba, w, h, k, cent_fixed)
    names = ["init centroids", "iterations", "final centroids", "labels", "replace", "dither", "meld"]
    diff = [nm for nm, x, y in zip(names, a, b) if not np.array_equal(x, y)]
    print(f"case {case}: {kind} {w}x{h} k={k} iterations {a[1]}" + (f"  MISMATCH {diff}" if diff else ""), flush=True)
    bad += 1 if diff else 0
print(f"{cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
