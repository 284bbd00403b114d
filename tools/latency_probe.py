import sys, time
sys.path.insert(0, "kmeans-gpu_amd/python"); sys.path.insert(0, "tests")
import numpy as np, torch
from PIL import Image
import kmeans_gpu_amd as kg
tokyo = np.array(Image.open("tests/golden/tokyo.png").convert("RGBA"))
p = kg.ImageProcessor()
for k in (8, 16):
    for mode in (0, 1):
        p.reduce(k, tokyo, reduce_mode=mode)
        t = time.perf_counter(); n = 10
        for _ in range(n): p.reduce(k, tokyo, reduce_mode=mode)
        print(f"reduce k={k} mode={mode}: {(time.perf_counter()-t)/n*1e3:.2f} ms")
pal = np.array(sorted(set(map(tuple, np.array(Image.open('tests/golden/apollo-1x.png').convert('RGBA')).reshape(-1,4)))), np.uint8)
p.find(tokyo, pal, 1)
t = time.perf_counter()
for _ in range(10): p.find(tokyo, pal, 1)
print(f"find dither apollo: {(time.perf_counter()-t)/10*1e3:.2f} ms")
# cfg5: find + dither, 64-entry palette, 8192x8192 on device buffers
from kmeans_gpu_amd import synth
pal64 = np.array(sorted(set(map(tuple, np.array(Image.open('tests/golden/resurrect_64.png').convert('RGBA')).reshape(-1,4)))), np.uint8)
cent = kg.palette_to_centroids(pal64)
n = 8192*8192
rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
out = torch.empty((n,4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode in (0, 1, 2):
    p.apply(rgba.data_ptr(), 8192, 8192, 0, cent, mode, out.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): p.apply(rgba.data_ptr(), 8192, 8192, 0, cent, mode, out.data_ptr(), st)
    torch.cuda.synchronize()
    print(f"cfg5 apply mode={mode} k=64 8192^2: {(time.perf_counter()-t)/3*1e3:.3f} ms")
