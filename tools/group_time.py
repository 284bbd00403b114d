import sys, time, numpy as np
sys.path.insert(0, "kmeans-gpu_amd/python")
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
img = synth.uniform_rgba_numpy(0x5EED0003, 8192 * 8192).reshape(8192, 8192, 4)
def t(f, n=4):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:])
for shrink in (256, 0):
    p = kg.ImageProcessor(shrink_max_dim=shrink)
    print("shrink", shrink, "processor.reduce k=256 dither %.2f ms" % t(lambda: p.reduce(256, img, reduce_mode=1)))
    p.close()
    for ranks, flags in ((1, 0), (2, kg.GROUP_LOOPBACK), (4, kg.GROUP_LOOPBACK)):
        g = kg.Group(devices=[0] * ranks, flags=flags, shrink_max_dim=shrink)
        print("   group of", ranks, "reduce %.2f ms" % t(lambda: g.reduce(256, img, reduce_mode=1)))
        g.close()
