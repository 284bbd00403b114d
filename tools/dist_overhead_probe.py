#!/usr/bin/env python3
"""Host and device cost of the per-iteration collective with ONE rank (RCCL initialised, all-reduce of the k x 4
accumulators really issued): how much of a 0.3 ms iteration does the exchange step add, and is the loop host-bound?
Run on the GPU box:  python tools/dist_overhead_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sharded_harness import ShardedLloyd
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
W = 8192; n = W * W; k = 256
rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
labels = torch.empty(n, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
proc = kg.ImageProcessor(shrink_max_dim=0)
sel = rgba[(torch.arange(k, device="cuda") * (n // k))].contiguous()
lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), stream); torch.cuda.synchronize()
cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()

def run(mode, steps=40):
    s = kg.Lloyd(proc, k); s.set_centroids(cent, stream)
    strategy = s.prepare(rgba.data_ptr(), n, True, stream)
    sh = ShardedLloyd(s, k, rgba, labels, stream=stream)
    sh.split_labels = strategy == "table"; sh.pipeline = False
    if mode == "none":
        sh.world = 1
    elif mode == "sync":                      # all-reduce on the compute stream, in order
        sh.world = 2; sh.collective = None
        sh.exchange = lambda async_op=False: dist.all_reduce(sh.acc, op=dist.ReduceOp.SUM) and None
    elif mode in ("async", "async+8cu"):      # as ShardedLloyd does it with world > 1
        sh.world = 2
        if mode == "async+8cu":               # ... including the CUs it leaves to the collective
            s.reserve_cus(8)
    sh.prime()
    for _ in range(5): sh.iterate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): sh.iterate()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s.close()
    return (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3

for mode in ("none", "sync", "async", "async+8cu", "none"):
    host, total = run(mode)
    print(f"{mode:9s} host enqueue {host:.3f} ms/step   wall {total:.3f} ms/step", flush=True)
dist.destroy_process_group()
