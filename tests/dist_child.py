"""Child process of tests/test_gpu_dist.py (not a test module): one rank of a torch.distributed group that drives
tests/sharded_harness.py ShardedLloyd with the REAL library (libkmeans_hip.so) on cuda:0 and compares it with the unsharded
loop computed in the same process.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment;  argv: backend (nccl | gloo)

backend nccl, world 1: every collective of sharded.py goes through RCCL (force_collectives) -- the asynchronous k x 4
all-reduce beside the label pass, the histogram all-reduce and the in-place all-gather on the tensors that alias the
library's label tables.  backend gloo, world 2+: the ranks share one GPU (RCCL refuses two ranks on a device; gloo stages
device tensors through the host), so the data really crosses ranks: row bands, and cells=True with its all-gather
(world 2) / per-owner broadcasts (world 3).  Exit code 0 = every comparison held."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend = sys.argv[1]
    import numpy as np
    import torch
    import torch.distributed as dist
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    from sharded_harness import ShardedLloyd, band_rows
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    kg.set_strategy("table")
    st = torch.cuda.current_stream().cuda_stream
    w, h, k, iters = 1024, 768, 24, 3
    n = w * h
    img = synth.uniform_rgba_torch(0xD157, n, device="cuda")
    proc = kg.ImageProcessor(shrink_max_dim=0)
    sel = img[(torch.arange(k, device="cuda") * (n // k))].contiguous()
    lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
    proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
    torch.cuda.synchronize()
    cent0 = np.ones((k, 4), np.float32)
    cent0[:, :3] = lab.cpu().numpy()

    # the unsharded loop
    ref = kg.Lloyd(proc, k)
    ref.set_centroids(cent0, st)
    assert ref.prepare(img.data_ptr(), n, True, st) == "table"
    want_labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    ref.assign_accumulate(img.data_ptr(), n, want_labels.data_ptr(), acc.data_ptr(), st)
    for _ in range(iters):
        ref.update(acc.data_ptr(), st)
        ref.assign_accumulate(img.data_ptr(), n, want_labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    want_cent, want_acc = ref.get_centroids(st), acc.clone()
    ref.close()

    r0, r1 = band_rows(h, rank, world)
    band = img[r0 * w:r1 * w].contiguous()
    failures = []
    for cells in (False, True):
        be = kg.Lloyd(proc, k)
        be.set_centroids(cent0, st)
        labels = torch.zeros((r1 - r0) * w, dtype=torch.int32, device="cuda")
        if not cells:
            assert be.prepare(band.data_ptr(), band.shape[0], True, st) == "table"
        sh = ShardedLloyd(be, k, band, labels, stream=st, cells=cells, force_collectives=(world == 1))
        sh.split_labels = True
        if cells:
            sh.bind_cells()
        sh.prime()
        for _ in range(iters):
            sh.iterate()
        sh.flush()
        torch.cuda.synchronize()
        name = "cells" if cells else "bands"
        if not torch.equal(labels, want_labels[r0 * w:r1 * w]):
            failures.append(f"{name}: labels differ")
        if not torch.equal(sh.acc, want_acc):
            failures.append(f"{name}: sums differ")
        if not np.array_equal(be.get_centroids(st).view(np.uint32), want_cent.view(np.uint32)):
            failures.append(f"{name}: centroids differ")
        sh.close()
        be.close()
    proc.close()
    dist.barrier()
    dist.destroy_process_group()
    if failures:
        print(f"rank {rank}/{world} ({backend}): " + "; ".join(failures), flush=True)
        sys.exit(1)
    print(f"rank {rank}/{world} ({backend}): row bands and cell-sharded loop equal the unsharded loop", flush=True)


if __name__ == "__main__":
    main()
