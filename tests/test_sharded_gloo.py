"""The N > 1 path on CPU: two gloo ranks drive kmeans_gpu_amd.sharded.ShardedLloyd (the host logic
bench.py and a multi-GPU caller use) with an oracle-backed stand-in for the per-GPU kernels, and must
reproduce the unsharded oracle bit-for-bit (integer accumulators make the all-reduce exact)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleBackend:
    """CPU stand-in with the kmeans_gpu_amd.Lloyd interface (tests only)."""

    def __init__(self, O, k, centroids4):
        self.O, self.k = O, k
        self.cent = O.centroids4(centroids4).copy()
        self.nconv = 0

    @staticmethod
    def _view(ptr, shape, dtype):
        import ctypes
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (ctypes.c_uint8 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def assign_accumulate(self, d_rgba, n, d_labels, d_acc, stream=0):
        px = self._view(d_rgba, (n, 4), np.uint8)
        labels, acc = self.O.assign_accumulate_rgba(px, self.cent)
        if d_labels:
            self._view(d_labels, (n,), np.uint32)[:] = labels
        self._view(d_acc, (self.k, 4), np.int64)[:] = acc

    def labels(self, d_rgba, n, d_labels, stream=0):
        px = self._view(d_rgba, (n, 4), np.uint8)
        self._view(d_labels, (n,), np.uint32)[:] = self.O.assign(self.O.rgb_to_lab(px), self.cent)

    def update(self, d_acc, stream=0):
        acc = self._view(d_acc, (self.k, 4), np.int64)
        self.cent, self.nconv = self.O.finalize(acc, self.cent, 1.0)

    def converged_count(self, stream=0):
        return self.nconv


def _worker(rank, world, port, q, split=False):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from kmeans_gpu_amd.sharded import ShardedLloyd, band_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.set_num_threads(2)
        w, h, k = 96, 61, 7
        img = O.synth_uniform(4242, w * h).reshape(h, w, 4)
        lab = O.rgb_to_lab(img)
        init = O.init_centroids(lab, w, h, k)
        r0, r1 = band_rows(h, rank, world)
        band = torch.from_numpy(np.ascontiguousarray(img[r0:r1]).reshape(-1, 4))
        labels = torch.zeros((r1 - r0) * w, dtype=torch.int32)
        be = OracleBackend(O, k, init)
        sh = ShardedLloyd(be, k, band, labels)
        sh.split_labels = split          # sums first, async all-reduce, labels from a separate pass
        it = sh.run(128, 8)
        q.put((rank, it, be.cent.copy(), labels.numpy().view(np.uint32).copy(), (r0, r1)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,split", [(2, False), (3, False), (2, True)])
def test_sharded_lloyd_equals_unsharded(oracle, world, split):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w, h, k = 96, 61, 7
    img = oracle.synth_uniform(4242, w * h).reshape(h, w, 4)
    lab = oracle.rgb_to_lab(img)
    want_c, want_labels, want_it = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
    for rank, it, cent, labels, (r0, r1) in res:
        assert it == want_it
        assert np.array_equal(cent.view(np.uint32), want_c.view(np.uint32))
        assert np.array_equal(labels, want_labels[r0 * w:r1 * w])
    assert res[0][4][0] == 0 and res[-1][4][1] == h


def test_band_rows_partition():
    from kmeans_gpu_amd.sharded import band_rows
    for h in (1, 7, 513, 8192):
        for g in (1, 2, 3, 8):
            rows = [band_rows(h, r, g) for r in range(g)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(rows[i][1] == rows[i + 1][0] for i in range(g - 1))


def _batch_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from kmeans_gpu_amd.sharded import ShardedBatch, band_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.set_num_threads(2)
        k, shapes = 5, [(64, 40), (33, 57), (80, 21)]
        backends, bands, labels = [], [], []
        for j, (w, h) in enumerate(shapes):
            img = O.synth_uniform(900 + j, w * h).reshape(h, w, 4)
            lab = O.rgb_to_lab(img)
            r0, r1 = band_rows(h, rank, world)
            backends.append(OracleBackend(O, k, O.init_centroids(lab, w, h, k)))
            bands.append(torch.from_numpy(np.ascontiguousarray(img[r0:r1]).reshape(-1, 4)))
            labels.append(torch.zeros((r1 - r0) * w, dtype=torch.int32))
        sb = ShardedBatch(backends, k, bands, labels)
        its = sb.run(128, 8)
        q.put((rank, its, [b.cent.copy() for b in backends], [l.numpy().view(np.uint32).copy() for l in labels]))
    finally:
        dist.destroy_process_group()


def test_sharded_batch_equals_per_image_oracle(oracle):
    """BASELINE config 4 in miniature: 3 images tiled over 2 ranks, one all-reduce per iteration"""
    from kmeans_gpu_amd.sharded import band_rows
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    k, shapes = 5, [(64, 40), (33, 57), (80, 21)]
    for j, (w, h) in enumerate(shapes):
        img = oracle.synth_uniform(900 + j, w * h).reshape(h, w, 4)
        lab = oracle.rgb_to_lab(img)
        want_c, want_labels, want_it = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
        for rank, its, cents, labs in res:
            r0, r1 = band_rows(h, rank, world)
            assert its[j] == want_it
            assert np.array_equal(cents[j].view(np.uint32), want_c.view(np.uint32))
            assert np.array_equal(labs[j], want_labels[r0 * w:r1 * w])
