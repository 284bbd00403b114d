"""The N > 1 path on CPU: two gloo ranks drive tests/sharded_harness.py ShardedLloyd (the host logic
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
    from sharded_harness import ShardedLloyd, band_rows
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
    from sharded_harness import band_rows
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
    from sharded_harness import ShardedBatch, band_rows
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
    from sharded_harness import band_rows
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


def _placed_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from sharded_harness import PlacedBatch, images_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.set_num_threads(2)

        def no_collective(*a, **kw):
            raise AssertionError("whole-image placement must not communicate")
        dist.all_reduce = no_collective
        k, shapes = 5, [(64, 40), (33, 57), (80, 21), (17, 90), (50, 50)]
        mine = images_of_rank(len(shapes), rank, world)
        backends, images, labels = [], [], []
        for j in mine:
            w, h = shapes[j]
            img = O.synth_uniform(700 + j, w * h).reshape(h, w, 4)
            backends.append(OracleBackend(O, k, O.init_centroids(O.rgb_to_lab(img), w, h, k)))
            images.append(torch.from_numpy(np.ascontiguousarray(img).reshape(-1, 4)))
            labels.append(torch.zeros(w * h, dtype=torch.int32))
        its = PlacedBatch(backends, k, images, labels).run(128, 8)
        q.put((rank, mine, its, [b.cent.copy() for b in backends], [l.numpy().view(np.uint32).copy() for l in labels]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_placed_batch_whole_images_no_collective(oracle, world):
    """BASELINE config 4 as shipped when the batch is at least as large as the node: image i on rank
    i % world, every image an independent loop, NO collective -- results equal the per-image oracle"""
    from sharded_harness import images_of_rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_placed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    k, shapes = 5, [(64, 40), (33, 57), (80, 21), (17, 90), (50, 50)]
    seen = []
    for rank, mine, its, cents, labs in res:
        assert mine == images_of_rank(len(shapes), rank, world)
        for slot, j in enumerate(mine):
            w, h = shapes[j]
            img = oracle.synth_uniform(700 + j, w * h).reshape(h, w, 4)
            lab = oracle.rgb_to_lab(img)
            want_c, want_labels, want_it = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
            assert its[slot] == want_it
            assert np.array_equal(cents[slot].view(np.uint32), want_c.view(np.uint32))
            assert np.array_equal(labs[slot], want_labels)
            seen.append(j)
    assert sorted(seen) == list(range(len(shapes)))


class OracleInitBackend(OracleBackend):
    """adds the sharded-init steps (kmg_lloyd_init_step / _init_pick_band / _set_centroid_rgba)"""

    @staticmethod
    def _pack(g):
        return ((g >> 4) << 4) | (15 - (g & 15))

    @staticmethod
    def init_first_key(width, height):
        x0 = int(np.float32(width) * np.float32(0.5625)); y0 = int(np.float32(height) * np.float32(0.93359375))
        return (1 << 32) | OracleInitBackend._pack(y0 * width + x0)

    def init_step(self, d_rgba, n, first, j, d_key, stream=0):
        key = self._view(d_key, (1,), np.int64)
        key[0] = 0
        if n == 0:
            return
        px = self._view(d_rgba, (n, 4), np.uint8)
        lab = self.O.rgb_to_lab(px)
        d = np.array([self.O.cie94(lab[i], self.cent[j - 1, :3]) for i in range(n)], np.float32)
        self.dist = np.minimum(np.float32(1000000.0) if j == 1 else self.dist, d)
        bits = self.dist.view(np.uint32).astype(np.int64)
        g = first + np.arange(n, dtype=np.int64)
        key[0] = int(((bits << 32) | (((g >> 4) << 4) | (15 - (g & 15)))).max())

    def init_pick_band(self, d_rgba, n, first, d_key, d_colour2, stream=0):
        kk = int(self._view(d_key, (1,), np.int64)[0])
        idx = 0
        if kk >> 32:
            low = kk & 0xFFFFFFFF
            idx = (low & ~15) | (15 - (low & 15))
        out = self._view(d_colour2, (2,), np.uint32)
        if first <= idx < first + n:
            out[0] = self._view(d_rgba, (n,), np.uint32)[idx - first]; out[1] = 1
        else:
            out[:] = 0

    def set_centroid_rgba(self, j, d_colour, stream=0):
        px = self._view(d_colour, (1,), np.uint32).view(np.uint8).reshape(1, 4)
        self.cent[j, :3] = self.O.rgb_to_lab(px)[0]
        self.cent[j, 3] = 1.0


def _init_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from sharded_harness import ShardedLloyd, band_rows, sharded_init
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.set_num_threads(2)
        w, h, k = 40, 33, 6
        img = O.synth_uniform(77, w * h).reshape(h, w, 4)
        img[5:9] = img[20:24]                      # duplicated rows: exact ties across the bands
        r0, r1 = band_rows(h, rank, world)
        band = torch.from_numpy(np.ascontiguousarray(img[r0:r1]).reshape(-1, 4))
        be = OracleInitBackend(O, k, np.zeros((k, 4), np.float32))
        sharded_init(be, k, band, w, h, r0)
        init = be.cent.copy()
        labels = torch.zeros((r1 - r0) * w, dtype=torch.int32)
        it = ShardedLloyd(be, k, band, labels).run(128, 8)
        q.put((rank, init, it, be.cent.copy()))
    finally:
        dist.destroy_process_group()


def test_sharded_init_then_lloyd_equals_oracle(oracle):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w, h, k = 40, 33, 6
    img = oracle.synth_uniform(77, w * h).reshape(h, w, 4)
    img[5:9] = img[20:24]
    lab = oracle.rgb_to_lab(img)
    want_init = oracle.init_centroids(lab, w, h, k)
    want_c, _, want_it = oracle.lloyd(lab, want_init)
    for rank, init, it, cent in res:
        assert np.array_equal(init.view(np.uint32), want_init.view(np.uint32))
        assert it == want_it and np.array_equal(cent.view(np.uint32), want_c.view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------
# cell-sharded cube pass (ShardedLloyd(cells=True)): the data flow of the GPU loop with an oracle-backed stand-in
# ---------------------------------------------------------------------------------------------------------------
def _colour_index(px):
    """kmg_table.h colour_index: cell-major, sub-cell-major order of the 2^24 colours"""
    r, g, b = (px[:, i].astype(np.uint32) for i in range(3))
    return (((r >> 3) << 19) | ((g >> 3) << 14) | ((b >> 3) << 9) | (((r >> 2) & 1) << 8) | (((g >> 2) & 1) << 7) |
            (((b >> 2) & 1) << 6) | ((r & 3) << 4) | ((g & 3) << 2) | (b & 3))


def _index_to_rgba(idx):
    idx = idx.astype(np.uint32)
    out = np.empty((idx.size, 4), np.uint8)
    out[:, 0] = (((idx >> 19) & 31) << 3) | (((idx >> 8) & 1) << 2) | ((idx >> 4) & 3)
    out[:, 1] = (((idx >> 14) & 31) << 3) | (((idx >> 7) & 1) << 2) | ((idx >> 2) & 3)
    out[:, 2] = (((idx >> 9) & 31) << 3) | (((idx >> 6) & 1) << 2) | (idx & 3)
    out[:, 3] = 255
    return out


class OracleCellBackend(OracleBackend):
    """The cell-sharded interface of kmeans_gpu_amd.Lloyd on the CPU: colour histogram, a share of the cube per rank,
    per-colour label table.  The label of a colour comes from the oracle's arg-min, the sums from the oracle's integer
    accumulation weighted with the histogram's counts."""

    def bind_image(self, d_rgba, n, stream=0):
        px = self._view(d_rgba, (n, 4), np.uint8)
        self.hist = np.bincount(_colour_index(px), minlength=1 << 24).astype(np.int32)
        self.lab_table = np.zeros(1 << 24, np.uint8)
        self.entries = np.zeros(32768, np.int32)

    def histogram_tensor(self):
        return torch.from_numpy(self.hist)

    def rebuild_from_histogram(self, n_total, stream=0):
        assert int(self.hist.sum()) == n_total
        self.colours = np.flatnonzero(self.hist).astype(np.uint32)
        self.share = self.colours

    def set_cell_share(self, part, parts, stream=0):
        c0, c1 = (32768 * part) // parts, (32768 * (part + 1)) // parts
        cell = self.colours >> 9
        self.share = self.colours[(cell >= c0) & (cell < c1)]

    def table_tensors(self):
        return torch.from_numpy(self.lab_table), torch.from_numpy(self.entries)

    def assign_accumulate(self, d_rgba, n, d_labels, d_acc, stream=0):
        assert not d_labels
        acc = np.zeros((self.k, 4), np.int64)
        if self.share.size:
            lab = self.O.rgb_to_lab(_index_to_rgba(self.share))
            labels = self.O.assign(lab, self.cent)
            self.lab_table[self.share] = labels.astype(np.uint8)
            counts = self.hist[self.share]
            acc = self.O.accumulate(np.repeat(lab, counts, axis=0), np.repeat(labels, counts), self.k)
        self._view(d_acc, (self.k, 4), np.int64)[:] = acc

    def labels_from_tables(self, d_rgba, n, d_labels, stream=0):
        px = self._view(d_rgba, (n, 4), np.uint8)
        self._view(d_labels, (n,), np.uint32)[:] = self.lab_table[_colour_index(px)]


def _cells_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from sharded_harness import ShardedLloyd, band_rows
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
        be = OracleCellBackend(O, k, init)
        sh = ShardedLloyd(be, k, band, labels, cells=True)
        it = sh.run(128, 8)
        q.put((rank, it, be.cent.copy(), labels.numpy().view(np.uint32).copy(), (r0, r1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cell_sharded_lloyd_equals_unsharded(oracle, world):
    """ShardedLloyd(cells=True) over gloo: histogram all-reduce at bind, per iteration the k x 4 all-reduce and the
    all-gather (world 2: equal shares) / per-owner broadcasts (world 3: 32768 cells do not divide) of the label tables --
    iteration count, centroids and every rank's band of the label map equal the unsharded oracle bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cells_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
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
