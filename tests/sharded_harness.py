"""TEST HARNESS (since round 6; rounds 1-4: kmeans_gpu_amd/sharded.py, the product's multi-GPU layer -- superseded by the C ABI's
kmg_group_*, csrc/kmg_group.hip).  What is left of it serves the CPU tests: tests/test_sharded_gloo.py drives these classes with an
oracle-backed stand-in over gloo (world 2 and 3) -- the data flow of row bands, cell shares, batches and the sharded initialisation --
and tests/dist_child.py / test_gpu_table.py reuse band_rows / cell_range / ShardedBatch around the real kernels.

Row-band data-parallel Lloyd loop: one process per GPU, torch.distributed for the single exchange step of the path.

The reference has no multi-device path.  Sharding rule: rank g of G owns the
image rows [g*H//G, (g+1)*H//G).  Assignment and the per-pixel half of the update are independent
per pixel; the only coupling is the k x 4 int64 accumulator table (sum qL, sum qa, sum qb, count),
which is all-reduced (sum) once per iteration.  Because the sums are exact integers the result is
bit-identical for any G, and every rank then performs the same centroid update locally -- no
broadcast is needed.  The Bayer index of the dither pass uses image coordinates, so the output
pass needs no exchange at all (kmg_dev_apply takes the band's first row).
"""
import torch
import torch.distributed as dist

__all__ = ["band_rows", "images_of_rank", "cell_range", "sharded_init", "ShardedLloyd", "ShardedBatch", "PlacedBatch"]

CELLS = 32768             # 8x8x8 colour cells of the cube (kmg_table.h kCells), 512 colours each
CELL_COLOURS = 512


def _require_current_stream(tensor, stream):
    """The library kernels run on the raw `stream`; the torch ops of this module (fill_, zero_, all_reduce,
    Work.wait) order against torch's CURRENT stream of the tensor's device.  The two must be the same
    stream, otherwise the collective races with the kernels that produce / consume the accumulators
    (torch side streams are non-blocking: the null stream does not order against them either)."""
    if not tensor.is_cuda:
        return
    current = torch.cuda.current_stream(tensor.device).cuda_stream
    if int(stream or 0) != int(current):
        raise ValueError(f"stream {int(stream or 0):#x} is not torch's current stream {int(current):#x} of {tensor.device}: "
                         "pass stream=torch.cuda.current_stream().cuda_stream (or enter torch.cuda.stream(...) first)")


def band_rows(height, rank, world):
    """Rows [r0, r1) owned by `rank`."""
    return (rank * height) // world, ((rank + 1) * height) // world


def cell_range(rank, world):
    """cells [c0, c1) whose colours `rank` labels in a cell-sharded loop (kmg_lloyd_set_cell_share uses the same rule)"""
    return (CELLS * int(rank)) // int(world), (CELLS * (int(rank) + 1)) // int(world)


def images_of_rank(n_images, rank, world):
    """Whole-image placement of a batch (BASELINE config 4 when the batch is at least as large as the
    node): image i lives on rank i % world."""
    return list(range(int(rank), int(n_images), int(world)))


def sharded_init(backend, k, band, width, height, row0, group=None, stream=0):
    """PlusPlusInitModule::compute (modules.rs:946-1246) for an image sharded in row bands.

    Every rank runs the local pass of its band; the arg-max is found with a MAX all-reduce of a 64-bit
    key (distance bits | image-wide pixel position under the reference's tie rule), the winning pixel's
    colour reaches all ranks with a SUM all-reduce of {colour, 1}, and every rank sets the same centroid.
    2 tiny collectives per centroid, no host synchronisation.  Identical to the unsharded init.

    backend : object with init_step / init_pick_band / set_centroid_rgba / init_first_key (kmeans_gpu_amd.Lloyd)
    band    : this rank's rows, uint8 tensor (rows*width, 4); row0 = image row of its first pixel
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    _require_current_stream(band, stream)
    n_local = int(band.shape[0]) if band.dim() == 2 else int(band.numel() // 4)
    first = int(row0) * int(width)
    ptr = band.data_ptr() if n_local else 0
    key = torch.zeros(1, dtype=torch.int64, device=band.device)
    colour = torch.zeros(2, dtype=torch.int32, device=band.device)

    def publish(j):
        backend.init_pick_band(ptr, n_local, first, key.data_ptr(), colour.data_ptr(), stream)
        if world > 1:
            dist.all_reduce(colour, op=dist.ReduceOp.SUM, group=group)
        backend.set_centroid_rgba(j, colour.data_ptr(), stream)

    key.fill_(backend.init_first_key(width, height))          # plus_plus_init.wgsl:161-168 `initial`
    publish(0)
    for j in range(1, int(k)):
        backend.init_step(ptr, n_local, first, j, key.data_ptr(), stream)
        if world > 1:
            dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
        publish(j)


# Compute units the label pass can leave to the collective (ShardedLloyd(reserve_cus=...)).  Opt-in: with one rank the
# reservation costs 5 us per iteration and buys nothing (profiles/r02_dist_overhead.txt); beyond one rank it has never
# been measured.
RESERVED_CUS = 0


class ShardedLloyd:
    """Drives one `Lloyd`-like backend per rank.

    backend  : object with assign_accumulate(d_rgba, n, d_labels, d_acc, stream), update(d_acc, stream),
               converged_count(stream)  (kmeans_gpu_amd.Lloyd on a GPU)
    rgba     : this rank's band, uint8 tensor (rows*width, 4) on the backend's device
    labels   : int32 tensor (rows*width,) or None
    """

    def __init__(self, backend, k, rgba, labels=None, group=None, stream=0, collective=None, local_only=False,
                 reserve_cus=RESERVED_CUS, cells=False, cell_hooks=None, rank=None, force_collectives=False):
        self.backend = backend
        self.k = int(k)
        self.rgba = rgba
        self.labels = labels
        self.n_local = int(rgba.shape[0]) if rgba.dim() == 2 else int(rgba.numel() // 4)
        self.group = group
        self.stream = stream
        self.acc = torch.zeros((self.k, 4), dtype=torch.int64, device=rgba.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if local_only:
            self.world = 1            # the whole image is here (PlacedBatch): no exchange step at all
        # collective(acc) -> None replaces the SUM all-reduce (tests: the other ranks' share computed locally)
        self.collective = collective
        if collective is not None:
            self.world = max(self.world, 2)
        # True when the backend produces labels with a separate pass (kmeans_gpu_amd.Lloyd after
        # prepare() chose the colour table): lets the collective overlap that pass
        self.split_labels = False
        # label pass of iteration t beside the cube pass of iteration t + 1 (Lloyd.iterate); False (the measured path:
        # the two passes time-slice a CU rather than overlap, profiles/r03_overlap_shapes.txt): step by step
        self.pipeline = False
        # One rank, colour table: prime() / iterate() as "assign, then update" (Lloyd.assign_update: the update rides on the
        # last launch of the assign pass -- no k_update launch, no memset).  Each call still performs one assignment with its
        # label map and sums and one centroid update, but a convergence count read between two calls belongs to the update
        # that FOLLOWED the last assignment; run() therefore never uses it.  Opt-in.
        self.fused = False
        # The asynchronous all-reduce is issued to run beside the label pass, but RCCL's kernel (256 threads, 20 KiB LDS,
        # 280 registers per lane) does not fit on a CU that hosts a label workgroup; reserve_cus > 0 launches the label pass
        # with that many workgroups fewer than CUs.  Reset in close().
        # CELL-SHARDED cube pass (strong scaling of one image; colour-table strategy, k <= 256): row bands alone leave the cube
        # pass -- which works on the image's COLOURS -- at full size on every rank.  With cells=True every rank binds the whole
        # image's colour histogram (its band's, all-reduced once), labels one share of the colour cube per iteration
        # (cell_range) and receives the other ranks' shares of the label tables by all-gather before it writes its band's
        # label map; the sums stay the k x 4 all-reduce.  Same labels and centroids as the unsharded loop, bit for bit.
        # Measured per-rank on one GPU (tools/strong_cells_per_rank.py): 1.70 / 2.54 / 3.71x at N = 2 / 4 / 8 against
        # 1.35 / 1.60 / 1.83x for row bands alone.
        # cell_hooks (tests, one process): object with reduce_count(t), reduce_histogram(t), gather_tables(lab_t, ent_t),
        # world, rank.
        # force_collectives (tests): issue every collective even in a group of ONE rank, so that a single GPU drives the
        # RCCL calls of this module -- on the tensors that alias the library's tables -- exactly as a multi-rank job does.
        self.force_collectives = bool(force_collectives) and dist.is_initialized()
        self.cells = bool(cells)
        self.cell_hooks = cell_hooks
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        if cell_hooks is not None:
            self.world = max(self.world, int(cell_hooks.world))
            self.rank = int(cell_hooks.rank)
        self._cells_bound = False
        self._reserved = 0
        if (reserve_cus and dist.is_initialized() and self.world > 1 and collective is None
                and hasattr(backend, "reserve_cus")):
            backend.reserve_cus(int(reserve_cus))
            self._reserved = int(reserve_cus)

    def bind_cells(self, n_total=None):
        """once per image (cells=True): the band's histogram, all-reduced into the image's; this rank's share of the cube"""
        be = self.backend
        _require_current_stream(self.acc, self.stream)
        collectives = self.world > 1 or self.force_collectives
        if self.n_local:
            be.bind_image(self.rgba.data_ptr(), self.n_local, self.stream)
        else:
            # a rank without rows still takes its share of the cube: the table of ONE dummy pixel, its count taken out again
            if not hasattr(self, "_dummy"):
                self._dummy = torch.zeros((1, 4), dtype=torch.uint8, device=self.acc.device)
            be.bind_image(self._dummy.data_ptr(), 1, self.stream)
        hist = be.histogram_tensor()
        if not self.n_local:
            hist.zero_()
        if n_total is None:
            t = torch.tensor([self.n_local], dtype=torch.int64, device=self.acc.device)
            if self.cell_hooks is not None:
                self.cell_hooks.reduce_count(t)
            elif collectives:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            n_total = int(t.item())
        if self.cell_hooks is not None:
            self.cell_hooks.reduce_histogram(hist)
        elif collectives:
            dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=self.group)
        be.rebuild_from_histogram(n_total, self.stream)
        be.set_cell_share(self.rank, self.world, self.stream)
        self._lab_t, self._ent_t = be.table_tensors()
        self._cells_bound = True

    def _gather_tables(self):
        """every rank's share of the per-colour labels and cell entries -> all ranks (in place)"""
        if self.cell_hooks is not None:
            self.cell_hooks.gather_tables(self._lab_t, self._ent_t)
            return
        if self.world == 1 and not self.force_collectives:
            return
        ranges = [cell_range(r, self.world) for r in range(self.world)]
        for t, per_cell in ((self._lab_t, CELL_COLOURS), (self._ent_t, 1)):
            views = [t[c0 * per_cell:c1 * per_cell] for c0, c1 in ranges]
            if CELLS % self.world == 0:
                dist.all_gather(views, views[self.rank], group=self.group)
            else:                                   # unequal shares: one broadcast per owner
                for r, v in enumerate(views):
                    dist.broadcast(v, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)

    def rebind(self, rgba=None, labels=None):
        """new pixels (or a new buffer) for this loop: the cell-sharded binding and the aliased table tensors are stale"""
        if rgba is not None:
            self.rgba = rgba
            self.n_local = int(rgba.shape[0]) if rgba.dim() == 2 else int(rgba.numel() // 4)
        if labels is not None:
            self.labels = labels
        self._cells_bound = False
        self._lab_t = self._ent_t = None

    def _pass_cells(self):
        lab_ptr = self.labels.data_ptr() if self.labels is not None else 0
        _require_current_stream(self.acc, self.stream)
        if not self._cells_bound:
            self.bind_cells()
        # (a rank without rows is bound to a dummy pixel: the pass walks its share of the image's colour table all the same)
        bound_ptr = self.rgba.data_ptr() if self.n_local else self._dummy.data_ptr()
        self.backend.assign_accumulate(bound_ptr, max(self.n_local, 1), 0, self.acc.data_ptr(), self.stream)
        work = self.exchange(async_op=True)
        if lab_ptr or self.world > 1 or self.force_collectives:
            self._gather_tables()
        if lab_ptr and self.n_local:
            self.backend.labels_from_tables(self.rgba.data_ptr(), self.n_local, lab_ptr, self.stream)
        if work is not None:
            work.wait()

    def _pass(self):
        """labels + sums of the current centroids, and the exchange of the sums.

        With the colour-table strategy (`split_labels`) the sums come from the cube pass and the
        label map from a separate gather pass that does not feed the collective: the all-reduce is
        issued asynchronously right after the sums and overlaps the label pass."""
        if self.cells:
            return self._pass_cells()
        lab_ptr = self.labels.data_ptr() if self.labels is not None else 0
        _require_current_stream(self.acc, self.stream)
        if self.n_local == 0:
            self.acc.zero_()
            self.exchange()
            return
        if self.split_labels and lab_ptr:
            self.backend.assign_accumulate(self.rgba.data_ptr(), self.n_local, 0, self.acc.data_ptr(), self.stream)
            work = self.exchange(async_op=True)
            self.backend.labels(self.rgba.data_ptr(), self.n_local, lab_ptr, self.stream)
            if work is not None:
                work.wait()           # the compute stream waits for the collective, not the host
        else:
            self.backend.assign_accumulate(self.rgba.data_ptr(), self.n_local, lab_ptr,
                                           self.acc.data_ptr(), self.stream)
            self.exchange()

    def exchange(self, async_op=False):
        """the path's one collective: sum of the k x 4 int64 accumulators over all bands"""
        if self.collective is not None:
            self.collective(self.acc)
            return None
        if self.world > 1 or self.force_collectives:
            return dist.all_reduce(self.acc, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return None

    def _pipelined(self):
        """kmeans_gpu_amd.Lloyd.iterate: the label pass of iteration t runs beside the update + cube pass of
        iteration t + 1 on the library's own stream (colour-table strategy with a label map)"""
        return (self.pipeline and not self.cells and self.split_labels and self.labels is not None and self.n_local > 0
                and hasattr(self.backend, "iterate"))

    def _fused(self):
        return (self.fused and self.world == 1 and not self.force_collectives and not self.cells and self.split_labels and self.labels is not None
                and self.n_local > 0 and hasattr(self.backend, "assign_update") and not self._pipelined())

    def _assign_then_update(self):
        _require_current_stream(self.acc, self.stream)
        self.backend.assign_update(self.rgba.data_ptr(), self.n_local, self.labels.data_ptr(), self.acc.data_ptr(), True,
                                   self.stream)

    def prime(self):
        """initial assignment (operations.rs:75-83), fused with the sums of the first update"""
        if self._fused():
            self._assign_then_update()
        elif self._pipelined():
            _require_current_stream(self.acc, self.stream)
            self.backend.iterate(self.rgba.data_ptr(), self.n_local, self.labels.data_ptr(), self.acc.data_ptr(), False, self.stream)
            self.exchange()
        else:
            self._pass()

    def iterate(self):
        """one Lloyd iteration (modules.rs:769-800): update from the global sums, re-assign"""
        if self._fused():
            self._assign_then_update()
        elif self._pipelined():
            _require_current_stream(self.acc, self.stream)
            self.backend.iterate(self.rgba.data_ptr(), self.n_local, self.labels.data_ptr(), self.acc.data_ptr(), True, self.stream)
            self.exchange()
        else:
            self.backend.update(self.acc.data_ptr(), self.stream)
            self._pass()

    def flush(self):
        """the label map of the last iteration is complete once the stream has passed this point"""
        if hasattr(self.backend, "flush"):
            self.backend.flush(self.stream)

    def close(self):
        """hand the backend back as it was found (a reservation of compute units is the loop's, not the backend's)"""
        self.flush()
        if self._reserved:
            self.backend.reserve_cus(0)
            self._reserved = 0

    def run(self, max_iterations=128, check_period=8):
        """ChooseCentroidModule::compute (modules.rs:763-840) over all bands.  Returns the
        iteration at which the loop stopped."""
        self.fused = False        # the loop reads the convergence count between update and re-assignment
        self.prime()
        it = 0
        for it in range(max_iterations):
            self.iterate()
            if it > 0 and it % check_period == 0:
                # identical on every rank: all ranks updated from the same global sums
                if self.backend.converged_count(self.stream) >= self.k:
                    break
        self.close()
        return it


class ShardedBatch:
    """A batch of images, each tiled over all ranks (BASELINE config 4: 16 images over 8 GPUs).

    Every image is an independent k-means problem (its own backend per rank, same k); the
    accumulators of the whole batch live in ONE (images, k, 4) int64 tensor, so an iteration costs a
    single all-reduce of images*k*32 bytes instead of one per image.  An image that has converged (at
    one of its every-`check_period` checks, exactly like the single-image loop) stops being updated.

    backends : list of Lloyd-like objects, one per image (centroids already set)
    bands    : list of uint8 tensors, this rank's band of each image
    labels   : list of int32 tensors (or None entries)
    """

    def __init__(self, backends, k, bands, labels=None, group=None, stream=0, collective=None):
        assert len(backends) == len(bands)
        # collective(acc, active) -> None replaces the SUM all-reduce (tests: the other ranks' share computed locally)
        self.collective = collective
        self.backends = list(backends)
        self.k = int(k)
        self.bands = list(bands)
        self.labels = list(labels) if labels is not None else [None] * len(bands)
        self.group = group
        self.stream = stream
        self.n_local = [int(b.shape[0]) if b.dim() == 2 else int(b.numel() // 4) for b in self.bands]
        self.acc = torch.zeros((len(bands), self.k, 4), dtype=torch.int64, device=self.bands[0].device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = [True] * len(bands)
        self.iterations = [0] * len(bands)

    def _pass(self):
        _require_current_stream(self.acc, self.stream)
        for i, be in enumerate(self.backends):
            if not self.active[i]:
                continue
            lab = self.labels[i].data_ptr() if self.labels[i] is not None else 0
            if self.n_local[i]:
                be.assign_accumulate(self.bands[i].data_ptr(), self.n_local[i], lab, self.acc[i].data_ptr(), self.stream)
            else:
                self.acc[i].zero_()
        if self.collective is not None:
            self.collective(self.acc, list(self.active))
        elif self.world > 1:
            # converged images keep their (stale, unused) rows: shapes stay fixed for the collective
            dist.all_reduce(self.acc, op=dist.ReduceOp.SUM, group=self.group)

    def run(self, max_iterations=128, check_period=8):
        self._pass()
        for it in range(max_iterations):
            if not any(self.active):
                break
            for i, be in enumerate(self.backends):
                if self.active[i]:
                    be.update(self.acc[i].data_ptr(), self.stream)
                    self.iterations[i] = it
            self._pass()
            if it > 0 and it % check_period == 0:
                for i, be in enumerate(self.backends):
                    if self.active[i] and be.converged_count(self.stream) >= self.k:
                        self.active[i] = False
                        self.acc[i].zero_()      # its rows stay in the collective but carry nothing
        return list(self.iterations)


class PlacedBatch:
    """A batch of images with WHOLE images per rank (images_of_rank): every image is an independent k-means
    problem that never leaves its GPU, so the batch needs no collective at all -- the right split whenever the
    batch has at least as many images as the node has GPUs (BASELINE config 4: 16 images over 8 GPUs = 2 each).
    Tiling every image over all ranks instead (ShardedBatch) would repeat the colour-table cube pass, whose cost
    does not depend on the number of pixels, once per image on EVERY rank and add an all-reduce per iteration.

    backends : Lloyd-like objects of THIS rank's images (centroids already set), images / labels likewise
    """

    def __init__(self, backends, k, images, labels=None, stream=0, split_labels=None):
        assert len(backends) == len(images)
        labels = list(labels) if labels is not None else [None] * len(images)
        self.loops = [ShardedLloyd(be, k, img, lab, stream=stream, local_only=True)
                      for be, img, lab in zip(backends, images, labels)]
        for i, loop in enumerate(self.loops):
            loop.split_labels = bool(split_labels[i]) if split_labels is not None else False

    def run(self, max_iterations=128, check_period=8):
        """every image to its own convergence (modules.rs:763-840); returns the iteration each one stopped at"""
        return [loop.run(max_iterations, check_period) for loop in self.loops]
