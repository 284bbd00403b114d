"""GPU tests of the colour-table strategy: it must reproduce the per-pixel scan (and therefore the
oracle) bit-for-bit -- labels, int64 sums, centroids, iteration count -- and its interval bounds /
candidate masks must be conservative for every one of the 2^24 colours."""
import numpy as np
import pytest

from conftest import set_strategy as _set_strategy

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream(torch):
    return torch.cuda.current_stream().cuda_stream


def _blobs(rng, n, centres, sigma=12.0):
    c = rng.integers(0, 256, (centres, 3))
    px = c[rng.integers(0, centres, n)] + rng.normal(0, sigma, (n, 3))
    out = np.full((n, 4), 255, np.uint8)
    out[:, :3] = np.clip(np.rint(px), 0, 255).astype(np.uint8)
    return out


def _centroid_sets(oracle, rng):
    """centroid tables that stress the bounds: random Lab of real colours, clustered, grey axis,
    duplicates, far outside the gamut"""
    px = rng.integers(0, 256, (300, 4), dtype=np.uint8)
    lab = oracle.rgb_to_lab(px)
    sets = {"random256": lab[:256], "random16": lab[:16], "single": lab[:1], "k300": lab[:300]}
    grey = np.stack([np.linspace(0, 100, 64), np.zeros(64), np.zeros(64)], 1).astype(np.float32)
    sets["grey64"] = grey
    sets["duplicates"] = np.concatenate([lab[:8], lab[:8], lab[3:4]])
    sets["tight"] = (lab[5] + rng.normal(0, 0.5, (40, 3))).astype(np.float32)
    sets["outside"] = np.array([[150, 0, 0], [-50, 0, 0], [50, 300, -300], [50, 0.001, -0.001]], np.float32)
    # small tables take the dominance test (kmg_cube.hip `dominated`): it is skipped when a component is above 1024 in
    # magnitude ("far"), and must stay conservative right below that ("edge1024")
    sets["far"] = np.array([[5000, 0, 0], [50, 2000, -2000], [1e6, 1e6, 1e6], [50, 0, 0], [60, 10, 10]], np.float32)
    sets["edge1024"] = np.array([[1000, 1000, -1000], [50, 5, 5], [52, 5, 5], [-1000, -1000, 1000], [50, 1023, 0],
                                 [51, -3, 4], [1023, 0, 0]], np.float32)
    # the same two for 32 < k <= 256 (k_cube_one: the test over candidate LISTS)
    sets["far64"] = np.concatenate([lab[:59], sets["far"]])
    sets["edge64"] = np.concatenate([lab[:57], sets["edge1024"]])
    sets["k512"] = oracle.rgb_to_lab(rng.integers(0, 256, (512, 4), dtype=np.uint8))                                   # two byte lists per cell
    sets["crowded_high"] = np.concatenate([lab[:260], (lab[9] + rng.normal(0, 0.3, (90, 3))).astype(np.float32)])        # > 63 in the second
    sets["crowded100"] = np.concatenate([(lab[7] + rng.normal(0, 0.3, (100, 3))).astype(np.float32), lab[100:256]])   # > 63 candidates
    return {k: oracle.centroids4(v) for k, v in sets.items()}


def test_bounds_and_masks_conservative_for_all_colours(torch_cuda, processor, oracle):
    import kmeans_gpu_amd as kg
    rng = np.random.default_rng(11)
    for name, cent in _centroid_sets(oracle, rng).items():
        s = kg.Lloyd(processor, cent.shape[0])
        s.set_centroids(cent)
        assert s.debug_check_table(_stream(torch_cuda)) == (0, 0, 0), name
        s.close()


@pytest.mark.parametrize("kind,n,k", [("uniform", 300_001, 16), ("uniform", 1_000_003, 256), ("uniform", 65_536, 300),
                                      ("blobs", 700_000, 64), ("flat", 100_000, 5), ("tokyo", 0, 8), ("tokyo", 0, 46),
                                      # the edges of the one-launch pass of small centroid tables (k <= 32: KP = 8 / 16 / 32)
                                      ("uniform", 500_000, 1), ("uniform", 500_000, 2), ("uniform", 2_500_000, 9),
                                      ("blobs", 600_000, 32), ("uniform", 600_000, 33)])
def test_table_pass_equals_pixel_scan(torch_cuda, processor, oracle, tokyo, kind, n, k):
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    rng = np.random.default_rng(n + k)
    if kind == "uniform":
        rgba = oracle.synth_uniform(n * 7 + k, n)
    elif kind == "blobs":
        rgba = _blobs(rng, n, 20)
    elif kind == "flat":
        rgba = np.tile(np.array([[12, 200, 77, 255], [12, 200, 78, 255], [0, 0, 0, 255]], np.uint8), (n // 3 + 1, 1))[:n]
    else:
        rgba = tokyo.reshape(-1, 4)
    n = rgba.shape[0]
    lab = oracle.rgb_to_lab(rgba)
    cent = oracle.centroids4(lab[rng.choice(n, k, replace=False)])
    d = _dev(torch, rgba)
    st = _stream(torch)

    def run(bind):
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent)
        if bind:
            s.bind_image(d.data_ptr(), n, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        out = []
        for _ in range(3):                       # three Lloyd iterations
            s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
            torch.cuda.synchronize()
            out.append((labels.cpu().numpy().copy(), acc.cpu().numpy().copy()))
            if bind and k <= 256:                # the LDS pair entries agree with the per-colour labels
                bad, resolved, total = s.debug_check_pairs(st)
                assert bad == 0 and total == n and resolved <= total
            s.update(acc.data_ptr(), st)
        c = s.get_centroids(st)
        s.close()
        return out, c

    brute, c_brute = run(False)
    table, c_table = run(True)
    for (lb, ab), (lt, at) in zip(brute, table):
        assert np.array_equal(lb, lt)
        assert np.array_equal(ab, at)
    assert np.array_equal(c_brute.view(np.uint32), c_table.view(np.uint32))
    # and the first pass against the oracle itself
    wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
    assert np.array_equal(table[0][0].view(np.uint32), wl) and np.array_equal(table[0][1], wa)


@pytest.mark.parametrize("name", ["far", "edge1024", "outside", "duplicates", "far64", "edge64"])
def test_table_pass_with_centroids_far_outside_the_gamut(torch_cuda, processor, oracle, name):
    """tables (the one-launch cube pass of k <= 32 with its dominance test; k = 64: the dominance phase of the general pass) with
    centroids no image produces: labels and sums of the table pass == the per-pixel scan == the oracle
    (find_centroid.wgsl:29-41: first minimum, strict <)"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    cent = _centroid_sets(oracle, np.random.default_rng(11))[name]
    k, n = cent.shape[0], 700_000
    rgba = oracle.synth_uniform(4242 + k, n)
    wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
    d = _dev(torch, rgba)
    st = _stream(torch)
    for strategy in ("scan", "table"):
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent)
        if strategy == "table":
            s.bind_image(d.data_ptr(), n, st)                     # (an explicit binding: the table whatever the cost model says)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl), (name, strategy)
        assert np.array_equal(acc.cpu().numpy(), wa), (name, strategy)
        if strategy == "table":
            bad, resolved, total = s.debug_check_pairs(st)
            assert bad == 0 and total == n
        s.close()


@pytest.mark.parametrize("k", [33, 12])
def test_table_sums_only_and_labels_only(torch_cuda, processor, oracle, k):
    """(k = 12: the one-launch cube pass of small centroid tables -- sums only, labels only through the partial rows, and
    the two-step partial sums)"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    n = 400_000
    rgba = oracle.synth_uniform(99, n)
    cent = oracle.centroids4(oracle.rgb_to_lab(rgba[:k]))
    wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
    d = _dev(torch, rgba)
    st = _stream(torch)
    s = kg.Lloyd(processor, k)
    s.set_centroids(cent)
    s.bind_image(d.data_ptr(), n, st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    s.assign_accumulate(d.data_ptr(), n, 0, acc.data_ptr(), st)
    s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), wa)
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl)
    # the two-step entry points (partial rows, then their reduction) through the table
    acc.zero_(); labels.zero_()
    s.assign_partials(d.data_ptr(), n, labels.data_ptr(), st)
    s.reduce_partials(n, acc.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), wa)
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl)
    # a different buffer (or length) is not the bound image: falls back to the per-pixel scan
    s.assign_accumulate(d.data_ptr(), n - 5, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    wl2, wa2 = oracle.assign_accumulate_rgba(rgba[:n - 5], cent)
    assert np.array_equal(acc.cpu().numpy(), wa2)
    s.unbind_image()
    s.close()


def test_lloyd_run_with_table_matches_oracle(torch_cuda, oracle, monkeypatch):
    """the whole loop (init on device, table strategy forced) == oracle, full-resolution mode"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy("table")
    w, h, k = 640, 480, 12
    img = _blobs(np.random.default_rng(3), w * h, 9).reshape(h, w, 4)
    lab = oracle.rgb_to_lab(img)
    want_c, want_labels, want_it = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
    p = kg.ImageProcessor(shrink_max_dim=0)
    d = _dev(torch, img)
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    s = kg.Lloyd(p, k)
    st = _stream(torch)
    s.init_centroids(d.data_ptr(), w, h, st)
    it = s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
    got_c = s.get_centroids(st)
    assert it == want_it
    assert np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels)
    s.close()
    p.close()


def test_full_size_table_equals_scan(torch_cuda, oracle):
    """BASELINE config 3 size (8192x8192, k=256): colour-table pass == per-pixel scan, every label and
    every accumulator; plus size-independent properties (counts sum to N, labels < k, idempotence)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    n, k = 8192 * 8192, 256
    p = kg.ImageProcessor(shrink_max_dim=0)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    st = _stream(torch)
    cent = oracle.centroids4(oracle.rgb_to_lab(synth.uniform_rgba_numpy(synth.SEED_CFG3, n // 64)[::(n // 64) // k][:k]))
    res = []
    for bind in (False, True):
        s = kg.Lloyd(p, k)
        s.set_centroids(cent)
        if bind:
            s.bind_image(rgba.data_ptr(), n, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        s.update(acc.data_ptr(), st)
        labels2 = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc2 = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(rgba.data_ptr(), n, labels2.data_ptr(), acc2.data_ptr(), st)
        torch.cuda.synchronize()
        res.append((labels, acc, labels2, acc2, s.get_centroids(st)))
        s.close()
    (l0, a0, l0b, a0b, c0), (l1, a1, l1b, a1b, c1) = res
    assert torch.equal(l0, l1) and torch.equal(a0, a1) and torch.equal(l0b, l1b) and torch.equal(a0b, a1b)
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert int(a1[:, 3].sum()) == n and int(a1b[:, 3].sum()) == n
    assert int(l1.max()) < k and int(l1.min()) >= 0
    # counts per label equal the accumulator's counts; sums re-derivable from labels (checksum of checksums)
    cnt = torch.bincount(l1b.to(torch.int64), minlength=k)
    assert torch.equal(cnt, a1b[:, 3])
    # oracle spot check on a slice of the image
    m = 1 << 16
    wl, _ = oracle.assign_accumulate_rgba(rgba[:m].cpu().numpy(), cent)
    assert np.array_equal(l1[:m].cpu().numpy().view(np.uint32), wl)
    p.close()


@pytest.mark.parametrize("strategy", ["brute", "table"])
def test_two_bands_on_one_gpu_equal_unsharded(torch_cuda, oracle, monkeypatch, strategy):
    """The multi-GPU data flow with the real kernels: two row bands (two Lloyd states), their int64
    accumulators summed (what the RCCL all-reduce does), both updated from the sum -- must equal the
    unsharded run and the oracle bit-for-bit, for either strategy."""
    import kmeans_gpu_amd as kg
    from sharded_harness import band_rows
    torch = torch_cuda
    _set_strategy(strategy)
    w, h, k = 512, 301, 24
    img = _blobs(np.random.default_rng(17), w * h, 30).reshape(h, w, 4)
    lab = oracle.rgb_to_lab(img)
    init = oracle.init_centroids(lab, w, h, k)
    want_c, want_labels, want_it = oracle.lloyd(lab, init)
    p = kg.ImageProcessor(shrink_max_dim=0)
    st = _stream(torch)
    d = _dev(torch, img.reshape(-1, 4))
    bands = []
    for r in range(2):
        r0, r1 = band_rows(h, r, 2)
        s = kg.Lloyd(p, k)
        s.set_centroids(init, st)
        n = (r1 - r0) * w
        ptr = d.data_ptr() + r0 * w * 4
        s.prepare(ptr, n, True, st)
        bands.append((s, ptr, n, torch.zeros(n, dtype=torch.int32, device="cuda"),
                      torch.zeros((k, 4), dtype=torch.int64, device="cuda"), r0))

    def assign_all():
        for s, ptr, n, labels, acc, _ in bands:
            s.assign_accumulate(ptr, n, labels.data_ptr(), acc.data_ptr(), st)
        total = bands[0][4] + bands[1][4]                      # the all-reduce
        for b in bands:
            b[4].copy_(total)

    assign_all()
    it = 0
    for it in range(128):
        for s, _, _, _, acc, _ in bands:
            s.update(acc.data_ptr(), st)
        assign_all()
        if it > 0 and it % 8 == 0 and bands[0][0].converged_count(st) >= k:
            break
    assert it == want_it
    for s, _, n, labels, _, r0 in bands:
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32))
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels[r0 * w:r0 * w + n])
        s.close()
    p.close()


def test_table_on_unaligned_band(torch_cuda, processor, oracle):
    """a band whose first pixel is not 16-byte aligned (odd width x odd first row) through the table"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    n, k = 200_003, 9
    rgba = oracle.synth_uniform(31, n + 3)
    cent = oracle.centroids4(oracle.rgb_to_lab(rgba[:k]))
    d = _dev(torch, rgba)
    st = _stream(torch)
    for off in (1, 3):
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent, st)
        labels = torch.zeros(n + 3, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.bind_image(d.data_ptr() + 4 * off, n, st)
        s.assign_accumulate(d.data_ptr() + 4 * off, n, labels.data_ptr() + 4 * off, acc.data_ptr(), st)
        torch.cuda.synchronize()
        wl, wa = oracle.assign_accumulate_rgba(rgba[off:off + n], cent)
        assert np.array_equal(labels.cpu().numpy().view(np.uint32)[off:off + n], wl)
        assert np.array_equal(acc.cpu().numpy(), wa)
        s.close()


def test_cfg3_full_lloyd_both_strategies(torch_cuda, oracle, monkeypatch):
    """BASELINE config 3: synthetic 8192x8192, k=256, reference init at full resolution, Lloyd to
    convergence (or MAX_ITERATION), then the dither pass -- run once per strategy; centroids, label
    maps, iteration counts and the dithered image must be identical, and the size-independent
    properties must hold (labels < k, counts sum to N, output colours are palette colours)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    w = h = 8192
    n, k = w * h, 256
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    st = _stream(torch)
    out = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor(shrink_max_dim=0, max_iterations=24)     # 24 iterations keep the scan run short
        s = kg.Lloyd(p, k)
        s.init_centroids(rgba.data_ptr(), w, h, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        it = s.run(rgba.data_ptr(), n, labels.data_ptr(), st)
        cent = s.get_centroids(st)
        img = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        p.apply(rgba.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, img.data_ptr(), st)
        torch.cuda.synchronize()
        out[strategy] = (it, cent, labels, img)
        s.close()
        p.close()
    (it0, c0, l0, i0), (it1, c1, l1, i1) = out["brute"], out["table"]
    assert it0 == it1
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert torch.equal(l0, l1) and torch.equal(i0, i1)
    assert int(l1.min()) >= 0 and int(l1.max()) < k
    assert int(torch.bincount(l1.to(torch.int64), minlength=k).sum()) == n
    # every dithered pixel is one of the k palette colours (lab_to_rgb of a centroid)
    pal = torch.unique(i1.view(torch.int32))
    assert pal.numel() <= k
    # the first rows against the oracle (init at full resolution is too slow for the CPU: use the
    # device's centroids, which equal the other strategy's, and check the assignment only)
    m = 1 << 15
    wl, _ = oracle.assign_accumulate_rgba(rgba[:m].cpu().numpy(), c1)
    assert np.array_equal(l1[:m].cpu().numpy().view(np.uint32), wl)


@pytest.mark.parametrize("k", [1, 3, 46, 64, 300])
def test_replace_output_pass_table_equals_scan(torch_cuda, oracle, monkeypatch, k):
    """find / reduce in replace mode through the colour table (forced) == per-pixel scan == oracle"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    w, h = 1000, 777
    img = _blobs(np.random.default_rng(k), w * h, 25, sigma=20.0).reshape(h, w, 4)
    pal = np.array(sorted(set(map(tuple, oracle.synth_uniform(k + 5, k)))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    d = _dev(torch, img.reshape(-1, 4))
    st = _stream(torch)
    outs = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor()
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent, kg.ReduceMode.Replace, out.data_ptr(), st)
        torch.cuda.synchronize()
        outs[strategy] = out.cpu().numpy().reshape(h, w, 4)
        p.close()
    assert np.array_equal(outs["brute"], outs["table"])
    assert np.array_equal(outs["table"], oracle.find(img, pal, oracle.MODE_REPLACE))


def test_batch_of_images_on_one_gpu(torch_cuda, processor, oracle):
    """ShardedBatch with the real kernels (one rank): three images, one accumulator tensor"""
    import kmeans_gpu_amd as kg
    from sharded_harness import ShardedBatch
    torch = torch_cuda
    st = _stream(torch)
    k, shapes = 6, [(200, 150), (333, 77), (64, 64)]
    p = kg.ImageProcessor(shrink_max_dim=0)
    backends, bands, labels, wants = [], [], [], []
    for j, (w, h) in enumerate(shapes):
        img = _blobs(np.random.default_rng(50 + j), w * h, 10).reshape(h, w, 4)
        lab = oracle.rgb_to_lab(img)
        init = oracle.init_centroids(lab, w, h, k)
        wants.append(oracle.lloyd(lab, init))
        s = kg.Lloyd(p, k)
        s.set_centroids(init, st)
        backends.append(s)
        bands.append(_dev(torch, img.reshape(-1, 4)))
        labels.append(torch.zeros(w * h, dtype=torch.int32, device="cuda"))
    its = ShardedBatch(backends, k, bands, labels, stream=st).run(128, 8)
    torch.cuda.synchronize()
    for j, (want_c, want_labels, want_it) in enumerate(wants):
        assert its[j] == want_it
        assert np.array_equal(backends[j].get_centroids(st).view(np.uint32), want_c.view(np.uint32))
        assert np.array_equal(labels[j].cpu().numpy().view(np.uint32), want_labels)
        backends[j].close()
    p.close()


@pytest.mark.parametrize("strategy", ["brute", "table"])
@pytest.mark.parametrize("w,h,k,bands", [(256, 171, 8, 2), (67, 41, 6, 3), (300, 200, 33, 4)])
def test_sharded_init_steps_equal_unsharded_init(torch_cuda, processor, oracle, tokyo, monkeypatch, w, h, k, bands, strategy):
    """the band-wise init (local pass + MAX of keys + SUM of {colour, 1}) with the real kernels, the two
    collectives emulated on one GPU, equals the unsharded device init and the oracle -- with the passes
    over the band's pixels ("brute") and over its colours ("table")"""
    import kmeans_gpu_amd as kg
    from sharded_harness import band_rows
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy(strategy)
    if (w, h) == (256, 171):
        img = oracle.resize(tokyo, w, h)
    else:
        rng = np.random.default_rng(w)
        pal = rng.integers(0, 256, (5, 4), dtype=np.uint8); pal[:, 3] = 255
        img = pal[rng.integers(0, 5, (h, w))]                   # few colours: many exact ties
    want = oracle.init_centroids(oracle.rgb_to_lab(img), w, h, k)
    d = _dev(torch, img.reshape(-1, 4))
    states = []
    for r in range(bands):
        r0, r1 = band_rows(h, r, bands)
        states.append((kg.Lloyd(processor, k), d.data_ptr() + r0 * w * 4, (r1 - r0) * w, r0 * w,
                       torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda")))

    def publish(j, key_value):
        total = torch.zeros(2, dtype=torch.int32, device="cuda")
        for s, ptr, n, first, key, colour in states:
            key.copy_(key_value)
            s.init_pick_band(ptr, n, first, key.data_ptr(), colour.data_ptr(), st)
            total += colour                                      # SUM all-reduce
        assert int(total[1]) == 1                                # exactly one band owns the pixel
        for s, *_ in states:
            s.set_centroid_rgba(j, total.data_ptr(), st)

    publish(0, torch.tensor([kg.Lloyd.init_first_key(w, h)], dtype=torch.int64, device="cuda"))
    for j in range(1, k):
        best = None
        for s, ptr, n, first, key, _ in states:
            s.init_step(ptr, n, first, j, key.data_ptr(), st)
            best = key.clone() if best is None else torch.maximum(best, key)   # MAX all-reduce
        publish(j, best)
    for s, *_ in states:
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want.view(np.uint32))
        s.close()


@pytest.mark.parametrize("n,k", [(1, 1), (1, 4), (7, 3), (1000, 16), (4097, 256)])
def test_table_tiny_and_flat_images(torch_cuda, processor, oracle, n, k):
    """degenerate inputs through the forced colour table: a handful of pixels, a single colour"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    for flat in (False, True):
        rgba = oracle.synth_uniform(n + k, n)
        if flat:
            rgba[:] = rgba[0]
        cent = oracle.centroids4(oracle.rgb_to_lab(oracle.synth_uniform(k * 3 + 1, k)))
        wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
        d = _dev(torch, rgba)
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent, st)
        s.bind_image(d.data_ptr(), n, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl)
        assert np.array_equal(acc.cpu().numpy(), wa)
        s.update(acc.data_ptr(), st)
        want_c, _ = oracle.finalize(wa, cent)
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32))
        s.close()


def test_dither_masks_conservative_for_all_colours(torch_cuda, processor, oracle):
    """pruned dither pass: over all 2^24 colours x 16 Bayer offsets the arg-min over the candidates of the
    pixel's (cell, Bayer index) equals the brute-force arg-min (sentinel start included)"""
    rng = np.random.default_rng(12)
    for name, cent in _centroid_sets(oracle, rng).items():
        if cent.shape[0] < 2:
            continue
        assert processor.debug_check_dither_masks(cent, _stream(torch_cuda)) == 0, name


@pytest.mark.parametrize("k", [2, 3, 46, 64, 65, 256, 257, 300, 512, 600])
def test_dither_output_pass_pruned_equals_scan(torch_cuda, oracle, monkeypatch, k):
    """find / reduce in dither mode: candidate-pruned pass (forced) == scan of all centroids == oracle,
    for a whole image and for a row band that starts at an odd image row"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    w, h = 1001, 700
    img = _blobs(np.random.default_rng(100 + k), w * h, 25, sigma=25.0).reshape(h, w, 4)
    pal = np.array(sorted(set(map(tuple, oracle.synth_uniform(k + 9, k)))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    d = _dev(torch, img.reshape(-1, 4))
    st = _stream(torch)
    want = oracle.find(img, pal, oracle.MODE_DITHER)
    r0, r1 = 333, 512
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor()
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
        band = torch.zeros(((r1 - r0) * w, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr() + 4 * r0 * w, w, r1 - r0, r0, cent, kg.ReduceMode.Dither, band.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(h, w, 4), want), strategy
        assert np.array_equal(band.cpu().numpy().reshape(r1 - r0, w, 4), want[r0:r1]), strategy
        p.close()


@pytest.mark.parametrize("k,mode", [(46, 0), (64, 0), (65, 0), (200, 0), (300, 0), (33, 2), (64, 2)])
def test_mask_word_output_passes_for_small_k_still_equal_the_oracle(torch_cuda, oracle, monkeypatch, k, mode):
    """k <= 256 takes the lists over Lab cells by default; the mask words per (RGB cell, Bayer index) -- the path of k > 256,
    with its sorted (k <= 64) and multi-word kernels -- stay selectable (KMG_DITHER_LISTS=0) and stay right: dither and meld"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy("table")
    _set_strategy("table+mask_words" if mode == 0 else "table")
    w, h = 777, 400
    img = np.concatenate([_blobs(np.random.default_rng(k), w * 200, 20, sigma=20.0), oracle.synth_uniform(k, w * 200)]).reshape(h, w, 4)
    pal = np.array(sorted(set(map(tuple, oracle.synth_uniform(k + 3, k)))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    p = kg.ImageProcessor()
    d = _dev(torch, img.reshape(-1, 4))
    for m, om in ((kg.ReduceMode.Dither, oracle.MODE_DITHER), (kg.ReduceMode.Meld, oracle.MODE_MELD)):
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent, m, out.data_ptr(), _stream(torch))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(h, w, 4), oracle.find(img, pal, om)), m
    p.close()


@pytest.mark.parametrize("name", ["black_white", "crowded45", "crowded200", "crowded_second_half"])
def test_dither_lists_continued_overflowing_and_missing(torch_cuda, oracle, monkeypatch, name):
    """the list passes (dither and meld, kmg_lists.hip) where their lists are not one plain record: a two-colour palette whose threshold
    throws dark pixels off the grid over Lab (no list: every centroid is scanned), 45 near-identical colours (continuation
    records), 200 of them (more than 63 candidates: no list) -- bytes equal the oracle's"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    rng = np.random.default_rng(len(name))
    w, h = 640, 301
    img = np.concatenate([_blobs(rng, w * 150, 12, sigma=30.0), oracle.synth_uniform(77, w * 151)]).reshape(h, w, 4)
    if name == "black_white":
        pal = np.array([[0, 0, 0, 255], [255, 255, 255, 255]], np.uint8)
    elif name == "crowded_second_half":
        # k = 380: 280 colours with r < 120 sort in front of a crowd of 100 -- more than 63 candidates in the SECOND list of a cell
        block = np.array([(120 + i % 6, 130 + (i // 6) % 6, 90 + i // 36, 255) for i in range(100)], np.uint8)
        rest = oracle.synth_uniform(6, 900); rest = rest[rest[:, 0] < 120][:280]
        pal = np.array(sorted(set(map(tuple, np.concatenate([block, rest])))), np.uint8)
        assert pal.shape[0] > 300 and (pal[:256, 0] < 120).all()
        img[:100, :, :3] = np.clip(rng.normal((122, 132, 92), 6.0, (100, w, 3)), 0, 255).astype(np.uint8)
    else:
        m = 45 if name == "crowded45" else 200
        block = np.array([(120 + i % 6, 130 + (i // 6) % 6, 90 + i // 36, 255) for i in range(m)], np.uint8)
        rest = oracle.synth_uniform(5, 256 - m)
        pal = np.array(sorted(set(map(tuple, np.concatenate([block, rest])))), np.uint8)
        img[:100, :, :3] = np.clip(rng.normal((122, 132, 92), 6.0, (100, w, 3)), 0, 255).astype(np.uint8)   # pixels among the crowd
    cent = kg.palette_to_centroids(pal)
    _set_strategy("table")
    p = kg.ImageProcessor()
    d = _dev(torch, img.reshape(-1, 4))
    out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    for mode, omode in ((kg.ReduceMode.Dither, oracle.MODE_DITHER), (kg.ReduceMode.Meld, oracle.MODE_MELD)):
        p.apply(d.data_ptr(), w, h, 0, cent, mode, out.data_ptr(), _stream(torch))
        torch.cuda.synchronize()
        want = oracle.find(img, pal, omode)
        got = out.cpu().numpy().reshape(h, w, 4)
        assert np.array_equal(got, want), f"{mode}: {int((got != want).any(-1).sum())} pixels differ"
    p.close()


@pytest.mark.parametrize("kind,w,h,k", [("tokyo", 256, 171, 2), ("tokyo", 256, 171, 33), ("few", 67, 41, 6),
                                        ("noise", 300, 200, 17), ("noise", 1, 1, 3), ("few", 16, 1, 4),
                                        # k >= 32 over the colours: several picks per launch (k_init_cells_multi) -- more centroids
                                        # than colours (every distance 0: pixel 0 again and again), noise, a photograph, k = 256
                                        ("few", 67, 41, 40), ("noise", 300, 200, 64), ("tokyo", 256, 171, 150), ("noise", 96, 64, 256)])
def test_init_over_colours_equals_init_over_pixels(torch_cuda, oracle, tokyo, monkeypatch, kind, w, h, k):
    """farthest-point init walking the image's colours (forced) == walking its pixels == oracle,
    including the arg-max tie rule on images with few colours"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    rng = np.random.default_rng(w * 7 + k)
    if kind == "tokyo":
        img = oracle.resize(tokyo, w, h)
    elif kind == "few":
        pal = rng.integers(0, 256, (5, 4), dtype=np.uint8); pal[:, 3] = 255
        img = pal[rng.integers(0, 5, (h, w))]
    else:
        img = oracle.synth_uniform(w * h + k, w * h).reshape(h, w, 4)
    want = oracle.init_centroids(oracle.rgb_to_lab(img), w, h, k)
    d = _dev(torch, img.reshape(-1, 4))
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor(shrink_max_dim=0)
        s = kg.Lloyd(p, k)
        s.init_centroids(d.data_ptr(), w, h, st)
        got = s.get_centroids(st)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), strategy
        # and the Lloyd loop that follows works from the same bound image
        labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
        want_c, want_l, _ = oracle.lloyd(oracle.rgb_to_lab(img), want)
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32)), strategy
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l), strategy
        s.close()
        p.close()


def test_buffer_reused_for_a_second_image(torch_cuda, oracle, monkeypatch):
    """one Lloyd object, one device buffer, two different images in turn (colour table forced): the
    initialisation re-binds, so the second problem never sees the first image's histogram"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    w, h, k = 160, 90, 7
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    d = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    for seed in (1, 2):
        img = _blobs(np.random.default_rng(seed), w * h, 6, sigma=15.0).reshape(h, w, 4)
        d.copy_(torch.from_numpy(img.reshape(-1, 4)))
        s.init_centroids(d.data_ptr(), w, h, st)
        s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
        lab = oracle.rgb_to_lab(img)
        want_c, want_l, _ = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32)), seed
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l), seed
    s.close()
    p.close()


def test_run_twice_on_one_buffer_with_set_centroids(torch_cuda, oracle, monkeypatch):
    """kmg_lloyd_run binds the image on its own behalf and must not leave that binding behind: a second
    run on the same device buffer with new pixels and set_centroids (no initialisation in between, e.g. a
    frame loop or a buffer recycled by the caching allocator) works from the NEW image's histogram.  A
    binding the caller made explicitly (prepare) survives the run."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    w, h, k = 200, 120, 9
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    d = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    for seed in (3, 4, 5):
        img = _blobs(np.random.default_rng(seed), w * h, 7, sigma=18.0).reshape(h, w, 4)
        d.copy_(torch.from_numpy(img.reshape(-1, 4)))
        lab = oracle.rgb_to_lab(img)
        init = oracle.init_centroids(lab, w, h, k)
        s.set_centroids(init, st)
        s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
        want_c, want_l, _ = oracle.lloyd(lab, init)
        assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32)), seed
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l), seed
    # explicit binding: kept across run() (the caller vouches for the contents)
    assert s.prepare(d.data_ptr(), w * h, True, st) == "table"
    s.set_centroids(init, st)
    s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.assign_accumulate(d.data_ptr(), w * h, 0, acc.data_ptr(), st)       # table pass: works only while bound
    torch.cuda.synchronize()
    assert s.debug_table_stats(st)["distinct_colours"] > 0
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l)
    s.close()
    p.close()


@pytest.mark.parametrize("k,kind", [(24, "blobs"), (300, "blobs"), (64, "uniform"), (256, "uniform")])
def test_pipelined_iterate_equals_step_by_step(torch_cuda, oracle, monkeypatch, k, kind):
    """kmg_lloyd_iterate (label pass of iteration t on the side stream beside the scan of iteration t + 1, issued by the
    next call or by a flush; two alternating sets of label tables) gives the same centroids, sums and -- after every
    iteration -- the same label map as update + assign_accumulate step by step, and as the oracle.  Noise has no hot cells
    (the label pass shares its compute units); blobs have them, and k = 300 has no pair tables (serial order inside)."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    w, h = 700, 500
    n = w * h
    img = _blobs(np.random.default_rng(k), n, 40, sigma=25.0) if kind == "blobs" else oracle.synth_uniform(k, n)
    lab = oracle.rgb_to_lab(img)
    init = oracle.centroids4(lab[np.random.default_rng(1).choice(n, k, replace=False)])
    d = _dev(torch, img)
    p = kg.ImageProcessor(shrink_max_dim=0)
    a, b = kg.Lloyd(p, k), kg.Lloyd(p, k)
    for s in (a, b):
        s.set_centroids(init, st)
        assert s.prepare(d.data_ptr(), n, True, st) == "table"
    la, lb = (torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2))
    acc_a, acc_b = (torch.zeros((k, 4), dtype=torch.int64, device="cuda") for _ in range(2))
    cent = init
    for it in range(6):
        if it:
            a.update(acc_a.data_ptr(), st)
        a.assign_accumulate(d.data_ptr(), n, la.data_ptr(), acc_a.data_ptr(), st)
        b.iterate(d.data_ptr(), n, lb.data_ptr(), acc_b.data_ptr(), it > 0, st)
        if it in (0, 3, 5):                      # look at the label map of this very iteration
            b.flush(st)
            torch.cuda.synchronize()
            assert torch.equal(la, lb), it
            assert torch.equal(acc_a, acc_b), it
            wl, wa = oracle.assign_accumulate_rgba(img, cent)
            assert np.array_equal(lb.cpu().numpy().view(np.uint32), wl) and np.array_equal(acc_b.cpu().numpy(), wa), it
        cent, _ = oracle.finalize(oracle.assign_accumulate_rgba(img, cent)[1], cent)
    b.flush(st)
    torch.cuda.synchronize()
    assert torch.equal(la, lb) and torch.equal(acc_a, acc_b)
    assert np.array_equal(a.get_centroids(st).view(np.uint32), b.get_centroids(st).view(np.uint32))
    # mixing in a synchronous pass afterwards still sees consistent tables
    b.update(acc_b.data_ptr(), st); a.update(acc_a.data_ptr(), st)
    a.assign_accumulate(d.data_ptr(), n, la.data_ptr(), acc_a.data_ptr(), st)
    b.assign_accumulate(d.data_ptr(), n, lb.data_ptr(), acc_b.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(la, lb) and torch.equal(acc_a, acc_b)
    a.close(); b.close(); p.close()


def test_meld_masks_conservative_for_all_colours(torch_cuda, processor, oracle):
    """pruned meld pass: over all 2^24 colours the two closest centroids found among the cell's candidates
    are the two closest of the full ordered scan"""
    rng = np.random.default_rng(13)
    for name, cent in _centroid_sets(oracle, rng).items():
        if cent.shape[0] < 2:
            continue
        assert processor.debug_check_meld_masks(cent, _stream(torch_cuda)) == 0, name


@pytest.mark.parametrize("k", [2, 3, 46, 64, 65, 256, 300, 512, 600])
def test_meld_output_pass_pruned_equals_scan(torch_cuda, oracle, monkeypatch, k):
    """find / reduce in meld mode: candidate-pruned pass (forced) is byte-identical to the scan of all
    centroids (the oracle comparison of the scan itself is tests/test_gpu_parity.py::test_meld_matches_oracle)"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    w, h = 801, 600
    img = _blobs(np.random.default_rng(200 + k), w * h, 25, sigma=25.0).reshape(h, w, 4)
    pal = np.array(sorted(set(map(tuple, oracle.synth_uniform(k + 11, k)))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    d = _dev(torch, img.reshape(-1, 4))
    st = _stream(torch)
    outs = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor()
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent, kg.ReduceMode.Meld, out.data_ptr(), st)
        torch.cuda.synchronize()
        outs[strategy] = out.cpu().numpy()
        p.close()
    assert np.array_equal(outs["brute"], outs["table"])


def test_partitioned_histogram_with_crowded_partitions(torch_cuda, oracle, monkeypatch):
    """> 2^21 pixels in very few colours: colour partitions larger than one histogram chunk (the multi-chunk,
    atomic-merge path of the partitioned build), many exact ties in the colour-based init"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    w, h, k = 2048, 1537, 4
    rng = np.random.default_rng(77)
    pal = np.array([[10, 20, 30, 255], [10, 20, 31, 255], [200, 100, 50, 255], [0, 0, 0, 255], [255, 255, 255, 255]], np.uint8)
    img = pal[rng.choice(5, size=(h, w), p=[0.5, 0.3, 0.1, 0.05, 0.05])]
    lab = oracle.rgb_to_lab(img)
    want_init = oracle.init_centroids(lab, w, h, k)
    want_c, want_l, _ = oracle.lloyd(lab, want_init)
    d = _dev(torch, img.reshape(-1, 4))
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.init_centroids(d.data_ptr(), w, h, st)
    assert np.array_equal(s.get_centroids(st).view(np.uint32), want_init.view(np.uint32))
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
    assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l)
    s.close()
    p.close()


def test_cfg2_full_size_assign_update(torch_cuda, oracle, monkeypatch):
    """BASELINE config 2: synthetic 4096x4096, k=16, assign + update only -- four iterations with each
    strategy: identical labels / sums / centroids, the size-independent properties, a prefix vs the oracle"""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w = h = 4096
    n, k = w * h, 16
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG2, n, device="cuda")
    sel = synth.uniform_rgba_at(synth.SEED_CFG2, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
    cent0 = oracle.centroids4(oracle.rgb_to_lab(sel))
    res = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor(shrink_max_dim=0)
        s = kg.Lloyd(p, k)
        s.set_centroids(cent0, st)
        assert s.prepare(rgba.data_ptr(), n, True, st) == ("table" if strategy == "table" else "scan")
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        for _ in range(4):
            s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
            s.update(acc.data_ptr(), st)
        torch.cuda.synchronize()
        res[strategy] = (labels, acc.clone(), s.get_centroids(st))
        s.close()
        p.close()
    (l0, a0, c0), (l1, a1, c1) = res["brute"], res["table"]
    assert torch.equal(l0, l1) and torch.equal(a0, a1) and np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert int(a1[:, 3].sum()) == n and int(l1.max()) < k
    assert int(torch.bincount(l1.to(torch.int64), minlength=k).sum()) == n
    # the whole run on the oracle (16.7 M pixels x 16 centroids x 4 passes: seconds with OpenMP)
    host = rgba.cpu().numpy()
    cent = cent0
    for _ in range(4):
        want_l, want_a = oracle.assign_accumulate_rgba(host, cent)
        cent, _ = oracle.finalize(want_a, cent)
    assert np.array_equal(l1.cpu().numpy().view(np.uint32), want_l)
    assert np.array_equal(a1.cpu().numpy(), want_a)
    assert np.array_equal(c1.view(np.uint32), cent.view(np.uint32))


def test_cfg5_full_size_find_dither(torch_cuda, oracle, monkeypatch):
    """BASELINE config 5: find + ordered dither with the 64-entry resurrect_64 palette on synthetic
    8192x8192: pruned pass == scan of all centroids, every output pixel is a palette colour, the first
    rows equal the oracle"""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    from PIL import Image
    import os
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n = w * h
    px = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
    pal = np.array(sorted(set(map(tuple, px))), np.uint8)
    assert pal.shape[0] == 64
    cent = kg.palette_to_centroids(pal)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
    outs = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor()
        out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        p.apply(rgba.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
        torch.cuda.synchronize()
        outs[strategy] = out
        p.close()
    assert torch.equal(outs["brute"], outs["table"])
    used = torch.unique(outs["table"].view(torch.int32)).cpu().numpy().view(np.uint32)
    # dither emits lab_to_rgb(centroid): at most 64 distinct colours
    assert used.size <= 64
    rows = 4
    head = rgba[: rows * w].cpu().numpy().reshape(rows, w, 4)
    want = oracle.find(head, pal, oracle.MODE_DITHER)
    assert np.array_equal(outs["table"][: rows * w].cpu().numpy().reshape(rows, w, 4), want)


@pytest.mark.gpu
def test_concurrent_output_passes_share_one_processor(torch_cuda, oracle, monkeypatch):
    """core/examples/parallel.rs: calls on one ImageProcessor run concurrently.  The table / pruned output passes take
    their scratch from per-processor blocks (one per call in flight): six threads, three modes, two palettes on the same
    processor must return what the same calls return one after the other."""
    import threading
    import kmeans_gpu_amd as kg
    _set_strategy("table")
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (384, 512, 4), dtype=np.uint8)
    img[..., 3] = 255
    pals = [rng.integers(0, 256, (k, 4), dtype=np.uint8) for k in (24, 200)]
    for p in pals:
        p[:, 3] = 255
    proc = kg.ImageProcessor(shrink_max_dim=0)
    jobs = [(pi, mode) for pi in range(2) for mode in (kg.ReduceMode.Replace, kg.ReduceMode.Dither, kg.ReduceMode.Meld)]
    serial = {j: proc.find(img, pals[j[0]], j[1]) for j in jobs}
    results, errors = {}, []

    def work(j):
        try:
            for _ in range(3):
                results[j] = proc.find(img, pals[j[0]], j[1])
        except Exception as e:      # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for j in jobs:
        assert np.array_equal(results[j], serial[j]), j
    want = oracle.find(img, pals[0], int(kg.ReduceMode.Dither))
    assert np.array_equal(serial[(0, kg.ReduceMode.Dither)], want)


@pytest.mark.gpu
def test_label_pass_with_reserved_compute_units(torch_cuda, oracle, monkeypatch):
    """kmg_lloyd_reserve_cus: the label pass on fewer CUs (room for a collective beside it) writes the same label map."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy("table")
    rng = np.random.default_rng(5)
    n, k = 3_000_017, 200                                     # not a multiple of the tile size
    rgba = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    cent = oracle.centroids4(oracle.rgb_to_lab(rgba[:k]))
    d = _dev(torch, rgba)
    proc = kg.ImageProcessor(shrink_max_dim=0)
    maps = []
    for reserve in (0, 8, 128):
        s = kg.Lloyd(proc, k)
        s.set_centroids(cent, _stream(torch))
        assert s.prepare(d.data_ptr(), n, True, _stream(torch)) == "table"
        s.reserve_cus(reserve)
        labels = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), _stream(torch))
        torch.cuda.synchronize()
        maps.append(labels.cpu().numpy())
        s.close()
    assert np.array_equal(maps[0], maps[1]) and np.array_equal(maps[0], maps[2])
    want, _ = oracle.assign_accumulate_rgba(rgba[:200_000], cent)
    assert np.array_equal(maps[0][:200_000].view(np.uint32), want)
    with pytest.raises(kg.KmgError):
        kg.Lloyd(proc, k).reserve_cus(129)


@pytest.mark.parametrize("k", [1024, 3072])
def test_large_k_through_the_colour_table(torch_cuda, oracle, monkeypatch, k):
    """KMG_MAX_K = 3072 and a k between: the words > 4 candidate masks, u16 labels, cells beyond the listing limit and
    ~150 KiB of LDS per cube workgroup.  One assign + accumulate pass of a bound image equals the oracle, and the
    exhaustive check of the cube pass (bounds, masks, per-colour labels over all 2^24 colours) finds nothing."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    n = 300_000
    img = oracle.synth_uniform(5, n)
    cent = oracle.centroids4(oracle.rgb_to_lab(img[:k]))
    want_l, want_a = oracle.assign_accumulate_rgba(img, cent)
    d = _dev(torch, img)
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(cent, st)
    assert s.prepare(d.data_ptr(), n, True, st) == "table"
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l)
    assert np.array_equal(acc.cpu().numpy(), want_a)
    assert s.debug_check_table(st) == (0, 0, 0)
    s.close()
    p.close()


@pytest.mark.parametrize("k", [24, 300])
@pytest.mark.parametrize("strategy", ["table", "brute"])
def test_assign_update_equals_the_two_calls(torch_cuda, oracle, monkeypatch, k, strategy):
    """kmg_lloyd_assign_update (the update rides on the last launch of the colour-table assign pass; no memset, no
    k_update launch) == kmg_lloyd_assign_accumulate + kmg_lloyd_update == the oracle: labels, sums, centroids and the
    convergence count after every iteration; with do_update = 0 the centroids stay."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy(strategy)
    w, h = 640, 400
    n = w * h
    img = _blobs(np.random.default_rng(7 * k), n, 40, sigma=25.0)
    lab = oracle.rgb_to_lab(img)
    init = oracle.centroids4(lab[np.random.default_rng(2).choice(n, k, replace=False)])
    d = _dev(torch, img)
    p = kg.ImageProcessor(shrink_max_dim=0)
    a, b = kg.Lloyd(p, k), kg.Lloyd(p, k)
    for s in (a, b):
        s.set_centroids(init, st)
        assert s.prepare(d.data_ptr(), n, True, st) == ("table" if strategy == "table" else "scan")
    la, lb = (torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2))
    acc_a, acc_b = (torch.full((k, 4), 7, dtype=torch.int64, device="cuda") for _ in range(2))   # stale contents must not matter
    cent = init
    for it in range(5):
        a.assign_accumulate(d.data_ptr(), n, la.data_ptr(), acc_a.data_ptr(), st)
        a.update(acc_a.data_ptr(), st)
        b.assign_update(d.data_ptr(), n, lb.data_ptr(), acc_b.data_ptr(), True, st)
        torch.cuda.synchronize()
        wl, wa = oracle.assign_accumulate_rgba(img, cent)
        cent, conv = oracle.finalize(wa, cent)
        assert torch.equal(la, lb) and np.array_equal(lb.cpu().numpy().view(np.uint32), wl), it
        assert torch.equal(acc_a, acc_b) and np.array_equal(acc_b.cpu().numpy(), wa), it
        assert np.array_equal(a.get_centroids(st).view(np.uint32), b.get_centroids(st).view(np.uint32)), it
        assert np.array_equal(b.get_centroids(st).view(np.uint32), cent.view(np.uint32)), it
        assert a.converged_count(st) == b.converged_count(st) == conv, it
    before = b.get_centroids(st).copy()
    b.assign_update(d.data_ptr(), n, lb.data_ptr(), acc_b.data_ptr(), False, st)
    torch.cuda.synchronize()
    wl, wa = oracle.assign_accumulate_rgba(img, cent)
    assert np.array_equal(lb.cpu().numpy().view(np.uint32), wl) and np.array_equal(acc_b.cpu().numpy(), wa)
    assert np.array_equal(b.get_centroids(st).view(np.uint32), before.view(np.uint32))
    # the label tables still describe that assignment: a labels-only pass reproduces it
    lb.zero_()
    b.labels(d.data_ptr(), n, lb.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(lb.cpu().numpy().view(np.uint32), wl)
    a.close(); b.close(); p.close()


@pytest.mark.parametrize("kind,k", [("noise", 256), ("blobs", 64), ("noise", 40)])
def test_cube_pass_redeals_its_tasks_without_losing_a_cell(torch_cuda, oracle, monkeypatch, kind, k):
    """The one-launch cube pass (32 < k <= 256) deals its wave tasks out again from pass to pass -- every workgroup with one
    partner, by what the tasks' items cost in the previous pass (kmg_table.h CubeBalance).  Twenty passes walk every dimension
    of the exchange more than twice: sums, centroids and labels of every pass == the per-pixel scan's; then a cell share (another
    work list: the deal starts over) and a second image in the same buffer (a re-bind)."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    w, h = 1024, 1024
    n = w * h
    rng = np.random.default_rng(11 * k)
    img = oracle.synth_uniform(97 + k, n) if kind == "noise" else _blobs(rng, n, 60, sigma=30.0)
    lab = oracle.rgb_to_lab(img)
    init = oracle.centroids4(lab[rng.choice(n, k, replace=False)])
    d = _dev(torch, img)
    p = kg.ImageProcessor(shrink_max_dim=0)
    res = {}
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        s = kg.Lloyd(p, k)
        s.set_centroids(init, st)
        s.prepare(d.data_ptr(), n, True, st)
        lb = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        out = []
        for it in range(20):
            s.assign_update(d.data_ptr(), n, lb.data_ptr() if it % 5 == 4 else 0, acc.data_ptr(), True, st)
            torch.cuda.synchronize()
            out.append((acc.cpu().numpy().copy(), s.get_centroids(st).view(np.uint32).copy(), lb.cpu().numpy().copy() if it % 5 == 4 else None))
        if strategy == "table":
            # another work list: one share of the cube, sums only, four passes; then the whole cube again
            whole = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            s.assign_accumulate(d.data_ptr(), n, 0, whole.data_ptr(), st)
            parts = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            for part in range(3):
                s.set_cell_share(part, 3, st)
                for _ in range(4):
                    one = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
                    s.assign_accumulate(d.data_ptr(), n, 0, one.data_ptr(), st)
                parts += one
            s.set_cell_share(0, 1, st)
            torch.cuda.synchronize()
            assert torch.equal(parts, whole), "the shares' sums do not add up to the whole cube's"
            again = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            s.assign_accumulate(d.data_ptr(), n, 0, again.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(again, whole)
            # a second image in the same buffer: the binding, and with it the deal, starts over
            img2 = oracle.synth_uniform(5 + k, n)
            d.copy_(_dev(torch, img2))
            s.set_centroids(init, st)
            s.prepare(d.data_ptr(), n, True, st)
            for it in range(3):
                s.assign_update(d.data_ptr(), n, lb.data_ptr(), acc.data_ptr(), True, st)
            torch.cuda.synchronize()
            cent = init
            for it in range(3):
                wl, wa = oracle.assign_accumulate_rgba(img2, cent)
                cent, _ = oracle.finalize(wa, cent)
            assert np.array_equal(acc.cpu().numpy(), wa) and np.array_equal(lb.cpu().numpy().view(np.uint32), wl)
            d.copy_(_dev(torch, img))
        res[strategy] = out
        s.close()
    for it, (a, b) in enumerate(zip(res["brute"], res["table"])):
        assert np.array_equal(a[0], b[0]), f"sums differ in pass {it}"
        assert np.array_equal(a[1], b[1]), f"centroids differ after pass {it}"
        assert (a[2] is None) == (b[2] is None) and (a[2] is None or np.array_equal(a[2], b[2])), f"labels differ in pass {it}"
    p.close()


@pytest.mark.parametrize("world", [2, 4])
def test_cell_sharded_cube_pass_equals_unsharded(torch_cuda, oracle, tokyo, monkeypatch, world):
    """The cell-sharded loop (tests/sharded_harness.py ShardedLloyd(cells=True)) with the real kernels, `world` ranks emulated
    on one GPU in lockstep: band histograms summed into every rank's table (histogram_tensor / rebuild_from_histogram), each
    rank's cube pass over its share of the cube (set_cell_share), the k x 4 sums added up, the ranks' shares of the label
    tables copied to all (table_tensors), every band's label map from labels_from_tables -- labels, sums and centroids
    of five iterations equal the unsharded run and the oracle.  The photograph has hot cells (kmg_table.h)."""
    import kmeans_gpu_amd as kg
    from sharded_harness import band_rows, cell_range
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    k = 40
    img = tokyo[:500, :700].copy()
    h, w = img.shape[:2]
    n = w * h
    lab = oracle.rgb_to_lab(img)
    init = oracle.centroids4(lab.reshape(-1, 3)[np.random.default_rng(5).choice(n, k, replace=False)])
    d = _dev(torch, img).reshape(-1, 4)
    p = kg.ImageProcessor(shrink_max_dim=0)
    ranks = []
    for r in range(world):
        r0, r1 = band_rows(h, r, world)
        band = d[r0 * w:r1 * w]
        s = kg.Lloyd(p, k)
        s.set_centroids(init, st)
        s.bind_image(band.data_ptr(), (r1 - r0) * w, st)
        ranks.append((s, band, (r1 - r0) * w, torch.zeros((r1 - r0) * w, dtype=torch.int32, device="cuda"),
                      torch.zeros((k, 4), dtype=torch.int64, device="cuda")))
    hists = [s.histogram_tensor() for s, *_ in ranks]
    total_hist = torch.stack(hists).sum(0).to(torch.int32)
    assert int(total_hist.sum()) == n
    for r, (s, *_rest) in enumerate(ranks):
        hists[r].copy_(total_hist)                                  # the all-reduce
        s.rebuild_from_histogram(n, st)
        s.set_cell_share(r, world, st)
    tables = [s.table_tensors() for s, *_ in ranks]
    cent = init
    for it in range(5):
        total = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        for s, band, nb, labels, acc in ranks:
            s.assign_accumulate(band.data_ptr(), nb, 0, acc.data_ptr(), st)
            total += acc
        for r in range(world):                                      # the all-gather of the label tables
            c0, c1 = cell_range(r, world)
            for q in range(world):
                if q != r:
                    tables[q][0][c0 * 512:c1 * 512].copy_(tables[r][0][c0 * 512:c1 * 512])
                    tables[q][1][c0:c1].copy_(tables[r][1][c0:c1])
        want_l, want_a = oracle.assign_accumulate_rgba(img, cent)
        assert np.array_equal(total.cpu().numpy(), want_a), it
        for r, (s, band, nb, labels, acc) in enumerate(ranks):
            s.labels_from_tables(band.data_ptr(), nb, labels.data_ptr(), st)
            r0, r1 = band_rows(h, r, world)
            torch.cuda.synchronize()
            assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l[r0 * w:r1 * w]), (it, r)
            s.update(total.data_ptr(), st)
        cent, _ = oracle.finalize(want_a, cent)
        for s, *_rest in ranks:
            assert np.array_equal(s.get_centroids(st).view(np.uint32), cent.view(np.uint32)), it
    for s, *_rest in ranks:
        s.close()
    p.close()


def test_second_image_binds_without_a_hipmalloc(torch_cuda, oracle, monkeypatch):
    """The processor keeps the device blocks of finished kmg_lloyd objects (colour table, workspace, init tables): a second
    image on a warm processor -- a frame loop, the second image of a rank's share of a batch -- initialises, binds and runs
    without a single fresh block, and gives the same result as on a cold processor."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    w, h, k = 640, 480, 32
    imgs = [_blobs(np.random.default_rng(s), w * h, 30, sigma=20.0) for s in (1, 2)]
    p = kg.ImageProcessor(shrink_max_dim=0)
    results = []
    counts = []
    for img in imgs + imgs[:1]:
        d = _dev(torch, img)
        labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        s = kg.Lloyd(p, k)
        s.init_centroids(d.data_ptr(), w, h, st)
        it = s.run(d.data_ptr(), w * h, labels.data_ptr(), st)
        results.append((it, s.get_centroids(st).copy(), labels.cpu().numpy().copy()))
        s.close()
        counts.append(p.debug_block_counts())
    assert counts[0][0] > 0 and counts[0][1] == 0                   # cold: fresh blocks only
    assert counts[1][0] == counts[0][0] and counts[2][0] == counts[0][0], counts     # warm: not one more hipMalloc
    assert counts[2][1] > counts[1][1] > 0
    assert results[0][0] == results[2][0] and np.array_equal(results[0][1].view(np.uint32), results[2][1].view(np.uint32))
    assert np.array_equal(results[0][2], results[2][2])
    lab = oracle.rgb_to_lab(imgs[1].reshape(h, w, 4))
    want_c, want_l, want_it = oracle.lloyd(lab, oracle.init_centroids(lab, w, h, k))
    assert results[1][0] == want_it and np.array_equal(results[1][1].view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(results[1][2].view(np.uint32), want_l)
    p.close()


def test_crowded_centroids_take_the_long_candidate_lists(torch_cuda, oracle, monkeypatch):
    """k = 256 centroids crowded into the dark corner of the cube (what a photograph does to k-means): the cells there have
    far more than 32 candidates, so the stage kernel bounds them per sub-cell from its long list (long_list_stage) and the
    scan kernel visits the sub-cells' own sets.  Exhaustive check of the cube pass over all 2^24 colours, and labels / sums /
    two iterations of an image drawn from that corner against the oracle."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    rng = np.random.default_rng(77)
    k, n = 256, 400_000
    pal = np.zeros((k, 4), np.uint8); pal[:, :3] = rng.integers(0, 48, (k, 3)); pal[:, 3] = 255
    cent = oracle.centroids4(oracle.rgb_to_lab(pal))
    img = np.zeros((n, 4), np.uint8); img[:, :3] = rng.integers(0, 64, (n, 3)); img[:, 3] = 255
    d = _dev(torch, img)
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(cent, st)
    assert s.prepare(d.data_ptr(), n, True, st) == "table"
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for it in range(2):
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        want_l, want_a = oracle.assign_accumulate_rgba(img, cent)
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l), it
        assert np.array_equal(acc.cpu().numpy(), want_a), it
        stats = s.debug_table_stats(st)
        assert stats["max_candidates"] > 32 and stats["cells_unlisted"] > 0, stats     # the long path really ran
        assert s.debug_check_table(st) == (0, 0, 0)
        assert s.debug_check_pairs(st)[0] == 0
        s.update(acc.data_ptr(), st)
        cent, _ = oracle.finalize(want_a, cent)
    s.close()
    p.close()


def test_idle_blocks_stay_bounded_when_images_grow(torch_cuda, oracle, monkeypatch):
    """The processor keeps the blocks of finished objects for the next image, but not without bound: binding images of
    growing size (each needs blocks the earlier ones cannot serve) leaves at most 24 idle blocks / 3 GiB behind, and the
    results stay those of the oracle."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    p = kg.ImageProcessor(shrink_max_dim=0)
    k = 8
    for i, n in enumerate([40_000 + 25_000 * j for j in range(40)]):
        _set_strategy("table" if i % 2 else "brute")
        img = synth.uniform_rgba_torch(900 + i, n, device="cuda")
        s = kg.Lloyd(p, k + i)                                   # growing workspaces too
        s.init_centroids(img.data_ptr(), n // 100, 100, st)      # growing distance maps (per-pixel init on the brute rounds)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        s.run(img.data_ptr(), n, labels.data_ptr(), st)
        if i in (0, 17, 39):
            host = img.cpu().numpy()
            lab = oracle.rgb_to_lab(host)
            want_c, want_l, _ = oracle.lloyd(lab, oracle.init_centroids(lab, n // 100, 100, k + i))
            assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l)
            assert np.array_equal(s.get_centroids(st).view(np.uint32), want_c.view(np.uint32))
        s.close()
        blocks, nbytes = p.debug_idle_blocks()
        assert blocks <= 24 and nbytes <= 3 << 30, (i, blocks, nbytes)
    mallocs, reuses = p.debug_block_counts()
    assert reuses > 0
    p.close()


def test_cell_share_refuses_what_it_cannot_answer(torch_cuda, oracle, monkeypatch):
    """With a cell share set a pass returns one share's sums: a label map, a fused update, the loop, the two-step partial
    sums and the table statistics are refused instead of returning a fraction of the image's result."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    _set_strategy("table")
    n, k = 300_000, 12
    img = synth.uniform_rgba_torch(4711, n, device="cuda")
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(oracle.centroids4(oracle.rgb_to_lab(img[:k].cpu().numpy())), st)
    s.bind_image(img.data_ptr(), n, st)
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.set_cell_share(0, 2, st)
    s.assign_accumulate(img.data_ptr(), n, 0, acc.data_ptr(), st)                 # the share's sums: fine
    torch.cuda.synchronize()
    assert 0 < int(acc[:, 3].sum()) < n
    for call in (lambda: s.assign_accumulate(img.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st),
                 lambda: s.assign_update(img.data_ptr(), n, 0, acc.data_ptr(), True, st),
                 lambda: s.assign_partials(img.data_ptr(), n, 0, st),
                 lambda: s.iterate(img.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st),
                 lambda: s.run(img.data_ptr(), n, labels.data_ptr(), st),
                 lambda: s.debug_table_stats(st)):
        with pytest.raises(kg.KmgError):
            call()
    s.set_cell_share(0, 1, st)                                                      # the whole cube again
    s.assign_accumulate(img.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    want_l, want_a = oracle.assign_accumulate_rgba(img.cpu().numpy(), s.get_centroids(st))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_l) and np.array_equal(acc.cpu().numpy(), want_a)
    s.close()
    p.close()


def test_dominance_phase_runs_on_spread_images_only(torch_cuda, processor, oracle):
    """32 < k <= 256: the cube pass of an image WITHOUT hot cells (noise) is the one-launch k_cube_one with its dominance tests --
    they remove candidates, decide sub-cells, and the tables stay exactly right (exhaustive check, pair entries, labels and sums
    against the oracle); an image with hot cells (flat areas: the photograph case) takes the three launches, which make none"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    k, n = 200, 1_500_000
    for kind in ("uniform", "hot"):
        rgba = oracle.synth_uniform(777, n)
        if kind == "hot":
            rgba[: n // 2, :3] = (rgba[: n // 2, :3] & 3) + 40       # half of the pixels in one cell
        cent = oracle.centroids4(oracle.rgb_to_lab(rgba[n // 2::(n // 2) // k][:k]))
        wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
        d = _dev(torch, rgba)
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent)
        s.bind_image(d.data_ptr(), n, st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl), kind
        assert np.array_equal(acc.cpu().numpy(), wa), kind
        stats = s.debug_table_stats(st)
        assert s.debug_check_pairs(st)[0] == 0
        assert s.debug_check_table(st) == (0, 0, 0)
        if kind == "uniform":
            assert stats["candidates_pruned"] > 0 and stats["sub_cells_pruned_to_one"] > 0, stats
        else:
            assert stats["candidates_pruned"] == 0 and stats["sub_cells_pruned_to_one"] == 0, stats
        s.close()


def test_dominance_phase_with_crowded_centroids(torch_cuda, processor, oracle):
    """k_cube_one where its packed items do not fit: 100 of 200 centroids crowded around one colour give the cells near it more
    than 12 candidates per pair of sub-cells (items by list position) and some more than 32 (no list: every colour against the
    cell's mask) -- labels and sums of a noise image == the oracle, the tables exhaustively right"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    rng = np.random.default_rng(23)
    n = 1_200_000
    rgba = oracle.synth_uniform(4711, n)
    lab = oracle.rgb_to_lab(rgba[:100])
    crowd = (oracle.rgb_to_lab(np.array([[120, 130, 90, 255]], np.uint8))[0] + rng.normal(0, 0.6, (100, 3))).astype(np.float32)
    cent = oracle.centroids4(np.concatenate([crowd, lab]))
    k = cent.shape[0]
    wl, wa = oracle.assign_accumulate_rgba(rgba, cent)
    d = _dev(torch, rgba)
    s = kg.Lloyd(processor, k)
    s.set_centroids(cent)
    s.bind_image(d.data_ptr(), n, st)
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), wl)
    assert np.array_equal(acc.cpu().numpy(), wa)
    stats = s.debug_table_stats(st)
    assert stats["candidates_pruned"] > 0 and stats["max_candidates"] > 32 and stats["cells_unlisted"] > 0, stats
    assert s.debug_check_pairs(st)[0] == 0
    assert s.debug_check_table(st) == (0, 0, 0)
    s.close()
