"""Oracle parity at BASELINE.json's full sizes (-m gpu): every label and every accumulator of one cfg3
pass, the literal CIE94 arg-min of the reference (find_centroid.wgsl:32-41) on every pixel, >= 1024 rows
of the cfg5 dither pass, and one rank's share of cfg4 (16 x 8192x8192 tiled over 8 GPUs)."""
import os

import numpy as np
import pytest

from conftest import set_strategy as _set_strategy

pytestmark = pytest.mark.gpu


def _stream(torch):
    return torch.cuda.current_stream().cuda_stream


def _cfg3_centroids(oracle, synth, n, k):
    """bench.py's initial centroids: shader Lab of the pixels at linear index j * floor(N / k)"""
    sel = synth.uniform_rgba_at(synth.SEED_CFG3, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
    return oracle.centroids4(oracle.rgb_to_lab(sel))


def test_cfg3_full_pass_every_label_and_sum_vs_oracle(torch_cuda, oracle, monkeypatch):
    """BASELINE config 3 (8192x8192, k=256), colour-table strategy: ALL 67 M labels and all k x 4 int64
    accumulators of an assign+accumulate pass equal the oracle's, for the initial centroids and again
    after one centroid update (oracle update from the oracle's own sums)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    _set_strategy("table")
    n, k = 8192 * 8192, 256
    st = _stream(torch)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    host = rgba.cpu().numpy()
    cent = _cfg3_centroids(oracle, synth, n, k)
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(cent, st)
    assert s.prepare(rgba.data_ptr(), n, True, st) == "table"
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for step in range(2):
        s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        want_l, want_a = oracle.assign_accumulate_rgba(host, cent)
        got_l = labels.cpu().numpy().view(np.uint32)
        assert np.array_equal(got_l, want_l), f"pass {step}: {int((got_l != want_l).sum())} labels differ"
        assert np.array_equal(acc.cpu().numpy(), want_a), f"pass {step}: accumulators differ"
        s.update(acc.data_ptr(), st)
        cent, _ = oracle.finalize(want_a, cent)
        assert np.array_equal(s.get_centroids(st).view(np.uint32), cent.view(np.uint32))
    # a third pass through the call bench.py times: kmg_lloyd_assign_update(..., labels, acc, do_update = 1) -- the centroid
    # update on the last launch of the assign pass (CubeTail): labels and sums of THIS assignment, centroids AFTER the update
    labels.zero_(); acc.zero_()
    s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
    torch.cuda.synchronize()
    want_l, want_a = oracle.assign_accumulate_rgba(host, cent)
    got_l = labels.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_l, want_l), f"fused pass: {int((got_l != want_l).sum())} labels differ"
    assert np.array_equal(acc.cpu().numpy(), want_a), "fused pass: accumulators differ"
    cent, _ = oracle.finalize(want_a, cent)
    assert np.array_equal(s.get_centroids(st).view(np.uint32), cent.view(np.uint32)), "fused pass: updated centroids differ"
    s.close()
    p.close()


def test_photograph_at_full_size_hot_cells_and_long_lists_vs_oracle(torch_cuda, oracle, monkeypatch):
    """The tiled 8192 x 8192 photograph of bench.py (synthetic_image("photo")), k = 256: its pixels crowd into a few dark
    cells, so the label pass keeps hot cells in LDS (k_labels_pairs<true>) and the cube pass bounds those cells' candidates
    from long lists (long_list_stage) -- both asserted to have run -- and every label and every sum of one assign pass after
    three Lloyd iterations equals the oracle's (find_centroid.wgsl:15-44, choose_centroid.wgsl:97-104)."""
    import bench
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy("table")
    st = _stream(torch)
    n, k = 8192 * 8192, 256
    rgba = bench.synthetic_image("photo", n, 0, k, 0x5EED0B10)
    host = rgba.cpu().numpy()
    sel = host[np.arange(k, dtype=np.int64) * (n // k)]
    cent = oracle.centroids4(oracle.rgb_to_lab(sel))
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(cent, st)
    assert s.prepare(rgba.data_ptr(), n, True, st) == "table"
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for _ in range(3):                                             # the centroids move into the crowded cells
        s.assign_update(rgba.data_ptr(), n, 0, acc.data_ptr(), True, st)
    cent = s.get_centroids(st)
    s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
    torch.cuda.synchronize()
    stats = s.debug_table_stats(st)
    assert stats["max_candidates"] > 32 and stats["cells_unlisted"] > 0, stats       # long candidate lists were taken
    mismatching, resolved, total = s.debug_check_pairs(st)
    assert mismatching == 0 and total == n
    want_l, want_a = oracle.assign_accumulate_rgba(host, cent)
    got_l = labels.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_l, want_l), f"{int((got_l != want_l).sum())} labels differ"
    assert np.array_equal(acc.cpu().numpy(), want_a), "accumulators differ"
    assert resolved / total > 0.6, "the hot cells did not resolve the crowded pixels in LDS"
    s.close()
    p.close()


@pytest.mark.parametrize("k", [300, 512])
def test_two_list_dither_and_meld_rows_at_full_size_vs_oracle(torch_cuda, oracle, k):
    """256 < k <= 512 on the 8192 x 8192 image of config 5: the dither pass over two byte lists per Lab cell
    (k_dither_lists<2>) and the meld pass that merges the two halves' sorted triples (k_meld_lists<2>) -- the library's own
    choice at this size -- against the oracle on three bands of 512 rows (mix_colors.wgsl:29-90)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n = w * h
    rng = np.random.default_rng(1000 + k)
    pal = np.full((k, 4), 255, np.uint8)
    pal[:, :3] = rng.integers(0, 256, (k, 3))
    pal = np.array(sorted(set(map(tuple, pal))), np.uint8)
    assert pal.shape[0] > 256
    cent = kg.palette_to_centroids(pal)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
    p = kg.ImageProcessor()
    out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    for mode, omode in ((kg.ReduceMode.Dither, oracle.MODE_DITHER), (kg.ReduceMode.Meld, oracle.MODE_MELD)):
        p.apply(rgba.data_ptr(), w, h, 0, cent, mode, out.data_ptr(), st)
        torch.cuda.synchronize()
        for r0, rows in ((0, 512), (4000, 512), (h - 512, 512)):     # r0 % 4 == 0: the oracle's Bayer rows line up
            src = rgba[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
            want = oracle.find(src, pal, omode)
            got = out[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
            assert np.array_equal(got, want), f"{mode.name} rows {r0}..{r0 + rows}: {int((got != want).any(-1).sum())} pixels differ"
    p.close()


def test_cfg3_labels_equal_the_literal_cie94_argmin_on_every_pixel(torch_cuda, oracle, monkeypatch):
    """The reference's arg-min is over the literal distance_cie94 with strict '<' (find_centroid.wgsl:32-41,
    delta_e.wgsl:1-22).  A label is a function of the 24-bit colour, so the literal arg-min of all 2^24
    colours (oracle, literal=1) is a complete table of the reference labels: every pixel of the cfg3 image
    must carry it -- for the initial centroids and for the centroids after four Lloyd iterations."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    _set_strategy("table")
    n, k = 8192 * 8192, 256
    st = _stream(torch)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    idx = np.arange(1 << 24, dtype=np.uint32)
    cube = np.empty((1 << 24, 4), np.uint8)
    cube[:, 0] = idx & 255; cube[:, 1] = (idx >> 8) & 255; cube[:, 2] = (idx >> 16) & 255; cube[:, 3] = 255
    cube_lab = oracle.rgb_to_lab(cube)
    v = rgba.view(torch.int32).reshape(-1)
    colour = (v & 0xFFFFFF).to(torch.int64)                        # r | g << 8 | b << 16
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.set_centroids(_cfg3_centroids(oracle, synth, n, k), st)
    s.prepare(rgba.data_ptr(), n, True, st)
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    for round_ in range(2):
        cent = s.get_centroids(st)
        s.assign_accumulate(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        lut = torch.from_numpy(oracle.assign(cube_lab, cent, literal=True).astype(np.int32)).cuda()
        want = lut[colour]
        bad = int((want != labels).sum())
        assert bad == 0, f"round {round_}: {bad} of {n} pixels differ from the literal CIE94 arg-min"
        for _ in range(4):
            s.update(acc.data_ptr(), st)
            s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
    s.close()
    p.close()


def _colour_cube():
    idx = np.arange(1 << 24, dtype=np.uint32)
    cube = np.empty((1 << 24, 4), np.uint8)
    cube[:, 0] = idx & 255; cube[:, 1] = (idx >> 8) & 255; cube[:, 2] = (idx >> 16) & 255; cube[:, 3] = 255
    return cube


def _tie_prone_centroids(oracle):
    """centroid tables on which ordering by the squared key alone is KNOWN to differ from the literal arg-min
    for some colours (tests/test_oracle_golden.py): the near-tie repair branch of every kernel is taken"""
    rng = np.random.default_rng(5)
    grey = np.repeat(np.arange(0, 256, 4, dtype=np.uint8)[:, None], 4, 1)
    grey16 = np.repeat(np.arange(0, 256, 16, dtype=np.uint8)[:, None], 4, 1)
    rnd = rng.integers(0, 256, (300, 4), dtype=np.uint8)
    lab = oracle.rgb_to_lab(rnd)
    return {"grey64": oracle.centroids4(oracle.rgb_to_lab(grey)),           # chunked scan, k <= 256 tables
            "grey16": oracle.centroids4(oracle.rgb_to_lab(grey16)),         # plain scan
            "random300": oracle.centroids4(lab),                            # u16 labels
            "duplicates": oracle.centroids4(np.concatenate([lab[:20], lab[:20], lab[7:8]]))}   # exact ties: lowest index


@pytest.mark.parametrize("strategy", ["brute", "table"])
def test_every_colour_gets_the_literal_argmin(torch_cuda, oracle, monkeypatch, strategy):
    """All 2^24 colours as one image, centroid tables with literal-distance ties and key near-ties: labels and
    sums of both strategies equal the oracle's literal arg-min (find_centroid.wgsl:32-41) -- including the
    colours where the key alone would decide differently."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy(strategy)
    st = _stream(torch)
    cube = _colour_cube()
    lab = oracle.rgb_to_lab(cube)
    d = torch.from_numpy(cube).cuda()
    n = cube.shape[0]
    p = kg.ImageProcessor(shrink_max_dim=0)
    flipped = 0
    for name, cent in _tie_prone_centroids(oracle).items():
        k = cent.shape[0]
        want = oracle.assign(lab, cent, literal=1)
        flipped += int((want != oracle.assign(lab, cent, literal=2)).sum())
        s = kg.Lloyd(p, k)
        s.set_centroids(cent, st)
        assert s.prepare(d.data_ptr(), n, True, st) == ("table" if strategy == "table" else "scan")
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        got = labels.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} colours differ from the literal arg-min"
        assert np.array_equal(acc.cpu().numpy(), oracle.accumulate(lab, want, k)), name
        if strategy == "table":
            assert s.debug_check_table(st) == (0, 0, 0), name
        s.close()
    assert flipped > 0            # the inputs really contain colours the key alone gets wrong
    p.close()


def test_dither_of_every_colour_takes_the_literal_argmin(torch_cuda, oracle, monkeypatch):
    """ordered dither (mix_colors.wgsl:50-83) of a 4096x1024 image holding 2^22 colours of the cube, grey palette:
    the per-pixel scan and the pruned pass both equal the oracle's literal arg-min"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    st = _stream(torch)
    w, h = 4096, 1024
    img = _colour_cube()[::4].reshape(h, w, 4).copy()
    pal = np.repeat(np.arange(0, 256, 4, dtype=np.uint8)[:, None], 4, 1); pal[:, 3] = 255
    cent = kg.palette_to_centroids(pal)
    want = oracle.find(img, pal, oracle.MODE_DITHER)
    d = torch.from_numpy(img.reshape(-1, 4)).cuda()
    for strategy in ("brute", "table"):
        _set_strategy(strategy)
        p = kg.ImageProcessor()
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(h, w, 4)
        assert np.array_equal(got, want), f"{strategy}: {int((got != want).any(-1).sum())} pixels differ"
        if strategy == "table":
            assert p.debug_check_dither_masks(cent, st) == 0
        p.close()


def test_cfg3_init_at_full_resolution_vs_oracle(torch_cuda, oracle, monkeypatch):
    """BASELINE config 3's own initialisation: the farthest-point init (plus_plus_init.wgsl:62-68,84-143,161-181,
    kmeans++_calc_diff.wgsl) of the WHOLE 8192x8192 image, k=256 -- 255 passes over 67 M pixels in the oracle --
    against the device init walking the image's colours (table strategy: k_init_fused, cell skipping, slots) and
    walking its pixels (k_init_pass): all 256 centroids bit-equal, in order."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n, k = w * h, 256
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    lab = oracle.rgb_to_lab(rgba.cpu().numpy())
    want = oracle.init_centroids(lab, w, h, k)
    del lab
    for strategy in ("table", "brute"):
        _set_strategy(strategy)
        p = kg.ImageProcessor(shrink_max_dim=0)
        s = kg.Lloyd(p, k)
        s.init_centroids(rgba.data_ptr(), w, h, st)
        got = s.get_centroids(st)
        diff = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))
        assert diff.size == 0, f"{strategy}: first differing centroid {int(diff[0])} of {diff.size}: {got[diff[0]]} != {want[diff[0]]}"
        s.close()
        p.close()


def test_cfg3_dither_k256_rows_vs_oracle(torch_cuda, oracle, monkeypatch):
    """BASELINE config 3's output pass: ordered dither (mix_colors.wgsl:50-83) with the k=256 centroids the cfg3 loop
    ends with (init at full resolution + Lloyd to convergence on the device), 8192x8192: the first 1024 rows, 1028 rows
    from row 4000 and the last 1028 rows equal the oracle's, for the pruned pass (default at this size) and the scan of
    all 256 centroids on the first band."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n, k = w * h, 256
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    p = kg.ImageProcessor(shrink_max_dim=0)
    s = kg.Lloyd(p, k)
    s.init_centroids(rgba.data_ptr(), w, h, st)
    s.run(rgba.data_ptr(), n, 0, st)
    cent = s.get_centroids(st)
    s.close()
    out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    p.apply(rgba.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
    torch.cuda.synchronize()
    first_want = None
    for r0, rows in ((0, 1024), (4000, 1028), (h - 1028, 1028)):     # r0 % 4 == 0: the oracle's Bayer rows line up
        src = rgba[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        want = oracle.apply(src, cent, oracle.MODE_DITHER)
        got = out[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        assert np.array_equal(got, want), f"rows {r0}..{r0 + rows}: {int((got != want).any(-1).sum())} pixels differ"
        if first_want is None:
            first_want = want
    # the same band through the per-pixel scan of all centroids (k_apply<DITHER>)
    _set_strategy("brute")
    p.apply(rgba.data_ptr(), w, 1024, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
    torch.cuda.synchronize()
    got = out[:1024 * w].cpu().numpy().reshape(1024, w, 4)
    assert np.array_equal(got, first_want), f"scan: {int((got != first_want).any(-1).sum())} pixels differ"
    p.close()


@pytest.mark.parametrize("palette", ["resurrect_64.png", "apollo-1x.png"])
def test_cfg5_find_dither_1024_rows_and_tail_vs_oracle(torch_cuda, oracle, palette):
    """BASELINE config 5 (find -m dither, fixed palette, 8192x8192): the first 1024 rows, 1028 rows that
    start inside the image (Bayer rows 3..2 across many periods) and the last 1028 rows equal the oracle."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    from PIL import Image
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n = w * h
    px = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", palette)).convert("RGBA")).reshape(-1, 4)
    pal = np.array(sorted(set(map(tuple, px))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
    p = kg.ImageProcessor()
    out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    p.apply(rgba.data_ptr(), w, h, 0, cent, kg.ReduceMode.Dither, out.data_ptr(), st)
    torch.cuda.synchronize()
    for r0, rows in ((0, 1024), (4000, 1028), (h - 1028, 1028)):     # r0 % 4 == 0: the oracle's Bayer rows line up
        src = rgba[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        want = oracle.find(src, pal, oracle.MODE_DITHER)
        got = out[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        assert np.array_equal(got, want), f"rows {r0}..{r0 + rows}: {int((got != want).any(-1).sum())} pixels differ"
    p.close()


def test_full_size_find_meld_rows_vs_oracle(torch_cuda, oracle):
    """find -m meld with the 64-entry palette on the 8192x8192 image of config 5 (the list pass over Lab cells,
    kmg_lists.hip): the first 512 rows, 516 rows from inside the image and the last 512 rows equal the oracle's bytes
    (mix_colors.wgsl:29-48, :85-90, lab_to_rgb.wgsl)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    from PIL import Image
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    n = w * h
    px = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
    pal = np.array(sorted(set(map(tuple, px))), np.uint8)
    cent = kg.palette_to_centroids(pal)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG5, n, device="cuda")
    p = kg.ImageProcessor()
    out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    p.apply(rgba.data_ptr(), w, h, 0, cent, kg.ReduceMode.Meld, out.data_ptr(), st)
    torch.cuda.synchronize()
    for r0, rows in ((0, 512), (4001, 516), (h - 512, 512)):
        src = rgba[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        want = oracle.find(src, pal, oracle.MODE_MELD)
        got = out[r0 * w:(r0 + rows) * w].cpu().numpy().reshape(rows, w, 4)
        assert np.array_equal(got, want), f"rows {r0}..{r0 + rows}: {int((got != want).any(-1).sum())} pixels differ"
    p.close()


def test_cfg4_batch_tiled_over_the_ranks_through_the_c_abi(torch_cuda, oracle):
    """BASELINE config 4 as north_star words it, behind the C ABI (kmg_group_lloyd_create_batch / _bind_batch / _run_batch): 16 images
    of 8192x8192 (seeds 0x5EED0400 + i), k=256, each TILED over the ranks in row bands, ONE all-reduce of the batch's 16 x k x 4
    sums per iteration.  The rank under test holds rows [3072, 4096) of every image -- rank 3 of 8 -- and two more ranks of the
    same loopback group on this GPU hold the rows above and below (the other seven ranks' share), so the exchange is the library's
    own.  After three iterations every image's centroids and labels equal the unsharded run on the whole image, and image 0's
    band equals the oracle."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w = h = 8192
    k, images, iters = 256, 16, 3
    bands_rows = [(0, 3072), (3072, 4096), (4096, 8192)]           # rank 3 of 8's band in the middle
    r0, r1 = bands_rows[1]
    n = w * h
    p = kg.ImageProcessor(shrink_max_dim=0, max_iterations=iters, check_period=8)
    cent0 = []
    for i in range(images):
        sel = synth.uniform_rgba_at(synth.SEED_CFG4 + i, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
        cent0.append(oracle.centroids4(oracle.rgb_to_lab(sel)))

    # unsharded runs, one image at a time (centroids + the band's labels kept); the images stay resident for the batch
    want, imgs = [], []
    for i in range(images):
        img = synth.uniform_rgba_torch(synth.SEED_CFG4 + i, n, device="cuda")
        s = kg.Lloyd(p, k)
        s.set_centroids(cent0[i], st)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        s.run(img.data_ptr(), n, labels.data_ptr(), st)
        want.append((s.get_centroids(st), labels[r0 * w:r1 * w].clone()))
        if i == 0:
            host_band = img[r0 * w:r1 * w].cpu().numpy()
        s.close()
        imgs.append(img)
        del labels
    torch.cuda.synchronize()
    p.close()

    with kg.Group(devices=[0, 0, 0], flags=kg.GROUP_LOOPBACK, shrink_max_dim=0, max_iterations=iters, check_period=8) as group:
        gl = kg.GroupLloyd(group, k, n_images=images)
        labels = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(images)]
        gl.bind_batch([[img.data_ptr() + a * w * 4 for a, _ in bands_rows] for img in imgs],
                      [[a for a, _ in bands_rows]] * images, [[b - a for a, b in bands_rows]] * images, w, h,
                      [[lab.data_ptr() + a * w * 4 for a, _ in bands_rows] for lab in labels])
        for i in range(images):
            gl.set_centroids(cent0[i], image=i)
        its = gl.run_batch()
        assert its == [iters - 1] * images, its
        for i in range(images):
            got = gl.get_centroids(image=i)
            assert np.array_equal(got.view(np.uint32), want[i][0].view(np.uint32)), f"image {i}"
            assert torch.equal(labels[i][r0 * w:r1 * w], want[i][1]), f"image {i}"
        wl, _ = oracle.assign_accumulate_rgba(host_band, gl.get_centroids(image=0))
        assert np.array_equal(labels[0][r0 * w:r1 * w].cpu().numpy().view(np.uint32), wl)
        gl.close()


def test_band_with_random_alpha_equals_the_oracle_and_the_opaque_band(torch_cuda, oracle, monkeypatch):
    """Alpha is ignored on input (rgb_to_lab.wgsl:78) and forced to 255 on output (lab_to_rgb.wgsl:37): a 1024-row band of the
    8192-wide benchmark image with RANDOM alpha bytes gives the labels, sums and dither / meld / replace bytes of the same band
    with alpha 255 -- and those of the oracle -- through both strategies (the colour table masks alpha when it builds colour
    indices, kmg_table.h colour_index; the per-pixel kernels never read it)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = _stream(torch)
    w, rows, k = 8192, 1024, 64
    n = w * rows
    opaque = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    rgba = opaque.clone()
    rgba[:, 3] = torch.randint(0, 256, (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    assert int((rgba[:, 3] != 255).sum()) > n // 2
    cent = _cfg3_centroids(oracle, synth, n, k)
    host = rgba.cpu().numpy()
    want_l, want_a = oracle.assign_accumulate_rgba(host[:w * 64], cent)            # (the oracle on the first 64 rows)
    p = kg.ImageProcessor(shrink_max_dim=0)
    results = {}
    for strategy in ("table", "brute"):
        _set_strategy(strategy)
        for name, src in (("alpha", rgba), ("opaque", opaque)):
            s = kg.Lloyd(p, k)
            s.set_centroids(cent, st)
            s.prepare(src.data_ptr(), n, True, st)
            labels = torch.zeros(n, dtype=torch.int32, device="cuda")
            acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            s.assign_accumulate(src.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st)
            outs = []
            for mode in (kg.ReduceMode.Replace, kg.ReduceMode.Dither, kg.ReduceMode.Meld):
                out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
                p.apply(src.data_ptr(), w, rows, 0, cent, mode, out.data_ptr(), st)
                outs.append(out)
            torch.cuda.synchronize()
            results[(strategy, name)] = (labels, acc, *outs)
            s.close()
    ref = results[("table", "opaque")]
    for key, got in results.items():
        for a, b in zip(got, ref):
            assert torch.equal(a, b), key
    assert np.array_equal(ref[0][:w * 64].cpu().numpy().view(np.uint32), want_l)
    assert int((ref[3][:, 3] != 255).sum()) == 0
    band = host[:w * 64].reshape(64, w, 4)
    assert np.array_equal(results[("table", "alpha")][3][:w * 64].cpu().numpy().reshape(64, w, 4), oracle.apply(band, cent, oracle.MODE_DITHER))
    p.close()
