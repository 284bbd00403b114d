"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.
Integer outputs (labels, accumulators, RGBA8) must be bit-exact; centroids are compared
bit-exactly as well because the sums are exact integers and the update uses the same IEEE ops."""
import numpy as np
import pytest

from conftest import load_rgba, sorted_palette, set_strategy as _set_strategy

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream(torch):
    return torch.cuda.current_stream().cuda_stream


def test_library_is_native(processor):
    import kmeans_gpu_amd as kg
    assert kg.lib().kmg_version().decode().startswith("kmeans_hip")


def test_lab_all_16m_colours(torch_cuda, processor, oracle):
    """S1: every 24-bit colour converts to exactly the oracle's Lab (rgb_to_lab.wgsl)."""
    torch = torch_cuda
    n = 1 << 24
    idx = np.arange(n, dtype=np.uint32)
    rgba = np.empty((n, 4), np.uint8)
    rgba[:, 0] = idx & 255; rgba[:, 1] = (idx >> 8) & 255; rgba[:, 2] = (idx >> 16) & 255; rgba[:, 3] = 255
    d = _dev(torch, rgba)
    lab = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    processor.rgb_to_lab(d.data_ptr(), n, lab.data_ptr(), _stream(torch))
    torch.cuda.synchronize()
    got = lab.cpu().numpy()
    want = oracle.rgb_to_lab(rgba)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n,k,seed", [(1, 1, 1), (5, 3, 2), (1023, 16, 3), (1024, 16, 4), (4099, 7, 5),
                                      (300_000, 16, 0x5EED0002), (200_003, 256, 0x5EED0003),
                                      # k_assign's pixels per thread follow the image size: 1 below 2^19 pixels (above), 2, 4, 8
                                      (600_001, 12, 6), (1_300_003, 33, 7), (2_200_001, 5, 8)])
def test_assign_accumulate_matches_oracle(torch_cuda, processor, oracle, n, k, seed):
    """S2 + S4: labels bit-exact, int64 accumulators bit-exact, incl. ragged tails."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    rgba = oracle.synth_uniform(seed, n)
    lab = oracle.rgb_to_lab(rgba)
    cent = oracle.centroids4(lab[(np.arange(k) * (n // k)) % n])
    want_labels, want_acc = oracle.assign_accumulate_rgba(rgba, cent)
    d = _dev(torch, rgba)
    labels = torch.full((n,), -1, dtype=torch.int32, device="cuda")        # (0xFFFFFFFF: no pixel may keep it)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s = kg.Lloyd(processor, k)
    s.set_centroids(cent)
    s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), _stream(torch))
    torch.cuda.synchronize()
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels)
    assert np.array_equal(acc.cpu().numpy(), want_acc)
    # labels only / sums only variants
    labels2 = torch.zeros(n, dtype=torch.int32, device="cuda")
    s.assign_accumulate(d.data_ptr(), n, labels2.data_ptr(), 0, _stream(torch))
    acc2 = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s.assign_accumulate(d.data_ptr(), n, 0, acc2.data_ptr(), _stream(torch))
    torch.cuda.synchronize()
    assert torch.equal(labels, labels2) and torch.equal(acc, acc2)
    s.close()


@pytest.mark.parametrize("n", [3_000, 70_000, 600_000, 1_100_000])
def test_assign_near_ties_are_decided_by_the_literal_distance(torch_cuda, processor, oracle, n):
    """centroid tables full of (near-)duplicates: the ordering key cannot separate them, the literal CIE94 distance with the
    reference's strict `<` (find_centroid.wgsl:32-41: first minimum wins) must -- the repair of k_assign's scan, for every
    pixels-per-thread variant (1, 1, 2, 4 by image size) and for both the plain (k < 32) and the chunked scan"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    rng = np.random.default_rng(n)
    rgba = oracle.synth_uniform(900 + n, n)
    base = oracle.rgb_to_lab(rgba[:12])
    for k in (24, 40):
        lab = np.tile(base, (4, 1))[:k].astype(np.float32)
        # a few entries one or two ulps off their twin: near-ties that are NOT exact duplicates
        bump = lab[12:16].view(np.uint32) + np.array([1, 2, 1, 3], np.uint32)[:, None]
        lab[12:16] = bump.view(np.float32)
        cent = oracle.centroids4(lab)
        want_labels, want_acc = oracle.assign_accumulate_rgba(rgba, cent)
        d = _dev(torch, rgba)
        labels = torch.zeros(n, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        s = kg.Lloyd(processor, k)
        s.set_centroids(cent)
        s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), _stream(torch))
        torch.cuda.synchronize()
        assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels), k
        assert np.array_equal(acc.cpu().numpy(), want_acc), k
        s.close()


def test_assign_unaligned_band(torch_cuda, processor, oracle):
    """a row band whose first pixel is not 16-byte aligned takes the scalar-load path"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    n, k = 10_001, 5
    rgba = oracle.synth_uniform(77, n + 3)
    lab = oracle.rgb_to_lab(rgba)
    cent = oracle.centroids4(lab[:k])
    d = _dev(torch, rgba)
    labels = torch.zeros(n + 3, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    s = kg.Lloyd(processor, k)
    s.set_centroids(cent)
    for off in (1, 2, 3):
        s.assign_accumulate(d.data_ptr() + 4 * off, n, labels.data_ptr() + 4 * off, acc.data_ptr(), _stream(torch))
        torch.cuda.synchronize()
        wl, wa = oracle.assign_accumulate_rgba(rgba[off:off + n], cent)
        assert np.array_equal(labels.cpu().numpy().view(np.uint32)[off:off + n], wl)
        assert np.array_equal(acc.cpu().numpy(), wa)
    s.close()


def test_update_and_lloyd_loop(torch_cuda, processor, oracle, tokyo):
    """S5 + S6 on the reference-sized problem: tokyo shrunk to 256x171, k = 8."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    nw, nh = oracle.resized_dims(tokyo.shape[1], tokyo.shape[0])
    small = oracle.resize(tokyo, nw, nh)
    lab = oracle.rgb_to_lab(small)
    k = 8
    init = oracle.init_centroids(lab, nw, nh, k)
    want_c, want_labels, want_it = oracle.lloyd(lab, init)
    d = _dev(torch, small)
    n = nw * nh
    labels = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = kg.Lloyd(processor, k)
    s.set_centroids(init)
    it = s.run(d.data_ptr(), n, labels.data_ptr(), _stream(torch))
    got_c = s.get_centroids()
    assert it == want_it
    assert np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels)
    s.close()


@pytest.mark.parametrize("k,max_it,period,conv", [(1, 128, 8, 1.0), (5, 128, 8, 1.0), (33, 128, 8, 1.0), (64, 128, 8, 1.0),
                                                  (12, 5, 8, 1.0), (12, 9, 8, 1.0), (12, 128, 3, 1.0), (40, 128, 1, 0.05),
                                                  (7, 1, 8, 1.0), (300, 20, 4, 1.0)])
def test_lloyd_loop_on_a_small_image_one_launch_per_iteration(torch_cuda, oracle, k, max_it, period, conv):
    """modules.rs:763-840 on the reference's working size (an image of <= 65 536 pixels): kmg_lloyd_run takes one launch per
    iteration there (update in every workgroup's prologue, sums through rotating buffers).  Iteration count, centroids and
    labels equal the oracle's for iteration caps and check periods that end the loop in every position of the rotation."""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    rng = np.random.default_rng(k * 1000 + max_it)
    w, h = 211, 157
    c = rng.integers(0, 256, (9, 3))
    px = c[rng.integers(0, 9, w * h)] + rng.normal(0, 14.0, (w * h, 3))
    rgba = np.full((w * h, 4), 255, np.uint8)
    rgba[:, :3] = np.clip(np.rint(px), 0, 255).astype(np.uint8)
    lab = oracle.rgb_to_lab(rgba)
    init = oracle.init_centroids(lab, w, h, k)
    want_c, want_labels, want_it = oracle.lloyd(lab, init, max_iterations=max_it, check_period=period, convergence=conv)
    p = kg.ImageProcessor(max_iterations=max_it, check_period=period, convergence=conv)
    d = _dev(torch, rgba)
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    s = kg.Lloyd(p, k)
    s.set_centroids(init)
    it = s.run(d.data_ptr(), w * h, labels.data_ptr(), _stream(torch))
    assert it == want_it
    assert np.array_equal(s.get_centroids().view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels)
    # a second run on the same object starts from the centroids the first one left (and from clean sum buffers)
    want_c2, want_labels2, want_it2 = oracle.lloyd(lab, want_c, max_iterations=max_it, check_period=period, convergence=conv)
    it2 = s.run(d.data_ptr(), w * h, labels.data_ptr(), _stream(torch))
    assert it2 == want_it2
    assert np.array_equal(s.get_centroids().view(np.uint32), want_c2.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels2)
    s.close(); p.close()


@pytest.mark.parametrize("w,h,k", [(1000, 650, 6), (1400, 900, 33)])
def test_lloyd_loop_per_pixel_scan_on_mid_size_images(torch_cuda, oracle, monkeypatch, w, h, k):
    """kmg_lloyd_run with the per-pixel scan on images of 2^19 ... 2^21 pixels (2 and 4 pixels per thread, the partial-sum slab and the
    separate reduce / update launches): iteration count, centroids and labels equal the oracle's"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    _set_strategy("brute")
    rng = np.random.default_rng(w + k)
    c = rng.integers(0, 256, (11, 3))
    px = c[rng.integers(0, 11, w * h)] + rng.normal(0, 20.0, (w * h, 3))
    rgba = np.full((w * h, 4), 255, np.uint8)
    rgba[:, :3] = np.clip(np.rint(px), 0, 255).astype(np.uint8)
    lab = oracle.rgb_to_lab(rgba)
    init = oracle.centroids4(lab[rng.choice(w * h, k, replace=False)])
    want_c, want_labels, want_it = oracle.lloyd(lab, init, max_iterations=10, check_period=4)
    p = kg.ImageProcessor(shrink_max_dim=0, max_iterations=10, check_period=4)
    d = _dev(torch, rgba)
    labels = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    s = kg.Lloyd(p, k)
    s.set_centroids(init)
    it = s.run(d.data_ptr(), w * h, labels.data_ptr(), _stream(torch))
    assert it == want_it
    assert np.array_equal(s.get_centroids().view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(labels.cpu().numpy().view(np.uint32), want_labels)
    s.close(); p.close()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 33, 256])     # (the workgroups' keys travel through two slot sets)
def test_init_centroids(torch_cuda, processor, oracle, tokyo, k):
    """S12 farthest-point init incl. its arg-max tie rule"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    nw, nh = oracle.resized_dims(tokyo.shape[1], tokyo.shape[0])
    small = oracle.resize(tokyo, nw, nh)
    want = oracle.init_centroids(oracle.rgb_to_lab(small), nw, nh, k)
    d = _dev(torch, small)
    s = kg.Lloyd(processor, k)
    s.init_centroids(d.data_ptr(), nw, nh, _stream(torch))
    got = s.get_centroids(_stream(torch))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    s.close()


def test_init_ties_flat_image(torch_cuda, processor, oracle):
    """few distinct colours -> many exact ties in the distance map"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    w, h, k = 67, 41, 6
    rng = np.random.default_rng(5)
    pal = rng.integers(0, 256, (4, 4), dtype=np.uint8); pal[:, 3] = 255
    img = pal[rng.integers(0, 4, (h, w))]
    want = oracle.init_centroids(oracle.rgb_to_lab(img), w, h, k)
    d = _dev(torch, img)
    s = kg.Lloyd(processor, k)
    s.init_centroids(d.data_ptr(), w, h, _stream(torch))
    got = s.get_centroids(_stream(torch))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    s.close()


@pytest.mark.parametrize("shape", [(513, 768), (300, 1000), (1000, 300), (257, 256), (17, 4000)])
def test_resize(torch_cuda, processor, oracle, shape):
    """S11 shrink to <= 256"""
    import kmeans_gpu_amd as kg
    torch = torch_cuda
    h, w = shape
    img = oracle.synth_uniform(h * 1000 + w, h * w).reshape(h, w, 4)
    nw, nh = kg.resized_dims(w, h)
    assert (nw, nh) == oracle.resized_dims(w, h)
    want = oracle.resize(img, nw, nh)
    d = _dev(torch, img)
    out = torch.zeros((nh, nw, 4), dtype=torch.uint8, device="cuda")
    processor.resize(d.data_ptr(), w, h, nw, nh, out.data_ptr(), _stream(torch))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_find_goldens(processor, tokyo):
    """the reference's three committed `find` outputs (samples.sh:6-8), bit-exact"""
    import kmeans_gpu_amd as kg
    pal3 = np.array([[5, 5, 5, 255], [255, 255, 255, 255], [255, 0, 0, 255]], np.uint8)
    out = processor.find(tokyo, pal3, kg.ReduceMode.Replace)
    assert np.array_equal(out, load_rgba("tokyo-find-replace-dark-white-red.png"))
    out = processor.find(tokyo, pal3, kg.ReduceMode.Dither)
    assert np.array_equal(out, load_rgba("tokyo-find-dither-dark-white-red.png"))
    out = processor.find(tokyo, sorted_palette("apollo-1x.png"), kg.ReduceMode.Dither)
    assert np.array_equal(out, load_rgba("tokyo-find-dither-apollo.png"))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 8, 16])
def test_reduce_matches_oracle(processor, oracle, tokyo, k, mode):
    """ImageProcessor::reduce end to end (shrink, init, Lloyd, output pass) == oracle, bit-exact"""
    got = processor.reduce(k, tokyo, reduce_mode=mode)
    want = oracle.reduce(tokyo, k, mode)
    assert np.array_equal(got, want)


def test_reduce_golden_colours(processor, tokyo):
    """reduce -c 8 lands on the golden's 8 colours within 1 LSB (samples.sh:3)"""
    got = processor.reduce(8, tokyo)
    g = load_rgba("tokyo-reduce-c8-kmeans-replace.png")
    c1 = np.unique(got.reshape(-1, 4), axis=0).astype(int)
    c2 = np.unique(g.reshape(-1, 4), axis=0).astype(int)
    assert c1.shape == c2.shape == (8, 4)
    assert np.abs(c1 - c2).max() <= 1


def test_palette_matches_oracle(processor, oracle, tokyo):
    got = processor.palette(8, tokyo)
    want = oracle.palette(tokyo, 8)
    assert np.array_equal(got, want)
    gold = load_rgba("tokyo-palette-c8-kmeans-s40.png")[0, ::40, :]
    assert np.abs(got.astype(int) - gold.astype(int)).max() <= 1


def test_concurrent_calls_on_one_processor(processor, oracle, tokyo):
    """core/examples/parallel.rs:36-50: 14 threads share one ImageProcessor (k = 2..15)"""
    import threading
    img = tokyo[::2, ::2].copy()
    results, errors = {}, []

    def work(k):
        try:
            results[k] = processor.reduce(k, img, reduce_mode=k % 2)
        except Exception as e:      # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2, 16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for k in range(2, 16):
        assert np.array_equal(results[k], oracle.reduce(img, k, k % 2)), k


def test_error_behaviour(processor, tokyo):
    """cli/src/args.rs:160-171 (k >= 1) and the documented status codes"""
    import kmeans_gpu_amd as kg
    with pytest.raises(kg.KmgError) as e:
        processor.reduce(0, tokyo)
    assert e.value.status == -1 and "higher than 0" in str(e.value)
    with pytest.raises(kg.KmgError) as e:
        processor.reduce(4, tokyo, algo=7)
    assert e.value.status == -1
    with pytest.raises(kg.KmgError) as e:
        processor.find(tokyo, np.zeros((0, 4), np.uint8))
    assert e.value.status == -1
    with pytest.raises(kg.KmgError) as e:
        processor.palette(5000, tokyo)   # > KMG_MAX_K = 3072
    assert e.value.status == -5          # KMG_MAX_K


def test_single_pixel_and_single_row_images(processor, oracle):
    for shape in [(1, 1), (1, 37), (41, 1), (3, 5)]:
        img = oracle.synth_uniform(shape[0] * 100 + shape[1], shape[0] * shape[1]).reshape(shape[0], shape[1], 4)
        for k in (1, 2, 3):
            for mode in (0, 1):
                assert np.array_equal(processor.reduce(k, img, reduce_mode=mode), oracle.reduce(img, k, mode)), (shape, k, mode)
        pal = oracle.synth_uniform(5, 6)
        for mode in (0, 1):
            assert np.array_equal(processor.find(img, pal, mode), oracle.find(img, pal, mode))


def test_more_clusters_than_pixels(processor, oracle):
    """empty clusters keep their centroid and block convergence (choose_centroid.wgsl:192-194):
    the loop runs to MAX_ITERATION"""
    img = oracle.synth_uniform(8, 12).reshape(3, 4, 4)
    assert np.array_equal(processor.reduce(20, img), oracle.reduce(img, 20, 0))
    assert np.array_equal(processor.palette(20, img), oracle.palette(img, 20))


@pytest.mark.parametrize("k", [1, 2, 3, 8, 46])
def test_meld_matches_oracle(processor, oracle, tokyo, k):
    """ReduceMode::Meld (mix_colors.wgsl main_meld).  Lab -> sRGB8 needs pow(c, 1/2.4) per pixel: device,
    host and oracle evaluate the same fixed binary64 sequence (kmg_math.h pow_inv_2p4) -- byte equality."""
    img = tokyo[::3, ::3].copy()
    if k == 46:
        pal = sorted_palette("apollo-1x.png")
    else:
        pal = np.array(sorted(map(tuple, oracle.synth_uniform(k, k))), np.uint8)
    got = processor.find(img, pal, 2)
    want = oracle.find(img, pal, oracle.MODE_MELD)
    assert np.array_equal(got, want), f"{int((got != want).sum())} channels differ"
    assert np.all(got[..., 3] == 255)


def test_meld_encode_table_is_the_encode_for_every_float(processor):
    """the meld pass reads the sRGB8 byte of a linear channel value from a threshold table (made on the device by the
    encode itself): compared with the encode for EVERY float bit pattern of either sign, NaN aside"""
    assert processor.debug_encode_table_check() == 0


@pytest.mark.parametrize("c", [116.0, 500.0, 200.0, 100.0, 7.787, 95.0489, 108.8840])
def test_device_division_by_the_shader_constants_is_the_ieee_quotient(processor, c):
    """lab_to_rgb.wgsl:45-59 / rgb_to_lab.wgsl divide by constants; the device multiplies by the reciprocal and corrects once
    with the residual (kmg_math.h div_const / div_white).  Compared with x / c for EVERY float x"""
    bad, lo, hi = processor.debug_division_check(c)
    assert bad == 0, f"{bad} of 2^32 quotients differ, |x| bit patterns {lo:#x} .. {hi:#x}"


def test_reduce_meld_end_to_end(processor, oracle, tokyo):
    got = processor.reduce(6, tokyo, reduce_mode=2)
    want = oracle.reduce(tokyo, 6, oracle.MODE_MELD)
    assert np.array_equal(got, want), f"{int((got != want).sum())} channels differ"


@pytest.mark.parametrize("w,h", [(16385, 3), (5, 9001)])
def test_images_beyond_the_reference_texture_limit(processor, oracle, w, h):
    """the reference stops at 8192 px per side (README.md:9-11, wgpu texture limit); this library has no
    such limit: reduce / find of a 16385-wide and a 9001-tall image equal the oracle"""
    import kmeans_gpu_amd as kg
    rng = np.random.default_rng(w)
    img = _gradient_noise(rng, w, h)
    for mode, omode in ((kg.ReduceMode.Replace, oracle.MODE_REPLACE), (kg.ReduceMode.Dither, oracle.MODE_DITHER)):
        got = processor.reduce(5, img, kg.Algorithm.Kmeans, mode)
        assert np.array_equal(got, oracle.reduce(img, 5, omode))
    pal = np.array([[0, 0, 0, 255], [255, 255, 255, 255], [128, 64, 32, 255]], np.uint8)
    assert np.array_equal(processor.find(img, pal, kg.ReduceMode.Dither), oracle.find(img, pal, oracle.MODE_DITHER))


def _gradient_noise(rng, w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
    img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
    img[..., 2] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[..., 3] = 255
    return img


@pytest.mark.parametrize("seed", range(8))
def test_random_images_reduce_find_palette_match_oracle(processor, oracle, seed):
    """random small images of awkward shapes through the host-buffer API against the oracle, byte for byte"""
    import kmeans_gpu_amd as kg
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    kind = seed % 3
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    elif kind == 1:
        pal = rng.integers(0, 256, (int(rng.integers(1, 7)), 4), dtype=np.uint8)
        img = pal[rng.integers(0, pal.shape[0], (h, w))]
    else:
        img = _gradient_noise(rng, w, h)
    # (random alpha stays: ignored on input -- rgb_to_lab.wgsl:78, filtered but unused by resize.wgsl -- and 255 on output)
    k = int(rng.choice([1, 2, 3, 6, 12, 40]))
    for mode, omode in ((kg.ReduceMode.Replace, oracle.MODE_REPLACE), (kg.ReduceMode.Dither, oracle.MODE_DITHER)):
        assert np.array_equal(processor.reduce(k, img, kg.Algorithm.Kmeans, mode), oracle.reduce(img, k, omode)), (w, h, k, mode)
    assert np.array_equal(processor.palette(k, img, kg.Algorithm.Kmeans), oracle.palette(img, k))
    colors = rng.integers(0, 256, (int(rng.integers(1, 20)), 4), dtype=np.uint8); colors[:, 3] = 255
    for mode, omode in ((kg.ReduceMode.Replace, oracle.MODE_REPLACE), (kg.ReduceMode.Dither, oracle.MODE_DITHER)):
        assert np.array_equal(processor.find(img, colors, mode), oracle.find(img, colors, omode)), (w, h, mode)


@pytest.mark.parametrize("mode", ["Replace", "Dither", "Meld"])
def test_apply_plan_in_bands_on_two_streams_equals_the_whole_pass(torch_cuda, processor, oracle, mode):
    """kmg_apply_plan_*: tables built once, the image processed in uneven row bands alternating between two streams, no host
    synchronisation in between -- the bytes equal kmg_dev_apply's on the whole image (and hence the oracle's: the Bayer index
    uses image rows, find_centroid / mix_colors.wgsl)."""
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    w, h, k = 2048, 1536, 40
    n = w * h
    img = synth.uniform_rgba_torch(31337, n, device="cuda")
    rng = np.random.default_rng(9)
    pal = np.full((k, 4), 255, np.uint8); pal[:, :3] = rng.integers(0, 256, (k, 3))
    cent = kg.palette_to_centroids(pal)
    m = getattr(kg.ReduceMode, mode)
    st = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    processor.apply(img.data_ptr(), w, h, 0, cent, m, whole.data_ptr(), st)
    want = oracle.find(img.cpu().numpy().reshape(h, w, 4)[:64], pal, getattr(oracle, "MODE_" + mode.upper()))
    assert np.array_equal(whole.cpu().numpy().reshape(h, w, 4)[:64], want)
    out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    plan = processor.apply_plan(cent, m, n, st)
    bounds = [0, 100, 101, 640, 1203, h]
    for i, (r0, r1) in enumerate(zip(bounds[:-1], bounds[1:])):
        s = streams[i % 2]
        plan.run(img[r0 * w:].data_ptr(), w, r1 - r0, r0, out[r0 * w:].data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    plan.close()
    assert torch.equal(out, whole)
