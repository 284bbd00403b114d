"""Pins the CPU oracle against everything the reference commits for this path: the golden images
of samples.sh (gfx/*.png, copied to tests/golden/) and the known-answer values of
core/src/shader_tests.rs.  No GPU."""
import numpy as np
import pytest

from conftest import load_rgba, sorted_palette

PAL3 = np.array([[5, 5, 5, 255], [255, 255, 255, 255], [255, 0, 0, 255]], np.uint8)   # samples.sh:6-7
GOLD8 = ["#12100E", "#2E221E", "#602B1C", "#515346", "#AF2C1B", "#CB7550", "#869791", "#E0E1D7"]


def _hex(s):
    return [int(s[1:3], 16), int(s[3:5], 16), int(s[5:7], 16)]


def test_find_replace_golden_bit_exact(oracle, tokyo):
    out = oracle.find(tokyo, PAL3, oracle.MODE_REPLACE)
    g = load_rgba("tokyo-find-replace-dark-white-red.png")
    assert np.array_equal(out, g)
    # 334541 / 40844 / 18599 pixels of #050505 / #FFFFFF / #FF0000
    cols, cnt = np.unique(out.reshape(-1, 4), axis=0, return_counts=True)
    assert dict(zip(map(tuple, cols[:, :3]), cnt)) == {(5, 5, 5): 334541, (255, 255, 255): 40844, (255, 0, 0): 18599}


def test_find_dither_golden_bit_exact(oracle, tokyo):
    out = oracle.find(tokyo, PAL3, oracle.MODE_DITHER)
    assert np.array_equal(out, load_rgba("tokyo-find-dither-dark-white-red.png"))


def test_find_dither_apollo_golden_bit_exact(oracle, tokyo):
    pal = sorted_palette("apollo-1x.png")
    assert len(pal) == 46
    out = oracle.find(tokyo, pal, oracle.MODE_DITHER)
    assert np.array_equal(out, load_rgba("tokyo-find-dither-apollo.png"))


def test_resurrect_palette_has_64_colours():
    """cli/src/args.rs:282-292 test_parse_palette"""
    assert len(sorted_palette("resurrect_64.png")) == 64


@pytest.mark.parametrize("mode,name", [(0, "tokyo-reduce-c8-kmeans-replace.png"), (1, "tokyo-reduce-c8-kmeans-dither.png")])
def test_reduce_c8_golden(oracle, tokyo, mode, name):
    """samples.sh:3-4.  The author's GPU used its own pow/sin/bilinear/f32-sum order, so the gate is:
    same 8 colours within 1 LSB, >= 99.5 % identical labels."""
    out = oracle.reduce(tokyo, 8, mode)
    g = load_rgba(name)
    c1, l1 = np.unique(out.reshape(-1, 4), axis=0, return_inverse=True)
    c2, l2 = np.unique(g.reshape(-1, 4), axis=0, return_inverse=True)
    assert len(c1) == len(c2) == 8
    assert np.abs(c1.astype(int) - c2.astype(int)).max() <= 1
    assert (l1.reshape(-1) == l2.reshape(-1)).mean() >= 0.995
    gold = np.array(sorted(_hex(h) for h in GOLD8))
    assert np.abs(np.array(sorted(map(list, c1[:, :3].astype(int)))) - gold).max() <= 1


def test_palette_c8_golden(oracle, tokyo):
    """samples.sh:5: eight 40x40 squares sorted by Lab L"""
    got = oracle.palette(tokyo, 8)
    gold = load_rgba("tokyo-palette-c8-kmeans-s40.png")
    assert gold.shape == (40, 320, 4)
    assert np.abs(got.astype(int) - gold[0, ::40].astype(int)).max() <= 1
    L = [oracle.palette_srgb8_to_lab(c[:3])[0] for c in got]
    assert L == sorted(L)


def test_cie94_kat(oracle):
    """core/src/shader_tests.rs:169-186: cie94(Lab(255,0,0), Lab(255,128,0)) = 19.094658 +- 0.01,
    Lab from the palette crate; the distance is asymmetric (delta_e.wgsl:17-18)."""
    a = oracle.palette_srgb8_to_lab([255, 0, 0])
    b = oracle.palette_srgb8_to_lab([255, 128, 0])
    assert abs(oracle.cie94(a, b) - 19.094658) < 0.01
    assert abs(oracle.cie94(b, a) - 20.300905) < 0.01
    assert abs(np.sqrt(oracle.cie94_key(a, b)) - 19.094658) < 0.01


def test_lab_kats(oracle):
    """shader Lab vs palette-crate Lab of the same colour differ (white point)"""
    red = np.array([[255, 0, 0, 255]], np.uint8)
    assert np.allclose(oracle.rgb_to_lab(red)[0], [53.24079, 80.08996, 67.203354], atol=2e-4)
    assert np.allclose(oracle.palette_srgb8_to_lab([255, 0, 0]), [53.2408, 80.09243, 67.20321], atol=2e-4)
    assert np.allclose(oracle.palette_srgb8_to_lab([5, 5, 5]), [1.3708744, 0, 0], atol=2e-4)
    assert np.allclose(oracle.palette_srgb8_to_lab([255, 255, 255]), [100, 0, 0], atol=2e-4)


def test_rand_constants(oracle):
    """plus_plus_init.wgsl:58-60,163-164 in IEEE binary32"""
    assert oracle.rand(42.0) == 0.5625
    assert oracle.rand(12.0) == 0.93359375


def test_init_pixel_and_iterations(oracle, tokyo):
    """SURVEY.md section 4 probe: 256x171 shrink, init pixel (144,159), stop at the iteration-16 check"""
    assert oracle.resized_dims(768, 513) == (256, 171)
    small = oracle.resize(tokyo, 256, 171)
    lab = oracle.rgb_to_lab(small)
    c0 = oracle.init_centroids(lab, 256, 171, 1)[0, :3]
    assert np.array_equal(c0, lab.reshape(171, 256, 3)[159, 144])
    cent, it = oracle.extract_palette_kmeans(tokyo, 8)
    assert it == 16
    L = np.sort(cent[:, 0])
    assert np.allclose(L, [5.014, 14.631, 24.634, 34.696, 39.511, 58.139, 61.237, 89.392], atol=0.02)


def test_hoisted_literal_equals_plain_literal_and_key_is_only_a_filter(oracle, tokyo):
    """The arg-min is the reference's: first minimum of the literal distance_cie94 (find_centroid.wgsl:32-41).
    The oracle's fast path hoists C1, SC, SH, C2 out of the loop -- same floats, so the same labels as
    orc_cie94 per pair, everywhere.  Ordering by the squared key alone (what the GPU kernels do BEFORE
    their near-tie repair) agrees on the fixtures but not on every colour: the grey-axis palette below has
    colours whose two nearest centroids tie in the literal distance but not in the key."""
    lab = oracle.rgb_to_lab(tokyo)
    for cent in (oracle.centroids4(np.array([oracle.palette_srgb8_to_lab(c[:3]) for c in sorted_palette("apollo-1x.png")])),
                 oracle.extract_palette_kmeans(tokyo, 8)[0]):
        want = oracle.assign(lab, cent, literal=1)
        assert np.array_equal(want, oracle.assign(lab, cent, literal=0))
        assert np.array_equal(want, oracle.assign(lab, cent, literal=2))
    px = oracle.synth_uniform(0x5EED0003, 1 << 16)
    lab = oracle.rgb_to_lab(px)
    cent = oracle.centroids4(lab[::256])
    assert np.array_equal(oracle.assign(lab, cent, literal=1), oracle.assign(lab, cent, literal=0))
    # a slab of the colour cube against 64 greys: the key alone mislabels a few colours, the hoisted literal none
    idx = np.arange(1 << 20, dtype=np.uint32)
    cube = np.empty((1 << 20, 4), np.uint8)
    cube[:, 0] = idx & 255; cube[:, 1] = (idx >> 8) & 255; cube[:, 2] = (idx >> 16) & 255; cube[:, 3] = 255
    lab = oracle.rgb_to_lab(cube)
    grey = oracle.centroids4(oracle.rgb_to_lab(np.repeat(np.arange(0, 256, 4, dtype=np.uint8)[:, None], 4, 1)))
    want = oracle.assign(lab, grey, literal=1)
    assert np.array_equal(want, oracle.assign(lab, grey, literal=0))
    assert 0 < int((want != oracle.assign(lab, grey, literal=2)).sum()) < 200


def test_pow_inv_2p4_is_the_rounded_double_pow(oracle):
    """lab_to_rgb.wgsl:21-35: the oracle's fixed evaluation of pow(c, f32(1/2.4)) agrees with a correctly
    rounded binary64 pow rounded to binary32 (numpy / libm) on all but a vanishing share of inputs, and is
    within 1 ulp always; shader_tests.rs:231-240's pow KAT style check (pow(2.1, 7) there) is about the WGSL
    builtin -- here the function itself is pinned."""
    rng = np.random.default_rng(7)
    c = np.concatenate([rng.uniform(0.0031308, 1.0, 200000), np.linspace(0.0031309, 0.99999, 50000)]).astype(np.float32)
    got = np.array([oracle.pow_inv_2p4(float(v)) for v in c[:20000]], np.float32)
    want = np.power(c[:20000].astype(np.float64), np.float64(np.float32(1.0) / np.float32(2.4))).astype(np.float32)
    ulp = np.abs(got.view(np.int32) - want.view(np.int32))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
    assert oracle.pow_inv_2p4(1.0) == 1.0 and oracle.pow_inv_2p4(7.5) == 1.0


def test_shrunk_dims_rule(oracle):
    """core/src/structures.rs:79-89"""
    assert oracle.resized_dims(3184, 2126) == (256, 170)
    assert oracle.resized_dims(100, 5000) == (5, 256)
    assert oracle.resized_dims(5000, 3) == (256, 1)
    assert oracle.resized_dims(300, 300) == (256, 256)


def test_init_argmax_is_independent_of_the_thread_count(oracle, tokyo):
    """orc_init_centroids folds the arg-max per OpenMP thread and then across threads with the reference's tie
    rule (plus_plus_init.wgsl:62-68: a later thread / workgroup wins a tie): the picks must not depend on how many
    threads share the fold.  Few colours -> many exact ties at the maximum; an image of ONE colour -> every
    distance 0 -> Candidate(0, 0.0), i.e. pixel 0, k times."""
    img = tokyo[100:180, 200:331].copy()                            # 80 x 131: not a multiple of 16 pixels
    img[..., :3] &= 0xC0                                            # 64 colours at most
    flat = np.zeros((33, 47, 4), np.uint8); flat[...] = (10, 200, 30, 255)
    threads = oracle.num_threads()
    try:
        for im, k in ((img, 12), (flat, 4)):
            h, w = im.shape[:2]
            lab = oracle.rgb_to_lab(im)
            got = []
            for t in (1, 2, 3, 7, 8):
                oracle.set_num_threads(t)
                got.append(oracle.init_centroids(lab, w, h, k))
            for g in got[1:]:
                assert np.array_equal(g.view(np.uint32), got[0].view(np.uint32))
    finally:
        oracle.set_num_threads(threads)
