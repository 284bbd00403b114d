"""The C-ABI library loads and exports every symbol include/kmeans_hip.h declares; host-only entry
points agree with the oracle; without a HIP device the processor fails loudly (no CPU path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, sorted_palette


def _declared():
    text = open(os.path.join(ROOT, "include", "kmeans_hip.h")).read()
    return sorted(set(re.findall(r"KMG_API\s+[\w\s\*]+?\b(kmg_\w+)\s*\(", text)))


def test_header_symbols_exported():
    import kmeans_gpu_amd as kg
    names = _declared()
    assert len(names) >= 25
    L = C.CDLL(kg.library_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(kg.SYMBOLS) == names


def test_library_contains_gfx950_code_object():
    import kmeans_gpu_amd as kg
    blob = open(kg.library_path(), "rb").read()
    assert b"gfx950" in blob
    assert b"k_assign" in blob and b"k_apply" in blob and b"k_update" in blob


def test_default_options_are_the_reference_constants():
    import kmeans_gpu_amd as kg
    o = kg.default_options()
    assert (o.shrink_max_dim, o.max_iterations, o.check_period, o.convergence) == (256, 128, 8, 1.0)


def test_no_device_fails_loudly():
    """On a box without a GPU the constructor must raise (there is no fallback)."""
    import torch
    import kmeans_gpu_amd as kg
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(kg.KmgError) as e:
        kg.ImageProcessor()
    assert e.value.status == -2 and "no CPU path" in str(e.value)


def test_null_and_bad_arguments():
    import kmeans_gpu_amd as kg
    L = kg.lib()
    assert L.kmg_processor_create(None) == -1
    assert b"NULL" in L.kmg_last_error()
    assert L.kmg_palette_to_centroids(None, 3, None) == -1
    t = C.c_float()
    c = np.zeros((1, 4), np.float32)
    assert L.kmg_dither_threshold(C.c_void_p(c.ctypes.data), 1, C.byref(t)) == -1   # needs k >= 2
    assert L.kmg_reduce(None, None, 0, 0, 0, 0, 0, None) == -1


def test_host_colour_helpers_match_oracle(oracle):
    import kmeans_gpu_amd as kg
    pal = sorted_palette("resurrect_64.png")
    got = kg.palette_to_centroids(pal)
    want = np.array([oracle.palette_srgb8_to_lab(c[:3]) for c in pal])
    assert np.array_equal(got[:, :3].view(np.uint32), want.view(np.uint32))
    assert np.all(got[:, 3] == 1.0)
    back = kg.centroids_to_palette(got)
    want_back = np.array([oracle.palette_lab_to_srgb8(c[:3]) for c in got])
    assert np.array_equal(back[:, :3], want_back) and np.all(back[:, 3] == 255)
    assert np.array_equal(back[:, :3], pal[:, :3])          # sRGB8 -> Lab -> sRGB8 round trip
    assert kg.dither_threshold(got) == oracle.dither_threshold(got)


def test_resized_dims_match_oracle(oracle):
    import kmeans_gpu_amd as kg
    rng = np.random.default_rng(1)
    for w, h in [(768, 513), (8192, 8192), (257, 1), (1, 257), (256, 256), (3184, 2126)] + \
                [tuple(int(v) for v in rng.integers(1, 9000, 2)) for _ in range(200)]:
        assert kg.resized_dims(w, h) == oracle.resized_dims(w, h)


def test_synth_generators_agree(oracle):
    from kmeans_gpu_amd import synth
    a = oracle.synth_uniform(synth.SEED_CFG3, 70001)
    assert np.array_equal(a, synth.uniform_rgba_numpy(synth.SEED_CFG3, 70001))
    assert np.array_equal(a, synth.uniform_rgba_torch(synth.SEED_CFG3, 70001, device="cpu", chunk=9999).numpy())
    assert np.array_equal(a[60000:], synth.uniform_rgba_numpy(synth.SEED_CFG3, 10001, first=60000))
