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


CSRC = os.path.join(ROOT, "kmeans-gpu_amd", "csrc")
# pure arithmetic / a stored pointer: nothing in them can throw
_NO_THROW = {"kmg_last_error", "kmg_version", "kmg_init_first_key", "kmg_kernel_name", "kmg_group_processor", "kmg_group_stream"}


def test_every_entry_point_is_a_function_try_block():
    """No C++ exception crosses the C ABI (include/kmeans_hip.h "Conventions"): every extern "C" definition in csrc/ is
    `try { ... } KMG_ABI_CATCH*` -- the reference returns anyhow::Result (core/src/lib.rs:38)."""
    defined, unguarded = set(), []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        lines = open(os.path.join(CSRC, f)).read().split("\n")
        for i, ln in enumerate(lines):
            if not ln.lstrip().startswith('extern "C"') or ln.rstrip().endswith(";"):
                continue
            name = re.search(r"\b(kmg_\w+|name)\s*\(", ln).group(1)
            j = i
            while ")" not in lines[j]:
                j += 1
            body = lines[j + 1].strip() if not re.search(r"\)\s*(try\s*)?\{", lines[j]) else lines[j]
            if name == "name":                                     # the KMG_GROUP_CALL macro of kmg_group.hip
                assert body.startswith("try {"), f"{f}:{i + 1}"
                defined |= set(re.findall(r"^KMG_GROUP_CALL\((kmg_\w+),", "\n".join(lines), re.M))
                continue
            defined.add(name)
            if name not in _NO_THROW and "try" not in body.split("{")[0]:
                unguarded.append(f"{f}:{i + 1} {name}")
    assert not unguarded, unguarded
    # (kmg_tools_*: entry points of the tools build only, -DKMG_TOOLS -- not in the header, not in the product library)
    assert {d for d in defined if not d.startswith("kmg_tools_")} == set(_declared())
    for f in ("kmg_api.hip", "kmg_apply.hip", "kmg_group.hip", "kmg_lloyd.hip", "kmg_processor.hip"):
        text = open(os.path.join(CSRC, f)).read()
        assert text.count("\ntry {") + text.count("\ntry { ") >= 1
        assert len(re.findall(r"^try \{", text, re.M)) == len(re.findall(r"^KMG_ABI_CATCH", text, re.M)), f


_OOM_CHILD = r"""
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import kmeans_gpu_amd as kg
L = kg.lib()
L.kmg_octree_palette.restype = C.c_int
# 4 M distinct colours: the octree needs ~ 5 M nodes of 80 bytes -- far more than the address space left below
n = 1 << 22
px = np.zeros((n, 4), np.uint8)
i = np.arange(n, dtype=np.uint32) * 4
px[:, 0], px[:, 1], px[:, 2], px[:, 3] = i & 255, (i >> 8) & 255, (i >> 16) & 255, 255
out = np.zeros((8, 4), np.uint8)
cnt = C.c_uint32(0)
vm = int(next(l for l in open("/proc/self/status") if l.startswith("VmSize")).split()[1]) * 1024
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (48 << 20), hard))
rc = L.kmg_octree_palette(C.c_void_p(px.ctypes.data), C.c_uint64(n), 8, C.c_void_p(out.ctypes.data), C.byref(cnt))
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
msg = L.kmg_last_error().decode()
print("RC", rc, msg)
# the library is still usable afterwards
rc2 = L.kmg_octree_palette(C.c_void_p(px.ctypes.data), C.c_uint64(1000), 8, C.c_void_p(out.ctypes.data), C.byref(cnt))
print("RC2", rc2, cnt.value)
"""


def test_host_allocation_failure_is_a_status_not_an_abort(tmp_path):
    """std::bad_alloc inside an entry point (the octree's node vector under a tight RLIMIT_AS) must come back as
    KMG_ERR_OUT_OF_MEMORY through the C ABI -- not as an exception unwinding into ctypes / Rust (process abort)."""
    import subprocess
    import sys
    child = tmp_path / "oom_child.py"
    child.write_text(_OOM_CHILD)
    r = subprocess.run([sys.executable, str(child), os.path.join(ROOT, "kmeans-gpu_amd", "python")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "RC -4 " in r.stdout and "bad_alloc" in r.stdout, r.stdout
    assert "RC2 0 8" in r.stdout, r.stdout
