"""The multi-device layer of the C ABI (kmg_group_*, include/kmeans_hip.h): ImageProcessor::new (core/src/lib.rs:38-65) over a
device list.  The checks are a torch-free C++ program (tests/native/check_group.cpp) that runs the same image through ONE
processor and through a group and compares bytes: a one-rank group that really loads RCCL and issues every collective
(ncclAllReduce / ncclAllGather on the compute stream), and groups of two and three ranks that share the box's one GPU
through the library's loopback exchange (RCCL takes one rank per device)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "kmeans-gpu_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "check_group")
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "native", "check_group.cpp"), "-o", exe, "-L", LIBDIR, "-lkmeans_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_group_compiles_against_the_header_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    r = subprocess.run([exe, "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert r.stdout.startswith("error -2 ") and "no CPU path" in r.stdout


def test_group_binding_argument_checks():
    import kmeans_gpu_amd as kg
    L = kg.lib()
    assert L.kmg_group_create(None, None) == -1
    o = kg.GroupOptions()
    L.kmg_default_group_options(o)
    assert o.struct_size == kg.C.sizeof(kg.GroupOptions) and o.n_devices == 0 and o.flags == 0
    assert (o.processor.shrink_max_dim, o.processor.max_iterations, o.processor.check_period) == (256, 128, 8)
    o.struct_size = 12
    h = kg.C.c_void_p()
    assert L.kmg_group_create(kg.C.byref(o), kg.C.byref(h)) == -1 and b"struct_size" in L.kmg_last_error()
    assert L.kmg_group_lloyd_step(None) == -1 and L.kmg_group_lloyd_run(None, None) == -1
    assert L.kmg_group_processor(None, 0) is None and L.kmg_group_stream(None, 0) is None


def _run(exe, *args, env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("CHECK_GROUP_STRATEGY", None)
    e.update(env or {})
    r = subprocess.run([exe, "run", *map(str, args)], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith("ok group of")
    return r.stdout


@pytest.mark.gpu
def test_one_rank_group_with_forced_rccl_collectives_equals_one_processor(tmp_path, torch_cuda):
    """kmg_lloyd_run against kmg_group_lloyd_run with ncclAllReduce on the compute stream between assign and update (and
    beside the label pass, and with the cell-sharded cube pass's histogram all-reduce + in-place all-gather), bit for bit;
    palette / find / reduce through the group = the single-device calls."""
    exe = _build(tmp_path)
    out = _run(exe, 1, 16, 1536, 1024)
    assert "rccl version" in out and "rccl version 0" not in out
    out = _run(exe, 1, 200, 1200, 900, env={"CHECK_GROUP_STRATEGY": "table"})
    assert "strategy of rank 0: table" in out


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_of_one_process_sharing_the_gpu_equal_one_processor(tmp_path, torch_cuda, ranks):
    exe = _build(tmp_path)
    _run(exe, ranks, 16, 1536, 1025)                               # uneven bands
    out = _run(exe, ranks, 64, 1100, 960, env={"CHECK_GROUP_STRATEGY": "table"})
    assert "strategy of rank 0: table" in out


@pytest.mark.gpu
def test_group_python_binding_matches_image_processor(torch_cuda, processor, tokyo):
    import kmeans_gpu_amd as kg
    img = np.ascontiguousarray(tokyo[:400, :600])
    with kg.Group(devices=[0, 0], flags=kg.GROUP_LOOPBACK) as g:
        assert (g.n_local, g.first_rank, g.world, g.rccl_version) == (2, 0, 2, 0)
        assert np.array_equal(g.palette(8, img), processor.palette(8, img))
        assert np.array_equal(g.reduce(8, img, reduce_mode=kg.ReduceMode.Dither), processor.reduce(8, img, reduce_mode=kg.ReduceMode.Dither))
        pal = processor.palette(5, img)
        assert np.array_equal(g.find(img, pal, kg.ReduceMode.Meld), processor.find(img, pal, kg.ReduceMode.Meld))
        outs = g.reduce_batch(6, [img, img[:100], img[50:300, 10:200]])
        for o, im in zip(outs, [img, img[:100], img[50:300, 10:200]]):
            assert np.array_equal(o, processor.reduce(6, np.ascontiguousarray(im)))
