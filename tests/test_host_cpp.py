"""The C++ host mirror of the reference API (kmeans-gpu_amd/host/kmeans_color_gpu.hpp) compiled against
libkmeans_hip.so: fails loudly without a device (CPU), and returns what the ctypes binding and the
oracle return (GPU)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "kmeans-gpu_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "check_host_api")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "kmeans-gpu_amd", "host"),
                    os.path.join(ROOT, "tests", "native", "check_host_api.cpp"), "-o", exe,
                    "-L", LIBDIR, "-lkmeans_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    r = subprocess.run([exe, "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert r.stdout.startswith("error ") and len(r.stdout.split(None, 2)[2].strip()) > 0


@pytest.mark.gpu
def test_cpp_mirror_matches_binding_and_oracle(tmp_path, processor, oracle, tokyo):
    import kmeans_gpu_amd as kg
    exe = _build(tmp_path)
    img = np.ascontiguousarray(tokyo[100:400, 200:633])            # 300 x 433 crop
    h, w = img.shape[:2]
    raw = tmp_path / "in.rgba"
    img.tofile(raw)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, "run", str(raw), str(w), str(h), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 3" in r.stdout and "octree meld" in r.stdout and "device list ok" in r.stdout
    data = np.fromfile(out, np.uint8)
    pos = 0

    def take(nbytes):
        nonlocal pos
        v = data[pos:pos + nbytes]
        pos += nbytes
        return v

    for algo in (kg.Algorithm.Kmeans, kg.Algorithm.Octree):
        n = int(take(4).view(np.uint32)[0])
        pal = take(4 * n).reshape(n, 4)
        assert np.array_equal(pal, processor.palette(8, img, algo))
    colors = np.array([[0, 0, 0, 255], [255, 255, 255, 255], [200, 30, 30, 255], [30, 60, 200, 255]], np.uint8)
    for mode, omode in ((kg.ReduceMode.Replace, oracle.MODE_REPLACE), (kg.ReduceMode.Dither, oracle.MODE_DITHER),
                        (kg.ReduceMode.Meld, None)):
        got = take(w * h * 4).reshape(h, w, 4)
        assert np.array_equal(got, processor.find(img, colors, mode))
        if omode is not None:
            assert np.array_equal(got, oracle.find(img, colors, omode))
    got = take(w * h * 4).reshape(h, w, 4)
    assert np.array_equal(got, processor.reduce(8, img, kg.Algorithm.Kmeans, kg.ReduceMode.Dither))
    assert np.array_equal(got, oracle.reduce(img, 8, oracle.MODE_DITHER))
    got = take(w * h * 4).reshape(h, w, 4)
    assert np.array_equal(got, processor.reduce(8, img, kg.Algorithm.Octree, kg.ReduceMode.Replace))
    assert pos == data.size
