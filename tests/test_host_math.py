"""Host-side checks of the product's arithmetic header (kmeans-gpu_amd/csrc/kmg_math.h), no GPU."""
import os
import subprocess

import numpy as np

from conftest import ROOT


def test_kmg_math_matches_oracle_and_libm(oracle, tmp_path):
    """cbrt_cr correctly rounded on [1e-3, 2]; Lab of all 2^24 colours, cie94 and the arg-min key
    bit-identical to the oracle (tests/native/check_math.cpp)."""
    exe = str(tmp_path / "check_math")
    csrc = os.path.join(ROOT, "kmeans-gpu_amd", "csrc")
    # kmg_oracle.c is C99: compile it separately to keep its language mode
    obj = str(tmp_path / "oracle.o")
    subprocess.run(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp", "-c",
                    os.path.join(ROOT, "oracle", "kmg_oracle.c"), "-o", obj], check=True)
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp",
           "-I", csrc, "-I", os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "native", "check_math.cpp"), obj, "-o", exe, "-lm"]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    out = dict(line.split() for line in r.stdout.strip().splitlines())
    assert out == {"cbrt_mismatches": "0", "lut_mismatches": "0", "lab_mismatches": "0",
                   "distance_mismatches": "0"}, r.stdout
    assert r.returncode == 0


def test_oracle_cbrt_is_correctly_rounded_vs_long_double(oracle):
    """(float)cbrt((double)x) == (float)cbrtl(x) on a dense sample (the exhaustive run is in
    tests/native/check_math.cpp against the product routine)."""
    xs = np.concatenate([np.linspace(0.008856, 1.01, 200001), np.geomspace(1e-3, 2.0, 100001)]).astype(np.float32)
    got = np.array([oracle.cbrt(float(x)) for x in xs[::37]], np.float32)
    want = np.cbrt(xs[::37].astype(np.longdouble)).astype(np.float32)
    assert np.array_equal(got, want)


def test_colour_index_equals_its_plain_form_for_every_colour(tmp_path):
    """kmg_table.h colour_index gathers the bits of a pixel with two 24-bit multiplies (five vector instructions fewer per pixel in
    the label pass): equal to the bit-by-bit form, alpha ignored, and inverted by index_to_rgb -- all 2^24 colours
    (tests/native/check_colour_index.cpp, host build of the same header)."""
    exe = str(tmp_path / "check_colour_index")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                    "-I", os.path.join(ROOT, "kmeans-gpu_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "check_colour_index.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "colour_index_mismatches 0" in r.stdout and "inverse_mismatches 0" in r.stdout, r.stdout
