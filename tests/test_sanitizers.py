"""the CPU oracle and the product's HOST-ONLY code (csrc/kmg_octree.h, kmg_color.h, kmg_math.h) under
AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle asan` builds tests/native/check_sanitized.cpp with
oracle/kmg_oracle.c into one instrumented executable that makes every call the golden tests make, on small ragged inputs
(GPU sanitizers are not available on this pool: the device code is covered by the exhaustive kmg_debug_check_* passes instead)."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("threads", ["1", "4"])
def test_oracle_and_host_code_under_asan_ubsan(threads):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, OMP_NUM_THREADS=threads, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "check_sanitized")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
