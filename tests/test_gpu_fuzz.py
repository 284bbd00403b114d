"""Randomised cross-check of the strategy families (tools/fuzz_parity.py): colour-table / pruned paths against
the per-pixel scans on random images, sizes, k and palettes -- init, Lloyd run and the three output modes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_problems_agree_across_strategies(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "40", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "40 cases, 0 mismatching" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "2"])
def test_alternative_cube_passes_of_small_centroid_tables(mode):
    """KMG_CUBE_SMALL = 0 (the general three-launch pass for k <= 32 too) and 2 (k_cube_small as the stage only, then the general
    scan and entries launches): the A/B switches of round 4 stay exact -- same random problems, fresh process (the switch is read once)."""
    env = dict(os.environ, KMG_CUBE_SMALL=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "24", "21"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "24 cases, 0 mismatching" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_random_default_calls_equal_the_oracle(seed):
    """tools/fuzz_default_call.py: kmg_reduce / kmg_palette at the reference's defaults (shrink to <= 256, init, Lloyd loop, output
    pass -- lib.rs:116-164) of random images, k and modes against oracle.reduce / oracle.palette, byte for byte"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_default_call.py"), "30", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "30 cases, 0 mismatching" in r.stdout
