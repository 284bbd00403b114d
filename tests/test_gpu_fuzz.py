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
