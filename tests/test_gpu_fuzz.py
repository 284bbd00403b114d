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


# every switch the TOOLS build of the library reads (kmg_internal.h KMG_TOOLS_ENV), set to a value that changes a launch shape there
# (the knock-outs of rounds 2-5 -- kernel variants that skip work and return wrong results -- are patches under tools/experiments/)
TOOLS_SWITCHES = {"KMG_ASSIGN_PPT": "1", "KMG_HOT_CELLS": "0", "KMG_CUBE_REPL": "1", "KMG_CUBE_SMALL": "0", "KMG_DITHER_SORT": "0", "KMG_BALANCE": "0",
                  "KMG_CUBE_GRID": "7", "KMG_SCAN_GRID": "5", "KMG_PAIRS_GRID": "3", "KMG_SMALL_GRID": "9", "KMG_DITHER_STATS": "1",
                  "KMG_SPLIT_LONG": "0"}


def test_product_library_does_not_contain_the_tools_switches():
    """`strings lib/libkmeans_hip.so | grep -c KNOCK` is 0: the knock-outs (kernel variants that return WRONG results) and the
    tuning switches are compiled into lib/libkmeans_hip_tools.so only (make tools, -DKMG_TOOLS)"""
    blob = open(os.path.join(ROOT, "kmeans-gpu_amd", "lib", "libkmeans_hip.so"), "rb").read()
    assert b"KNOCK" not in blob
    for name in TOOLS_SWITCHES:
        assert name.encode() not in blob, name
    for name in (b"KMG_LOG", b"KMG_RCCL_LIBRARY"):                            # what the product does read: logging, the RCCL path
        assert name in blob
    for name in (b"KMG_STRATEGY", b"KMG_DITHER_LISTS"):                       # kmg_options.strategy since round 6
        assert name not in blob, name


@pytest.mark.gpu
def test_product_library_ignores_every_tools_switch():
    """with all of them set, the product library still returns what the per-pixel scans return (and the oracle: the next test)"""
    env = dict(os.environ, **TOOLS_SWITCHES)
    env.pop("KMG_LIBRARY", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "24", "21"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "24 cases, 0 mismatching" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_default_call.py"), "12", "5"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "12 cases, 0 mismatching" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_random_default_calls_equal_the_oracle(seed):
    """tools/fuzz_default_call.py: kmg_reduce / kmg_palette at the reference's defaults (shrink to <= 256, init, Lloyd loop, output
    pass -- lib.rs:116-164) of random images, k and modes against oracle.reduce / oracle.palette, byte for byte"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_default_call.py"), "30", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "30 cases, 0 mismatching" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7])
def test_random_group_calls_equal_the_single_processor(seed):
    """tools/fuzz_group.py: palette / find / reduce through kmg_group_* (one rank with forced RCCL collectives, two to five ranks
    sharing the GPU through the loopback exchange; the reference's shrink and full resolution) against the single processor"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_group.py"), "24", str(seed)], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "24 cases, 0 mismatching" in r.stdout
