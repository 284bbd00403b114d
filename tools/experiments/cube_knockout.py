#!/usr/bin/env python3
"""Knock-out timing of k_cube (run on the GPU box): python tools/cube_knockout.py > gpurun_out/knock.txt
Each line = one bench.py run with a KMG_CUBE_FLAGS knock-out (results of those runs are wrong by design)."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toolslib import use_tools_library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("baseline", 0), ("no pair entry", 0x100), ("no sums", 0x200), ("no colour scan", 0x400),
         ("all cells single (candidates only)", 0x800), ("no sub-cell stage", 0x1000), ("no label stores", 0x2000),
         ("no scan+pairs+sums", 0x700), ("no scan+pairs+sums+labels", 0x2700),
         ("scan without colour loads", 0x4000), ("colour loads without scan", 0x8000),
         ("no loads, no scan loop", 0xC000), ("no loads+pairs+sums+labels", 0x6300)]
extra = sys.argv[1:]
for name, fl in cases:
    env = dict(os.environ, KMG_CUBE_FLAGS=hex(fl))
    use_tools_library(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "10"] + extra,
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
        print(f"{name:40s} k_cube {d['kernels']['k_cube']['ms_per_launch']*1e3:8.1f} us   step {d['ms_per_step']*1e3:8.1f} us", flush=True)
    except Exception as e:
        print(name, "failed", e, out[-3:])
