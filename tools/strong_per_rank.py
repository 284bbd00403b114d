#!/usr/bin/env python3
"""Strong scaling of ONE 8192x8192 image over N row bands, as far as one GPU can show it: the per-rank step
time for rows = 8192 / N (what every rank of an N-GPU `bench.py --scaling strong` run executes between two
all-reduces).  Writes gpurun_out/<tag>_strong_per_rank.json; the RCCL all-reduce of k x 4 int64 (8 KiB at k = 256,
latency bound) comes on top.    python tools/strong_per_rank.py <tag>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "strong"
out = {"workload": "synthetic uniform 8192x8192, k=256, one Lloyd iteration with the label map", "per_rank": {}}
for strategy in ("table", "scan"):
    for n in (1, 2, 4, 8):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "10",
               "--rows", str(8192 // n), "--strategy", strategy]
        line = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(line)
        out["per_rank"][f"{strategy}_N{n}"] = {"rows": 8192 // n, "ms_per_step": d["ms_per_step"],
                                                "kernels_ms": {k: v["ms_per_launch"] for k, v in d["kernels"].items()}}
        print(strategy, n, d["ms_per_step"], flush=True)
t1 = out["per_rank"]["table_N1"]["ms_per_step"]
out["speedup_bound_table"] = {f"N{n}": t1 / out["per_rank"][f"table_N{n}"]["ms_per_step"] for n in (1, 2, 4, 8)}
out["speedup_bound_best_strategy"] = {f"N{n}": t1 / min(out["per_rank"][f"table_N{n}"]["ms_per_step"], out["per_rank"][f"scan_N{n}"]["ms_per_step"]) for n in (1, 2, 4, 8)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_strong_per_rank.json"), "w"), indent=1)
