#!/bin/bash
# timeline of BASELINE config 3's initialisation (tools/cfg3_init_time.py) under the rocprofv3 kernel trace, run on the GPU box:
#   bash tools/init_timeline.sh <tag>      -> gpurun_out/<tag>_inittl.txt   (the last call: every kernel, its duration, the gap before it)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-init}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_inittlprof
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_inittlprof -- python3 $ROOT/tools/cfg3_init_time.py > $ROOT/gpurun_out/${TAG}_inittl.txt 2>&1
python3 - $ROOT/gpurun_out/${TAG}_inittlprof >> $ROOT/gpurun_out/${TAG}_inittl.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kmg::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = from the last k_init_first on
last = max(i for i, r in enumerate(rows) if "k_init_first" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
multi = []
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void kmg::", "")[:28]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    if "k_init_cells_multi" in nm: multi.append(((e - s) / 1e3, gap))
    else: print(f"{nm:30s} at {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap before {gap:6.1f}")
    prev = e
print("k_init_cells_multi launches", len(multi), "kernel time %.3f ms" % (sum(d for d, g in multi) / 1e3), "gaps %.3f ms" % (sum(g for d, g in multi) / 1e3))
print("durations us:", [round(d, 1) for d, g in multi])
print("gaps us:", [round(g, 1) for d, g in multi])
print("span of the call's kernels %.3f ms" % ((prev - t0) / 1e6))
PY
