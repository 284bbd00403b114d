#!/bin/bash
# launches of a default kmg_palette call (tools/default_palette_trace.py) under the rocprofv3 kernel trace, on the GPU box:
#   bash tools/default_palette_trace.sh <tag> [k]     -> gpurun_out/<tag>/palette_trace.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ptrace}; K=${2:-256}
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/$TAG/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG/prof -- python3 $ROOT/tools/default_palette_trace.py $K > $ROOT/gpurun_out/$TAG/palette_trace.txt 2>&1
python3 - $ROOT/gpurun_out/$TAG/prof >> $ROOT/gpurun_out/$TAG/palette_trace.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kmg::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the launches after the last gap of more than 200 us
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200000: cut = i
last = rows[cut:]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3
print(f"last call: {len(last)} launches, first start to last end {span:.0f} us, kernels busy {busy:.0f} us")
by = collections.OrderedDict()
for r in last:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]
    d = by.setdefault(n, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in by.items(): print(f"  {n:52s} x{c:4d}  {t:8.1f} us  ({t / c:5.1f} each)")
PY
rm -rf $ROOT/gpurun_out/$TAG/prof
