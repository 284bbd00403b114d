#!/bin/bash
# per-kernel times of BASELINE config 3 (tools/cfg3_probe.py) under the rocprofv3 kernel trace, run on the GPU box:
#   bash tools/init_phases.sh <tag>      -> gpurun_out/<tag>_init.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-init}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_initprof
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_initprof -- python3 $ROOT/tools/cfg3_probe.py > $ROOT/gpurun_out/${TAG}_init.txt 2>&1
python3 - $ROOT/gpurun_out/${TAG}_initprof >> $ROOT/gpurun_out/${TAG}_init.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{n:46s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
# gaps between consecutive init launches (start to start) of the last initialisation
python3 - $ROOT/gpurun_out/${TAG}_initprof >> $ROOT/gpurun_out/${TAG}_init.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_init_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-256:]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
dur = [e - s for s, e in zip(st, en)]
gap = [st[i + 1] - en[i] for i in range(len(rows) - 1)]
print("last init: launches", len(rows), "span %.2f ms" % ((en[-1] - st[0]) / 1e6), "kernel time %.2f ms" % (sum(dur) / 1e6), "gaps %.2f ms" % (sum(gap) / 1e6))
print("durations us, every 16th:", [round(d / 1e3, 1) for d in dur[::16]])
print("gaps us, every 16th:", [round(g / 1e3, 1) for g in gap[::16]])
PY
