#!/bin/bash
# per-kernel times of the three cube launches + the label pass (rocprofv3 kernel trace), run on the GPU box:
#   bash tools/cube_phases.sh <tag> [extra bench.py args]      -> gpurun_out/<tag>_phases.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-phases}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 "$@" > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
python3 - $ROOT/gpurun_out/${TAG}_prof > $ROOT/gpurun_out/${TAG}_phases.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "kmg" in r["Name"] and int(r["Calls"]) > 2:
        n = r["Name"].split("(")[0].replace("void ", "")[:44]
        print(f"{n:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
python3 - $ROOT/gpurun_out/${TAG}_prof.log >> $ROOT/gpurun_out/${TAG}_phases.txt <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        print('ms_per_step', json.loads(line)['ms_per_step'])
PY
