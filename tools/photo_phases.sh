#!/bin/bash
# per-kernel times of the Lloyd iteration on bench.py's tiled photograph (rocprofv3 kernel trace), run on the GPU box:
#   bash tools/photo_phases.sh <tag>      -> gpurun_out/<tag>_photo.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-photo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_photoprof
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_photoprof -- python3 $ROOT/tools/photo_phases.py > $ROOT/gpurun_out/${TAG}_photo.log 2>&1
python3 - $ROOT/gpurun_out/${TAG}_photoprof > $ROOT/gpurun_out/${TAG}_photo.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "kmg" in r["Name"] and int(r["Calls"]) > 10:
        n = r["Name"].split("(")[0].replace("void ", "")[:44]
        print(f"{n:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
tail -1 $ROOT/gpurun_out/${TAG}_photo.log >> $ROOT/gpurun_out/${TAG}_photo.txt
