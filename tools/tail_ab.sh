#!/bin/bash
# Is the cube pass's tail, which rides on the label pass's last workgroup, on that launch's critical path?  rocprofv3 kernel stats of
# bench.py's loop with the tail on the label pass (product) and as a launch of its own (tools build, KMG_TAIL_ON_LABELS=0), alternating.
#   bash tools/tail_ab.sh <tag>   -> gpurun_out/<tag>_tail_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-tail}
make -j8 -C $ROOT/kmeans-gpu_amd tools > /dev/null && export KMG_LIBRARY=$ROOT/kmeans-gpu_amd/lib/libkmeans_hip_tools.so
OUT=$ROOT/gpurun_out/${TAG}_tail_ab.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for mode in 1 0; do
    export KMG_TAIL_ON_LABELS=$mode
    rm -rf $ROOT/gpurun_out/${TAG}_tailprof
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_tailprof -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/${TAG}_tail_bench.json 2> /dev/null
    echo "== KMG_TAIL_ON_LABELS=$mode (run $rep)" >> $OUT
    python3 - $ROOT/gpurun_out/${TAG}_tailprof $ROOT/gpurun_out/${TAG}_tail_bench.json >> $OUT <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_labels_pairs" in r["Name"] or "k_cube" in r["Name"] or "k_update" in r["Name"]:
        print(f"  {r['Name'].split('(')[0].replace('void kmg::', ''):28s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:7.2f} us")
print("  ms_per_step", json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["ms_per_step"])
PY
  done
done
cat $OUT
