#!/bin/bash
# A/B of two builds of the library on ONE box: rocprofv3 kernel stats of bench.py's loop, alternating.
#   RUNS=3 bash tools/lib_ab.sh <tag> <lib A> <lib B> [<lib C> ...]   -> gpurun_out/<tag>_lib_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}; shift; LIBS="$@"; RUNS=${RUNS:-3}
OUT=$ROOT/gpurun_out/${TAG}_lib_ab.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $RUNS); do
  for lib in $LIBS; do
    export KMG_LIBRARY=$ROOT/$lib
    rm -rf $ROOT/gpurun_out/${TAG}_abprof
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_abprof -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/${TAG}_ab_bench.json 2> /dev/null
    echo "== $lib (run $rep)" >> $OUT
    python3 - $ROOT/gpurun_out/${TAG}_abprof $ROOT/gpurun_out/${TAG}_ab_bench.json >> $OUT <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_labels_pairs" in r["Name"] or "k_cube" in r["Name"]:
        print(f"  {r['Name'].split('(')[0].replace('void kmg::', ''):28s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:7.2f} us")
print("  ms_per_step", json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["ms_per_step"])
PY
  done
done
rm -rf $ROOT/gpurun_out/${TAG}_abprof
cat $OUT
