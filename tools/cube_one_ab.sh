#!/bin/bash
# rocprofv3 kernel durations of the cube pass's variants, tools build, on the GPU box:  bash tools/cube_one_ab.sh <tag>
#   KMG_CUBE_ONE=0 (the four launches), KMG_ONE_BLOCK=512 / 1024 (k_cube_one with 8 / 16 waves per workgroup)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$ROOT/gpurun_out
make -j8 -C $ROOT/kmeans-gpu_amd tools > /dev/null && export KMG_LIBRARY=$ROOT/kmeans-gpu_amd/lib/libkmeans_hip_tools.so
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_cube_one_ab.txt
for v in "KMG_CUBE_ONE=0" "KMG_ONE_BLOCK=512" "KMG_ONE_BLOCK=1024"; do
  rm -rf $OUT/${TAG}_ab
  export KMG_CUBE_ONE=1 KMG_ONE_BLOCK=1024
  export $v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ab -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_ab.log 2>&1
  echo "== $v: $(grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG}_ab.log | tail -1)" >> $OUT/${TAG}_cube_one_ab.txt
  python3 - $(find $OUT/${TAG}_ab -name "*kernel_stats.csv" | head -1) >> $OUT/${TAG}_cube_one_ab.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("void kmg::k_cube", "void kmg::k_labels")):
        print("   %-60s calls %5s  avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
rm -rf $OUT/${TAG}_ab
cat $OUT/${TAG}_cube_one_ab.txt
