#!/bin/bash
# Everything the judged numbers come from, in one go on the GPU box:  bash tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json             python bench.py (N=1, defaults, with cpu_baseline and extras)
#   gpurun_out/<tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of bench.py --no-cpu-baseline --no-extras
#   gpurun_out/<tag>_pmc_fetch_write.csv    two separate passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE (rows of our kernels)
#   gpurun_out/<tag>_pmc_units.txt          TA / TCP / TCC / LDS / SQ counter groups, one pass each (tools/pmc_labels.sh)
#   gpurun_out/<tag>_photo_phases.txt       rocprofv3 --kernel-trace --stats of tools/photo_phases.py (the tiled photograph)
#   gpurun_out/<tag>_cfg2_bench.json, _cfg2_kernel_stats.csv, _cfg2_pmc.txt   BASELINE config 2 (4096^2, k = 16): bench.py --only cfg2,
#                                           rocprofv3 --kernel-trace --stats and the counter groups of the same loop (tools/pmc_cfg2.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$ROOT/gpurun_out
cd $ROOT && python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ks $OUT/${TAG}_pf $OUT/${TAG}_pw
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pf -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pw -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_pw.log 2>&1
cp $(find $OUT/${TAG}_ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 - $OUT $TAG <<'PY'
import csv, glob, sys
out, tag = sys.argv[1:3]
rows, hdr = [], None
for d in ("pf", "pw"):
    for f in glob.glob(f"{out}/{tag}_{d}/**/*counter_collection.csv", recursive=True):
        r = list(csv.reader(open(f))); hdr = r[0]
        ni = hdr.index("Kernel_Name")
        rows += [x for x in r[1:] if "kmg::" in x[ni]]
with open(f"{out}/{tag}_pmc_fetch_write.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(hdr); w.writerows(rows)
PY
cd $ROOT && bash tools/pmc_labels.sh ${TAG}_pmc > $OUT/${TAG}_pmc.log 2>&1
cp $OUT/${TAG}_pmc/summary.txt $OUT/${TAG}_pmc_units.txt
cd $ROOT && bash tools/run_cfg2_profile.sh ${TAG}_cfg2run > $OUT/${TAG}_cfg2.log 2>&1
cp $OUT/${TAG}_cfg2run/cfg2_bench.json $OUT/${TAG}_cfg2_bench.json
cp $OUT/${TAG}_cfg2run/cfg2_kernel_stats.csv $OUT/${TAG}_cfg2_kernel_stats.csv
bash tools/pmc_cfg2.sh ${TAG}_cfg2pmc >> $OUT/${TAG}_cfg2.log 2>&1
cp $OUT/${TAG}_cfg2pmc/summary.txt $OUT/${TAG}_cfg2_pmc.txt
# the iteration on bench.py's tiled photograph, per kernel (the pass without the dominance phase, long-list cells split)
cd $ROOT && bash tools/photo_phases.sh ${TAG} > /dev/null 2>&1
cp $OUT/${TAG}_photo.txt $OUT/${TAG}_photo_phases.txt
