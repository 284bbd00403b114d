#!/bin/bash
# BASELINE config 2 (4096^2, k = 16) on the GPU box: bash tools/run_cfg2_profile.sh <tag>
#   gpurun_out/<tag>/cfg2_bench.json        python bench.py --only cfg2 (both strategies, HIP-event kernel times)
#   gpurun_out/<tag>/cfg2_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same loop (table strategy)
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
cd $R && python3 bench.py --only cfg2 > gpurun_out/$TAG/cfg2_bench.json 2> gpurun_out/$TAG/cfg2_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG/ks
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/ks -- python3 $R/bench.py --only cfg2 --strategy table --no-extras > $R/gpurun_out/$TAG/ks.log 2>&1 || exit 1
cp $(find $R/gpurun_out/$TAG/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$TAG/cfg2_kernel_stats.csv
rm -rf $R/gpurun_out/$TAG/ks
cat $R/gpurun_out/$TAG/cfg2_bench.json
python3 - $R/gpurun_out/$TAG/cfg2_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'kmg::' in r['Name']:
        print(r['Name'][:70].ljust(72), r['Calls'], r['AverageNs'])
PY
