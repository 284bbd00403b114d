#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py command line:  bash tools/kstats.sh <tag> <bench.py arguments ...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG/ks
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/ks -- python3 $R/bench.py "$@" > $R/gpurun_out/$TAG/ks.log 2>&1 || exit 1
cp $(find $R/gpurun_out/$TAG/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$TAG/kernel_stats.csv
rm -rf $R/gpurun_out/$TAG/ks
python3 - $R/gpurun_out/$TAG/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'kmg::' in r['Name'] and int(r['Calls']) > 4:
        print(r['Name'].split('(')[0][:60].ljust(62), r['Calls'], round(float(r['AverageNs'])/1e3, 1), round(float(r['MinNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1))
PY
