#!/bin/bash
# round 5: cost / gain probe of second-plane extension entries in the label pass (KMG_LABEL_KNOCK=9, tools build; results wrong)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
make -j8 -C $ROOT/kmeans-gpu_amd tools > /dev/null && export KMG_LIBRARY=$ROOT/kmeans-gpu_amd/lib/libkmeans_hip_tools.so
for kn in ${KNOCKS:-0 9 0 9}; do
  KMG_LABEL_KNOCK=$kn python3 $ROOT/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knock $kn: k_labels %.1f us  step %.1f us' % (d['kernels']['k_labels']['ms_per_launch']*1e3, d['ms_per_step']*1e3))"
done
