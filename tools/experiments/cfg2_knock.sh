#!/bin/bash
# knock-out budget of the one-launch cube pass (k_cube_small) at BASELINE config 2 (4096^2, k = 16), on the GPU box; tools build,
# results of the knocked-out runs are wrong by design:  bash tools/cfg2_knock.sh   (flags: 0x200 no sums, 0x400 no colour scan,
# 0x4000 no entries phase, 0x8000 no dominance test)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
make -j8 -C $ROOT/kmeans-gpu_amd tools > /dev/null && export KMG_LIBRARY=$ROOT/kmeans-gpu_amd/lib/libkmeans_hip_tools.so
for f in 0 0x200 0x400 0x4000 0x8000 0x4600; do
  KMG_CUBE_FLAGS=$f python3 $ROOT/bench.py --only cfg2 --strategy table --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', d.get('cfg2_table_ms_per_step'), d.get('cfg2_table_kernels_ms'))"
done
