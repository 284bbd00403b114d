#!/bin/bash
# kernel times of the cube launches for a few grid sizes (run on the GPU box): bash tools/grid_sweep.sh
make -C ${GRAFT_REPO_ROOT:-$(pwd)}/kmeans-gpu_amd tools > /dev/null && export KMG_LIBRARY=${GRAFT_REPO_ROOT:-$(pwd)}/kmeans-gpu_amd/lib/libkmeans_hip_tools.so   # the grid switches exist in the tools build only
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_grid_sweep.txt
: > $OUT
for g in "1536 3584 2048" "3072 5376 4096" "4608 7168 8192" "6144 10752 2048"; do
  set -- $g
  echo "== stage $1 scan $2 pairs $3" >> $OUT
  KMG_CUBE_GRID=$1 KMG_SCAN_GRID=$2 KMG_PAIRS_GRID=$3 bash $ROOT/tools/cube_phases.sh r02_gs --no-overlap
  grep cube $ROOT/gpurun_out/r02_gs_phases.txt >> $OUT
done
