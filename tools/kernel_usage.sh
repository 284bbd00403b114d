#!/bin/bash
# registers / LDS / occupancy of the kernels of one translation unit (no GPU needed): bash tools/kernel_usage.sh kmg_cube [filter]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
F=${1:-kmg_cube}; PAT=${2:-k_}
mkdir -p $ROOT/kmeans-gpu_amd/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -o $ROOT/kmeans-gpu_amd/build/$F.s $ROOT/kmeans-gpu_amd/csrc/$F.hip 2> $ROOT/kmeans-gpu_amd/build/$F.usage.txt
python3 - $ROOT/kmeans-gpu_amd/build/$F.usage.txt "$PAT" <<'PY'
import re, sys, subprocess
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    key, val = m.groups()
    if key == "Function Name":
        cur = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur: rows[cur][key.split(" ")[0]] = val
for k, v in rows.items():
    if sys.argv[2] in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs'):>4s} SGPR {v.get('SGPRs'):>4s} scratch {v.get('ScratchSize'):>4s} occ {v.get('Occupancy'):>2s} lds {v.get('LDS')}")
PY
