#!/bin/bash
# rocprofv3 evidence for the OUTPUT passes (tools/apply_probe.py), run on the GPU box:  bash tools/profile_apply.sh <tag>
#   gpurun_out/<tag>_apply_kernel_stats.csv     --kernel-trace --stats
#   gpurun_out/<tag>_apply_pmc.txt              FETCH_SIZE / WRITE_SIZE and TA / TCP / SQ unit counters per kernel (one pass per group)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_aks $OUT/${TAG}_apmc; mkdir -p $OUT/${TAG}_apmc
python3 $ROOT/tools/apply_probe.py > $OUT/${TAG}_apply_host.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_aks -- python3 $ROOT/tools/apply_probe.py 4 > $OUT/${TAG}_aks.log 2>&1
cp $(find $OUT/${TAG}_aks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_apply_kernel_stats.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/${TAG}_apmc/p$i -- python3 $ROOT/tools/apply_probe.py 2 > $OUT/${TAG}_apmc/p$i.log 2>&1
done
python3 - "$OUT/${TAG}_apmc" > $OUT/${TAG}_apply_pmc.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in sorted(acc.items()):
    if not any(t in kn for t in ("k_dither", "k_offset", "k_lab_", "k_meld", "k_labels", "k_cube", "k_apply")): continue
    print(kn)
    for c, v in sorted(d.items()):
        print(f"  {c:36s} mean/launch {sum(v)/len(v):16.1f}  launches {len(v)}")
PY
