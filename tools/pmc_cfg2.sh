#!/bin/bash
# PMC passes over BASELINE config 2's loop (4096^2, k = 16, colour table): one rocprofv3 run per counter group.
# usage: bash tools/pmc_cfg2.sh <outdir under gpurun_out>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_cfg2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --only cfg2 --strategy table --steps 3 --no-extras > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for kn, d in acc.items():
        if not any(t in kn for t in ("k_labels", "k_cube", "k_assign", "k_update")): continue
        fo.write(kn + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:42s} mean/launch {sum(v)/len(v):16.1f}  launches {len(v)}\n")
PY
rm -rf $OUT/p[0-9]*
