#!/bin/bash
# PMC passes over the benchmark iteration (run on the GPU box): one rocprofv3 run per counter group.
# usage: bash tools/pmc_labels.sh <outdir under gpurun_out>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_TAG_STALL_sum TCC_REQ_sum" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-overlap > $OUT/p$i.log 2>&1
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
        if not any(t in kn for t in ("k_labels", "k_cube", "k_cell_candidates", "k_update")): continue
        fo.write(kn + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:42s} mean/launch {sum(v)/len(v):16.1f}  launches {len(v)}\n")
PY
