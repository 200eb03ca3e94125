#!/bin/bash
# Counters of the closed-form launch (tools/pmc_target.py): one rocprofv3 --pmc pass per counter group, summarised per kernel.
set -o pipefail
TAG=${1:-pmcperm}
OUT=${GRAFT_REPO_ROOT:?}/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
            "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -- python3 ${GRAFT_REPO_ROOT:?}/tools/pmc_target.py > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        if "perm" in k or "or_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-24s %.5g (n=%d)"%(c, sum(v)/len(v), len(v)))
PY
