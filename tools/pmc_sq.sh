#!/bin/bash
# SQ counters for the sweep kernels (one pass, <= 8 SQ counters), summarised per kernel.
set -o pipefail
TAG=${1:-sq}; shift
CTRS=${1:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-extra-points}
OUT=${GRAFT_REPO_ROOT:?}/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/sq -- python3 ${GRAFT_REPO_ROOT:?}/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        if "sweep" in k or "hmc" in k or "or_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in d.items(): print("   %-24s %.4g (n=%d)"%(c, sum(v)/len(v), len(v)))
PY
