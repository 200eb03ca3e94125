#!/bin/bash
# Instruction-fetch side of the sweep kernels (the fused Schwinger launch is 59 KB of code; the instruction cache is 64 KB
# per pair of CUs): SQ / SQC fetch counters, one --pmc pass beside --kernel-trace -> gpurun_out/ifetch_<tag>.txt
#   bash tools/pmc_ifetch.sh TAG [bench.py args]
set -o pipefail
TAG=${1:-ifetch}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-extra-points}
ROOT=${GRAFT_REPO_ROOT:?}
OUT=$ROOT/gpurun_out/ifetch_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQC\?_[A-Z_]*\(IFETCH\|ICACHE\|INST_CACHE\)[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt
n=0
for P in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_SCA"; do
  n=$((n + 1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$n -- python3 $ROOT/bench.py $ARGS > $OUT/p$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/p$n.log; }
done
python3 - <<PY | tee $ROOT/gpurun_out/ifetch_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "perm_heat" in row["Kernel_Name"] or "gff_or_heat" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0].replace("void mlmcpi::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        print("   %-30s %.5g  (%d launches)" % (c, sum(d[c]) / len(d[c]), len(d[c])))
PY
find $OUT -name "*.csv" -delete
