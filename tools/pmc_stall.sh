#!/bin/bash
# Where the idle issue slots of the sweep kernels go: three SQ counter passes (each --pmc on its own beside --kernel-trace,
# the program directly after `--`), per-launch averages per kernel -> gpurun_out/stall_<tag>.txt
#   bash tools/pmc_stall.sh TAG [bench.py args]
set -o pipefail
TAG=${1:-stall}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-extra-points}
ROOT=${GRAFT_REPO_ROOT:?}
OUT=$ROOT/gpurun_out/stall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
P3="SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC"
n=0
for P in "$P1" "$P2" "$P3"; do
  n=$((n + 1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$n -- python3 $ROOT/bench.py $ARGS > $OUT/p$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/p$n.log; exit 1; }
done
python3 - <<PY | tee $ROOT/gpurun_out/stall_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void mlmcpi::", "")
        if "mlmcpi" in row["Kernel_Name"]:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mlmcpi" in row["Kernel_Name"]:
            dur[row["Kernel_Name"].split("(")[0].replace("void mlmcpi::", "")].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    n = len(next(iter(d.values())))
    if not dur.get(k) or sum(dur[k]) < 1e5:
        continue
    print("%s   launches %d   avg %.1f us" % (k, n, sum(dur[k]) / len(dur[k]) / 1e3))
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        print("   %-26s %.5g" % (c, m[c]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in m:
                print("   %-26s / SQ_WAVE_CYCLES = %.3f" % (c, m[c] / wc))
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        print("   active lanes per VALU instruction (THREAD_CYCLES / ACTIVE_INST / 64... raw ratio) = %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]))
    if "SQ_INST_LEVEL_LDS" in m and "SQ_INSTS_LDS" in m:
        print("   LDS latency (INST_LEVEL_LDS / INSTS_LDS) = %.1f cycles" % (m["SQ_INST_LEVEL_LDS"] / m["SQ_INSTS_LDS"]))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print("   LDS bank conflict cycles / active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"])))
PY
find $OUT -name "*.csv" -delete
