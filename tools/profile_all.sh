#!/bin/bash
# Round profile: per workload one `rocprofv3 --kernel-trace --stats` run of bench.py, one SQ counter pass, and (sweep
# workloads) the two TCC passes for HBM traffic.  Every --pmc pass is on its own beside --kernel-trace only; the program
# goes directly after `--`.  Output: gpurun_out/prof_<tag>/<workload>/{stats,sq,mix1,mix2,FETCH_SIZE,WRITE_SIZE}
# The eight workloads do not fit one gpurun call comfortably: run WORKLOADS="schwinger gff rotor_sweep" with tag T and the
# rest with tag Tb (two gpurun calls), then merge Tb's summary.json and *_kernel_stats.csv into gpurun_out/prof_T/ (json
# dict update; tools/make_traffic_json.py reads that directory).  Do NOT re-run profile_summarise.py afterwards: the raw
# traces are deleted at the end of this script and it would write an empty summary.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:?}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-schwinger gff rotor_hmc quartic_mlmc quartic_mlmc_hier rotor_sweep quartic_hmc ho_hmc}; do
  ARGS="--workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-extra-points"
  if [ $W = schwinger ]; then ARGS="$ARGS --probes"; fi   # + single launches of the HBM-bound kernels (bench.py hbm_bound_probes)
  mkdir -p $OUT/$W
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/stats -- python3 $ROOT/bench.py $ARGS > $OUT/$W/stats.log 2>&1 || { echo "$W: stats pass failed"; tail -3 $OUT/$W/stats.log; exit 1; }
  grep '^{' $OUT/$W/stats.log | tail -1 > $OUT/$W/bench_profiled.json
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$W/sq -- python3 $ROOT/bench.py $ARGS > $OUT/$W/sq.log 2>&1 || { echo "$W: SQ pass failed"; tail -3 $OUT/$W/sq.log; exit 1; }
  # the dynamic VALU instruction mix by class (two passes of 8 SQ counters): what the cost-weighted issue bound is made of
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $OUT/$W/mix1 -- python3 $ROOT/bench.py $ARGS > $OUT/$W/mix1.log 2>&1 || { echo "$W: mix1 pass failed"; tail -3 $OUT/$W/mix1.log; exit 1; }
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --kernel-trace --output-format csv -d $OUT/$W/mix2 -- python3 $ROOT/bench.py $ARGS > $OUT/$W/mix2.log 2>&1 || { echo "$W: mix2 pass failed"; tail -3 $OUT/$W/mix2.log; exit 1; }
  if [ $W = schwinger ] || [ $W = gff ] || [ $W = rotor_sweep ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$W/$C -- python3 $ROOT/bench.py $ARGS > $OUT/$W/$C.log 2>&1 || { echo "$W: $C pass failed"; tail -3 $OUT/$W/$C.log; exit 1; }
    done
  fi
  echo "$W profiled"
done
python3 $ROOT/tools/profile_summarise.py $OUT
# gpurun merges at most 64 MiB back: the raw traces and counter dumps are condensed in summary.json / *_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
