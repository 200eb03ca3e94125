#!/bin/bash
# One GPU-box session of round 4: parity tests, bench, and whatever experiment scripts are named after the tag.
#   gpurun -- 'bash tools/session.sh TAG [step ...]'     steps: tests bench issue avail prof
set -o pipefail
TAG=${1:-s}; shift
STEPS=${@:-tests bench}
mkdir -p gpurun_out
for S in $STEPS; do
  case $S in
    tests)
      timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
      rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu_$TAG.log; tail -6 gpurun_out/pytest_gpu_$TAG.log
      if [ $rc -gt 1 ]; then exit $rc; fi ;;
    bench)
      timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
      cut -c1-400 gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
      if [ $rc -ne 0 ]; then exit $rc; fi ;;
    issue)
      timeout -k 10 120 tools/build/valu_issue_bench > gpurun_out/valu_issue_$TAG.txt 2>&1; rc=$?; tail -14 gpurun_out/valu_issue_$TAG.txt
      if [ $rc -ne 0 ]; then exit $rc; fi ;;
    avail)
      timeout -k 10 120 rocprofv3 --list-avail > gpurun_out/rocprof_avail.txt 2>&1; echo "list-avail exit $?"
      grep -c . gpurun_out/rocprof_avail.txt ;;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${GRAFT_REPO_ROOT:?}/gpurun_out/prof_$TAG -- python3 ${GRAFT_REPO_ROOT:?}/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-points > ${GRAFT_REPO_ROOT:?}/gpurun_out/prof_$TAG.log 2>&1 ); rc=$?
      echo "rocprof exit $rc"; find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-200
      if [ $rc -ne 0 ]; then exit $rc; fi ;;
    hierscan) bash tools/scan_hier_chains.sh $TAG || exit 1 ;;
    benchall) bash tools/bench_all.sh $TAG || exit 1 ;;
    stall) bash tools/pmc_stall.sh $TAG || exit 1 ;;
    ab) bash tools/ab.sh $TAG ${AB_VARIANTS:-"" pf r03} || exit 1 ;;
    *) echo "unknown step $S"; exit 64 ;;
  esac
done
