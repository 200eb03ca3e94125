#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace profile.  Later steps only run when
# the earlier ones ended normally (exit 0/1), never after a timeout or kill.
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-run}
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu_$TAG.log
tail -5 gpurun_out/pytest_gpu_$TAG.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; rc=$?; tail -2 gpurun_out/smoke_$TAG.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d ${GRAFT_REPO_ROOT:?}/gpurun_out/prof_$TAG -- python3 ${GRAFT_REPO_ROOT:?}/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-points > ${GRAFT_REPO_ROOT:?}/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
find ${GRAFT_REPO_ROOT:?}/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -2 | xargs -r head -12
