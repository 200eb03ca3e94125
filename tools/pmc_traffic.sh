#!/bin/bash
# HBM traffic of the sweep kernels from PMC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
# WRITE_SIZE in separate passes (TCC slots), no tracing domains besides the kernel trace.
set -o pipefail
TAG=${1:-pmc}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-extra-points}
OUT=${GRAFT_REPO_ROOT:?}/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 ${GRAFT_REPO_ROOT:?}/bench.py $ARGS > $OUT/$C.log 2>&1 || { echo "pmc pass $C failed"; tail -5 $OUT/$C.log; exit 1; }
done
python3 ${GRAFT_REPO_ROOT:?}/tools/pmc_summarise.py $OUT
