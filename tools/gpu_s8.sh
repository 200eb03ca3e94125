#!/bin/bash
# r05 session 8: mapped heat-bath cells -- parity of the Schwinger paths, same-box A/B against the HEAD build ("base"),
# and the co-residency analysis of the stamps build (tools/exp_stamps_overlap.py)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "schwinger or closed or heat or fused or lattice" > gpurun_out/pytest_s8.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_s8.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_main.txt 2>&1 || exit 1
MLMCPI_LIB_VARIANT=base timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_base.txt 2>&1 || exit 1
if diff gpurun_out/hash_main.txt gpurun_out/hash_base.txt > gpurun_out/hash_diff.txt; then echo "HASH_EQUAL $(wc -l < gpurun_out/hash_main.txt) draws"; else echo HASH_DIFFER; head -5 gpurun_out/hash_diff.txt; exit 1; fi
bash tools/ab.sh hbmap "" base || exit 1
MLMCPI_LIB_VARIANT=stamps timeout -k 10 300 python tools/exp_stamps_overlap.py 10 32 > gpurun_out/stamps_overlap.txt 2>&1; rc=$?
cat gpurun_out/stamps_overlap.txt
exit $rc
