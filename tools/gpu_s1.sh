#!/bin/bash
# session: variant equality (r04 vs working tree), closed-form tests, pins, A/B timing
set -o pipefail
mkdir -p gpurun_out
MLMCPI_LIB_VARIANT=r04 timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_r04.txt 2> gpurun_out/hash_r04.err || { tail -5 gpurun_out/hash_r04.err; exit 1; }
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_new.txt 2> gpurun_out/hash_new.err || { tail -5 gpurun_out/hash_new.err; exit 1; }
if diff gpurun_out/hash_r04.txt gpurun_out/hash_new.txt > gpurun_out/hash_diff.txt; then echo "HASHES EQUAL ($(wc -l < gpurun_out/hash_new.txt) cases)"; else echo "HASHES DIFFER"; head -20 gpurun_out/hash_diff.txt; fi
timeout -k 10 600 python -m pytest tests/test_reference_python_pins.py tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -p no:cacheprovider -k "closed_form or reference_held or reference_python or one_launch or 1024 or fused_qoi" > gpurun_out/pytest_s1.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_s1.log
if [ $rc -gt 1 ]; then exit $rc; fi
bash tools/ab.sh s1 "" r04
