#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MLMCPI_LIB_VARIANT=r05a timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_r05a.txt 2> gpurun_out/hash_r05a.err || { tail -5 gpurun_out/hash_r05a.err; exit 1; }
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_new.txt 2> gpurun_out/hash_new.err || { tail -5 gpurun_out/hash_new.err; exit 1; }
if diff gpurun_out/hash_r05a.txt gpurun_out/hash_new.txt > gpurun_out/hash_diff.txt; then echo "HASHES EQUAL ($(wc -l < gpurun_out/hash_new.txt) cases)"; else echo "HASHES DIFFER"; head -20 gpurun_out/hash_diff.txt; fi
timeout -k 10 400 python -m pytest tests/test_gpu_statistics.py -m gpu -q --timeout 300 -p no:cacheprovider -s -k "windowed or reproduces_the_bias" > gpurun_out/pytest_s5.log 2>&1; grep -E "device level 0|\[z\]|passed|failed|Error|assert" gpurun_out/pytest_s5.log | head -20
bash tools/ab.sh s5 "" r05a r04
