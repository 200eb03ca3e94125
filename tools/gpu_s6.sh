#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V=${1:-par}
MLMCPI_LIB_VARIANT=$V timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_$V.txt 2> gpurun_out/hash_$V.err || { tail -5 gpurun_out/hash_$V.err; exit 1; }
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_new.txt 2> gpurun_out/hash_new.err || { tail -5 gpurun_out/hash_new.err; exit 1; }
if diff gpurun_out/hash_$V.txt gpurun_out/hash_new.txt > gpurun_out/hash_diff.txt; then echo "HASHES EQUAL ($(wc -l < gpurun_out/hash_new.txt) cases)"; else echo "HASHES DIFFER"; head -20 gpurun_out/hash_diff.txt; fi
bash tools/ab.sh s6$V "" $V
