#!/bin/bash
# r05 session 9: variant A/B with bit-for-bit check.  usage: gpu_s9.sh TAG variant [variant ...]   ("" = product build)
set -o pipefail
mkdir -p gpurun_out
TAG=$1; shift
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_main.txt 2>&1 || { tail -5 gpurun_out/hash_main.txt; exit 1; }
for V in "$@"; do
  [ -z "$V" ] && continue
  MLMCPI_LIB_VARIANT=$V timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_$V.txt 2>&1 || { tail -5 gpurun_out/hash_$V.txt; exit 1; }
  if diff gpurun_out/hash_main.txt gpurun_out/hash_$V.txt > gpurun_out/hash_diff_$V.txt; then echo "HASH_EQUAL $V $(wc -l < gpurun_out/hash_main.txt) draws"; else echo "HASH_DIFFER $V"; head -5 gpurun_out/hash_diff_$V.txt; exit 1; fi
done
bash tools/ab.sh $TAG "$@"
