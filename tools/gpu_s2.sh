#!/bin/bash
# session: early / late angles of the first half, phase stamps of the quadrant-plane build
set -o pipefail
mkdir -p gpurun_out
MLMCPI_LIB_VARIANT=stamps timeout -k 10 200 python tools/exp_stamps_perm.py > gpurun_out/stamps_r05a.txt 2> gpurun_out/stamps_r05a.err || { tail -5 gpurun_out/stamps_r05a.err; exit 1; }
cat gpurun_out/stamps_r05a.txt
bash tools/ab.sh s2 "" late r04
