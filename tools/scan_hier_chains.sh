#!/bin/bash
# chains-per-level scan of the hierarchical MLMC workload (hmc_trajectory_kernel<1,8> on the coarsest level, M_lat = 2048:
# 4 waves per chain): where does the rate stop growing with the batch?   -> gpurun_out/scan_hier_<tag>.txt
TAG=${1:-hier}
mkdir -p gpurun_out
: > gpurun_out/scan_hier_$TAG.txt
for B in 512 1024 2048 4096; do
  timeout -k 10 500 python bench.py --workload quartic_mlmc_hier --chains $B --steps 6 --warmup 2 --no-cpu-baseline --epsilon 1.0 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('chains %5d  ms_per_step %9.3f  value %8.1f G site-steps/s  launch_ms %.3f' % ($B, d['ms_per_step'], d['value'] / 1e9, d['roofline']['launch_ms']))
" >> gpurun_out/scan_hier_$TAG.txt || echo "chains $B failed" >> gpurun_out/scan_hier_$TAG.txt
done
cat gpurun_out/scan_hier_$TAG.txt
