#!/bin/bash
# Same-box A/B timing of library variants (abi.py: MLMCPI_LIB_VARIANT; make -C mlmcpathintegral_amd/csrc variant ...):
#   bash tools/ab.sh TAG "" pf r03      -> gpurun_out/ab_TAG.txt, three rounds over the variants ("" = the product build)
# Box-to-box spread of the headline is 7 %, launch-to-launch on one box < 1 %: only lines of one call are comparable.
TAG=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
for round in 1 2 3; do
  for V in "$@"; do
    MLMCPI_LIB_VARIANT=$V timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-points --allow-variant ${AB_ARGS:-} 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
ks = ' '.join('%s %.4f' % (k['kernel'].split('(')[0][-24:], k['launch_ms']) for k in d.get('kernels', []))
print('variant %-6s round $round  ms_per_step %.4f  value %.1f G/s  %s' % ('$V' or 'main', d['ms_per_step'], d['value'] / 1e9, ks))
" >> $OUT || { echo "variant '$V' failed" >> $OUT; }
  done
done
cat $OUT
