#!/bin/bash
# The three overrelaxation kernel families (MLMCPI_OR_KERNEL) and, for the LDS-resident one, its workgroup size
# (MLMCPI_OR_THREADS), at 12 overrelaxation sweeps per step.
for k in block patch lds; do for nt in "" 256 512 1024; do
  [ "$k" != lds ] && [ -n "$nt" ] && continue
  for f in 2 4 6; do
    [ "$k" = patch ] && [ $f -gt 4 ] && continue
    MLMCPI_OR_KERNEL=$k MLMCPI_OR_THREADS=$nt timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-points --fuse $f --n-overrelax 12 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$k NT=${nt:-default} fuse $f', '%.3f ms/step' % r['ms_per_step'], [(k['kernel'], round(k['launch_ms'], 4), k['launches_per_step']) for k in r['kernels'][:-2]])" || exit 1
  done; done; done
