#!/bin/bash
# One chain per GPU (BASELINE configs[3] read literally): step time under the tuning knobs.
for env in "" "MLMCPI_OR_KERNEL=patch" "MLMCPI_SWEEP_TILE=32x32x256" "MLMCPI_SWEEP_TILE=32x16x256" "MLMCPI_SWEEP_TILE=64x16x256"; do
  for fuse in 0 2; do
    env $env python bench.py --chains 1 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-points --fuse $fuse 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$env fuse=$fuse', round(r['ms_per_step'],4), 'ms/step', round(r['value']/1e9,1), 'G/s', [(k['kernel'][:30], round(k['launch_ms'],4), k['launches_per_step']) for k in r['kernels']])
"
  done
done
