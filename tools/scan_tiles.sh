#!/bin/bash
# Tuning scan of the generic kernels: tile shape x workgroup size x fused sweeps (results are identical by construction).
for cfg in 64x32x256:1 64x32x256:3 64x64x256:2 64x64x512:2 64x56x512:2 128x32x512:2 64x64x512:3 128x64x1024:2 96x64x1024:3 128x64x1024:1 64x64x512:1 128x32x256:1; do
  tile=${cfg%%:*}; f=${cfg##*:}
  MLMCPI_SWEEP_TILE=$tile timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-points --fuse $f 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('tile $tile fuse $f', '%.3f ms/step' % r['ms_per_step'], '%.1f G/s' % (r['value']/1e9), [(round(k['launch_ms'], 4), k['launches_per_step']) for k in r['kernels']])" || exit 1
done
