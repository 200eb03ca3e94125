#!/bin/bash
# Step time against the most overrelaxation sweeps per launch (--fuse; 0 = library default), default workload.
for f in 0 1 2 3 4 5 6; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-points --fuse $f 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('fuse $f', 'launches', r['config']['overrelaxation_launches'], '%.3f ms/step' % r['ms_per_step'], '%.1f G/s' % (r['value']/1e9), [(k['kernel'], round(k['launch_ms'], 4), k['launches_per_step']) for k in r['kernels'][:-1]])"; done
