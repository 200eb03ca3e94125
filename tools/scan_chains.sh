#!/bin/bash
# Step time against the number of chains per GPU, default workload.
for B in 1 2 4 8 16 32 64 128; do
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-points --chains $B 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('chains $B', '%.1f G/s' % (r['value']/1e9), '%.4f ms/step' % r['ms_per_step'], [(k['kernel'], round(k['launch_ms'], 4), k['launches_per_step']) for k in r['kernels']])" || exit 1
done
