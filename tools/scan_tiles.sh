#!/bin/bash
# Tuning scan: tile shape x workgroup size x fused sweeps (results are identical by construction).
for cfg in 64x32x256:1 64x32x256:3 64x64x256:2 64x64x512:2 64x56x512:2 128x32x512:2 64x64x512:3 128x64x1024:2 96x64x1024:3 128x64x1024:1 64x64x512:1 128x32x256:1; do
  tile=${cfg%%:*}; f=${cfg##*:}
  MLMCPI_SWEEP_TILE=$tile timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fuse $f | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('tile $tile fuse',r['config']['fuse'],'value %.1f G/s'%(r['value']/1e9),'OR/sweep %.3f ms'%(r['roofline']['launch_ms']*(-(-10//$f))/10),'OR %.1f G upd/s'%(r['roofline']['updates_per_s']/1e9),'HB %.2f ms'%r['heatbath']['launch_ms'])" || exit 1
done
