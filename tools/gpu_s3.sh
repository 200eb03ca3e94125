#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "closed_form or one_launch or fused_qoi or lattice_sweeps_match or 1024" > gpurun_out/pytest_s3.log 2>&1; rc=$?; tail -8 gpurun_out/pytest_s3.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_s3.json 2> gpurun_out/bench_s3.err; rc=$?; tail -3 gpurun_out/bench_s3.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/bench_s3.json'))
print('ms_per_step',r['ms_per_step'],'value',r['value']/1e9,'draws/s',r['draws_per_s'],'exec',r['executed_updates_per_s']/1e9)
for p in r['fast_path_cliff']: print(p['point'], round(p['ms_per_step'],4), p['over_headline'], p.get('over_committed_gff_512_line'))
print(r['random_order'])
print(r['cpu_baseline'])
print(r['roofline'])
PY
