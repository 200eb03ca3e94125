#!/bin/bash
# Sample clocks and power of GPU 0 while the default bench runs (is the heat-bath kernel power limited?).
mkdir -p gpurun_out
python bench.py --steps ${1:-1500} --warmup 5 --no-cpu-baseline ${@:2} > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep 3
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|memory|HBM)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.3
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/power_bench.json')); print('bench', d['value']/1e9, d['ms_per_step'], d['roofline']['launch_ms'], d['heatbath']['launch_ms'])"
