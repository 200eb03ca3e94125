#!/bin/bash
# Secondary workloads (BASELINE configs 2, 3, 5-finest) -- one JSON line each.
set -o pipefail
mkdir -p gpurun_out
for W in rotor_hmc gff quartic_hmc ho_hmc rotor_sweep quartic_mlmc; do
  timeout -k 10 400 python bench.py --workload $W --steps 5 --warmup 1 > gpurun_out/bench_$W.json 2> gpurun_out/bench_$W.err || { echo "$W failed"; tail -5 gpurun_out/bench_$W.err; exit 1; }
  python - <<PY
import json
r=json.load(open("gpurun_out/bench_$W.json"))
rf=r.get("roofline",{})
print("$W", "value %.2f G/s"%(r["value"]/1e9), "ms/step %.3f"%r["ms_per_step"], "roofline frac %.3f"%rf.get("frac",0), "launch_ms %.3f"%rf.get("launch_ms",0), "HB", r.get("heatbath",{}).get("launch_ms"), "cpu %.3g (%s cores)"%(r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"]), "x%.0f"%r.get("gpu_over_cpu",0), "qoi", r["qoi_mean"])
PY
done
