#!/bin/bash
# Every bench workload once (short), JSON lines into gpurun_out/bench_<tag>_<workload>.json; then the N = 2 rehearsal of
# the rank-spawning path on one GPU (gloo backend, ranks share the device).
set -o pipefail
TAG=${1:-all}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err || { tail -5 gpurun_out/bench_${TAG}.err; exit 1; }
cut -c1-600 gpurun_out/bench_${TAG}.json
# the C++ path's own records: the driver's JSON lines as printed
for B in 32 1; do
  N=400; [ $B = 1 ] && N=2000
  timeout -k 10 300 host/driver --method throughput --action schwinger --Mt_lat 1024 --sampler heatbath --batch $B --n_samples $N --n_burnin 30 2> gpurun_out/driver_${TAG}_b$B.err | grep '^{' | tail -1 > gpurun_out/driver_${TAG}_b$B.json || { echo "driver b$B failed"; tail -5 gpurun_out/driver_${TAG}_b$B.err; exit 1; }
  cut -c1-300 gpurun_out/driver_${TAG}_b$B.json
done
for W in gff rotor_hmc quartic_hmc ho_hmc quartic_mlmc quartic_mlmc_hier rotor_sweep; do
  NOCPU="--no-cpu-baseline"; case $W in gff|rotor_hmc|quartic_mlmc) NOCPU="";; esac   # CPU leg for the BASELINE configs only
  timeout -k 10 400 python bench.py --workload $W --steps 10 --warmup 2 $NOCPU > gpurun_out/bench_${TAG}_$W.json 2> gpurun_out/bench_${TAG}_$W.err || { echo "$W failed"; tail -5 gpurun_out/bench_${TAG}_$W.err; exit 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/bench_${TAG}_$W.json"))
print("$W", "%.4g" % r["value"], r["unit"], "ms/step %.3f" % r["ms_per_step"], "roofline frac %.3f" % r["roofline"]["frac"], "valu_frac", r["roofline"].get("valu_frac"), "qoi", r.get("qoi_mean"))
PY
done
# the hierarchical sampler where its two-level steps DO accept (T_final = M_lat / 32, a = 1 / 32 on the finest level): the
# multilevel path with moving chains at bench scale, acceptance per level in the line
timeout -k 10 500 python bench.py --workload quartic_mlmc_hier --t-final 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_quartic_mlmc_hier_T1024.json 2> gpurun_out/bench_${TAG}_quartic_mlmc_hier_T1024.err || { echo "hier T1024 failed"; tail -5 gpurun_out/bench_${TAG}_quartic_mlmc_hier_T1024.err; exit 1; }
python - <<PY
import json
r = json.load(open("gpurun_out/bench_${TAG}_quartic_mlmc_hier_T1024.json"))
print("hier T1024", "%.4g" % r["value"], "ms/step %.3f" % r["ms_per_step"], "estimate", r["mlmc"]["estimate"], "+-", r["mlmc"]["error"], "frozen", r["mlmc"]["frozen_levels"], "z", r["mlmc"]["run_to_epsilon"].get("single_level_fine_hmc", {}).get("z"))
print("   acceptance", r["mlmc"]["hierarchical_acceptance_rank0"])
PY
MLMCPI_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 1 --chains 8 --no-extra-points > gpurun_out/bench_${TAG}_n2.json 2> gpurun_out/bench_${TAG}_n2.err; echo "n2 rehearsal exit $?"; cut -c1-300 gpurun_out/bench_${TAG}_n2.json
MLMCPI_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --workload quartic_mlmc --steps 3 --warmup 1 --chains 64 > gpurun_out/bench_${TAG}_mlmc_n2.json 2> gpurun_out/bench_${TAG}_mlmc_n2.err; echo "mlmc n2 rehearsal exit $?"; cut -c1-300 gpurun_out/bench_${TAG}_mlmc_n2.json
timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_n2_nogpu.json 2> gpurun_out/bench_${TAG}_n2_nogpu.err; echo "n2 without a second GPU: exit $? (must be non-zero)"
