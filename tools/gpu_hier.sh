#!/bin/bash
# The hierarchical MLMC pieces on the GPU: two-level / hierarchy tests, then the config-5 bench lines (reference composition
# and direct samplers).
set -o pipefail
TAG=${1:-hier}
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider -k "twolevel or hierarch or config5 or mlmc" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?"; grep -E "hierarchical 3-level|passed|failed" gpurun_out/pytest_$TAG.log
timeout -k 10 500 python bench.py --workload quartic_mlmc_hier --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_hier.json 2> gpurun_out/bench_${TAG}_hier.err; echo "bench exit $?"; tail -3 gpurun_out/bench_${TAG}_hier.err
python - <<PY
import json
r = json.load(open("gpurun_out/bench_${TAG}_hier.json"))
print("hier", "%.4g" % r["value"], "ms/step %.3f" % r["ms_per_step"], r["config"]["sub_sampling_rank0"], {k: r["mlmc"][k] for k in ("estimate", "error", "level_means", "level_tau_int", "hierarchical_acceptance_rank0", "acceptance_rank0", "run_to_epsilon")})
PY
timeout -k 10 500 python bench.py --workload quartic_mlmc_hier --t-final 1024 --dt 0.05 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_hier_T1024.json 2> gpurun_out/bench_${TAG}_hier_T1024.err; echo "bench exit $?"; tail -2 gpurun_out/bench_${TAG}_hier_T1024.err
python - <<PY
import json
r = json.load(open("gpurun_out/bench_${TAG}_hier_T1024.json"))
print("hier T=1024", "%.4g" % r["value"], "ms/step %.3f" % r["ms_per_step"], r["config"]["sub_sampling_rank0"], {k: r["mlmc"][k] for k in ("estimate", "error", "level_means", "level_tau_int", "hierarchical_acceptance_rank0", "acceptance_rank0", "run_to_epsilon")})
PY
