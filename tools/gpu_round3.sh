set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout 400 -p no:cacheprovider -k "parity or host_layer or comm" > gpurun_out/pytest_gpu_r02e.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu_r02e.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r02e.json 2> gpurun_out/bench_r02e.err; cut -c1-200 gpurun_out/bench_r02e.json
bash tools/pmc_sq.sh r02e_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-points --thermalise 5
bash tools/pmc_sq.sh r02e_sq2 "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-points --thermalise 5
