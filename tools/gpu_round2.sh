set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu_r02d.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_gpu_r02d.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 ./host/driver --method throughput --action schwinger --Mt_lat 1024 --sampler heatbath --batch 32 --n_samples 20 --n_burnin 30 > gpurun_out/driver_r02d.log 2>&1; tail -2 gpurun_out/driver_r02d.log
timeout -k 10 300 ./host/driver --method throughput --action schwinger --Mt_lat 1024 --sampler heatbath --batch 1 --n_samples 50 --n_burnin 30 > gpurun_out/driver_r02d_b1.log 2>&1; tail -1 gpurun_out/driver_r02d_b1.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02d.json 2> gpurun_out/bench_r02d.err; cut -c1-400 gpurun_out/bench_r02d.json
