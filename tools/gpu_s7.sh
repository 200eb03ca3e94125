#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MLMCPI_LIB_VARIANT=r05b timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_r05b.txt 2> gpurun_out/hash_r05b.err || { tail -5 gpurun_out/hash_r05b.err; exit 1; }
timeout -k 10 300 python tools/exp_variant_hash.py > gpurun_out/hash_new.txt 2> gpurun_out/hash_new.err || { tail -5 gpurun_out/hash_new.err; exit 1; }
if diff gpurun_out/hash_r05b.txt gpurun_out/hash_new.txt > gpurun_out/hash_diff.txt; then echo "HASHES EQUAL ($(wc -l < gpurun_out/hash_new.txt) cases)"; else echo "HASHES DIFFER"; head -20 gpurun_out/hash_diff.txt; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "gff or lattice_sweeps_match or closed_form or one_launch" > gpurun_out/pytest_s7.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_s7.log
if [ $rc -gt 1 ]; then exit $rc; fi
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
for M, B in ((1000, 256), (130, 8192), (512, 1024)):
    act = abi.lattice_action(abi.GFF, M, M, mass=10.0)
    for kern in ("", "lds"):
        abi.set_option("MLMCPI_OR_KERNEL", kern)
        x = ops.lattice_initialise(act, B, 3, 0); w = torch.empty_like(x); s = 0
        def step():
            global x, w, s
            x, w, _ = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, 3, 0, s, 3); s += 11
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
        abi.set_option("MLMCPI_OR_KERNEL", "")
        print(f"gff {M}x{M} B={B} kernel={kern or 'default'}: {el*1e3:.3f} ms = {M*M*11*B/el/1e9:.0f} G site-updates/s", flush=True)
PY
