"""Experiment: run one kernel variant for a few seconds at a time and print wall-clock marks, to correlate with a
rocm-smi power / clock log taken by a shell loop beside it:
  (for i in $(seq 110); do echo "T $(date +%s.%N)"; rocm-smi -P -c -u | grep -E "Power|sclk"; sleep 0.3; done > gpurun_out/smi.log) &
  python tools/exp_power.py; wait"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
abi.load()
seed, size, B = 2481317, 1024, 32
act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)
x = ops.lattice_initialise(act, B, seed, 0)
s = torch.empty_like(x)
def burn(label, fn, secs=4.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    el = time.time() - t0
    print(f"{t0:.2f} .. {t0+el:.2f}  {label}: {1e3*el/n:.4f} ms/call", flush=True)
    time.sleep(1.0)
time.sleep(2.0)
for kernel, K in (("block", 6), ("block", 5), ("block", 1), ("patch", 4), ("block", 4)):
    abi.set_option("MLMCPI_OR_KERNEL", kernel)
    burn(f"OR {kernel} K={K}", lambda: ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K))
abi.set_option("MLMCPI_OR_KERNEL", "")
burn("heat bath sweep", lambda: ops.lattice_sweep_draw_pingpong(act, x, s, 0, 1, seed, 0, 0, 4))
y = torch.empty_like(x)
burn("copy (torch)", lambda: y.copy_(x))
