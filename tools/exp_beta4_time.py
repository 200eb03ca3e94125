import os, sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
BETA = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=BETA)
x = ops.lattice_initialise(act, 32, 7, 0); w = torch.empty_like(x); s = 0
def draw():
    global x, w, s
    x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, 7, 0, s, 1); s += 11
for _ in range(5): draw()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): draw()
    torch.cuda.synchronize(); print(os.environ.get("MLMCPI_LIB_VARIANT", "main"), "beta=%g ms per draw %%.4f" % BETA % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
