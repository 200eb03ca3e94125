"""Rate of the 1-D rotor overrelaxed heat-bath sampler on the device (M_lat = 65536, B chains)."""
import sys, time, torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
M, B = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
act = abi.path_action(abi.ROTOR, M, M / 8.0, 0.25)
x = ops.path_initialise(act, B, 3)
scr = torch.empty_like(x)
for n_or, n_hb in ((1, 1), (10, 1), (10, 0), (0, 1)):
    for k in range(3):
        ops.path_sweep_draw(act, x, scr, n_or, n_hb, 3, 0, 100 * k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for k in range(n):
        ops.path_sweep_draw(act, x, scr, n_or, n_hb, 3, 0, 1000 + 100 * k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{n_or} OR + {n_hb} HB: {dt*1e3:.3f} ms per draw, {M*B*(n_or+n_hb)/dt/1e9:.1f} G site-updates/s; chi_t {float(ops.qoi_susceptibility(x, M/8.0).mean()):.4f}")
