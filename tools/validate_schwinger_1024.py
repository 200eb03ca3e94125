"""One-off validation at the headline size: Schwinger 1024^2, beta = 1 (or argv[4]), B chains, default sampler (10 OR + 1 HB
per draw): average plaquette against I1(beta)/I0(beta) and Q^2/(4 pi^2) against its mean over chains (sanity).
   python tools/validate_schwinger_1024.py [draws] [chains] [seed] [beta] [burn-in draws]"""
import math, sys, torch
sys.path.insert(0, ".")
from scipy import special
from mlmcpathintegral_amd import abi, ops
n_burn, n = 60, int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 99
beta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
if len(sys.argv) > 5:
    n_burn = int(sys.argv[5])
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=beta)
x = ops.lattice_initialise(act, B, seed)
s = torch.empty_like(x)
sweep = 0
plaq, chi = [], []
for k in range(n_burn + n):
    x, s = ops.lattice_sweep_draw_pingpong(act, x, s, 10, 1, seed, 0, sweep)
    sweep += 11
    if k >= n_burn:
        plaq.append(ops.qoi_avg_plaquette(x, 1024, 1024))
        chi.append(ops.qoi_2d_susceptibility(x, 1024, 1024))
P = torch.stack(plaq)          # [n, B]
cm = P.mean(dim=0)
m, e = float(cm.mean()), float(cm.std(unbiased=True)) / math.sqrt(B)
exact = special.i1e(beta) / special.i0e(beta)
C = torch.stack(chi).mean(dim=0)
# V chi_t = V/(4 pi^2) * <(sum_P theta_P)^2>/V ... compare with the large-volume value P * Phi(beta): per plaquette
# variance of the wrapped plaquette angle under exp(beta cos): sum over plaquettes (approximately independent)
import numpy as np
th = np.linspace(-np.pi, np.pi, 200001)
w = np.exp(beta * (np.cos(th) - 1.0)); w /= w.sum()
var_theta = float((w * th * th).sum())
print(f"beta {beta}: plaquette {m:.7f} +- {e:.7f} (exact {exact:.7f}, deviation {(m-exact)/e:+.2f} sigma)")
print(f"Q^2/(4 pi^2) {float(C.mean()):.1f} +- {float(C.std(unbiased=True))/math.sqrt(B):.1f} (independent-plaquette estimate {1024*1024*var_theta/(4*math.pi**2):.1f})")
