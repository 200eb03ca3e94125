"""Thin torch-tensor front end of the C ABI, used by tests and bench.py.

PyTorch is plumbing only: it owns device memory and the stream; every computation goes through
libmlmcpi_hip.so.  All state tensors are float64 CUDA tensors of shape [B, n] in the reference's
SampleState layout (see include/mlmcpi_hip.h).
"""
import ctypes as C

import torch

from . import abi
from .abi import GFF, HARMONIC, QUARTIC, ROTOR, SCHWINGER  # noqa: F401


def _p(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f64(B, n, like):
    return torch.empty((B, n), dtype=torch.float64, device=like.device)


def _check_state(x, n):
    assert x.dtype == torch.float64 and x.dim() == 2 and x.shape[1] == n, (x.shape, n)


# ---- 1-D paths -----------------------------------------------------------------------------------
def path_evaluate(act, x):
    _check_state(x, act.M)
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    abi.call("mlmcpi_path_evaluate", C.byref(act), _p(x), x.shape[0], _p(out), _stream())
    return out


def path_force(act, x):
    _check_state(x, act.M)
    f = torch.empty_like(x)
    abi.call("mlmcpi_path_force", C.byref(act), _p(x), _p(f), x.shape[0], _stream())
    return f


def path_initialise(act, B, seed, chain0=0, device="cuda"):
    x = torch.empty((B, act.M), dtype=torch.float64, device=device)
    abi.call("mlmcpi_path_initialise", C.byref(act), _p(x), B, seed, chain0, _stream())
    return x


def qoi_xsquared(x):
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    abi.call("mlmcpi_qoi_xsquared", _p(x), x.shape[1], x.shape[0], _p(out), _stream())
    return out


def qoi_susceptibility(x, T_final):
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    abi.call("mlmcpi_qoi_susceptibility", _p(x), x.shape[1], float(T_final), x.shape[0], _p(out), _stream())
    return out


class PathHMC:
    """HMCSampler::draw on B device-resident chains (sampler/hmcsampler.cc:8-69)."""

    def __init__(self, act, B, nt, dt, n_rep=1, seed=1, chain0=0, device="cuda"):
        self.act, self.B, self.nt, self.dt, self.n_rep, self.seed, self.chain0 = act, B, nt, dt, n_rep, seed, chain0
        nbytes = C.c_size_t(0)
        abi.call("mlmcpi_path_hmc_workspace_bytes", C.byref(act), B, nt, C.byref(nbytes))
        self.work = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        self.accept = torch.zeros(B, dtype=torch.int32, device=device)
        self.energies = torch.zeros((B, 4), dtype=torch.float64, device=device)
        self.traj = 0
        self.n_total = 0
        self.n_accepted = torch.zeros(B, dtype=torch.int64, device=device)

    def draw(self, x, count_stats=True):
        _check_state(x, self.act.M)
        abi.call("mlmcpi_path_hmc_draw", C.byref(self.act), _p(x), self.B, self.nt, float(self.dt), self.n_rep,
                 self.seed, self.chain0, self.traj, _p(self.work), _p(self.accept), _p(self.energies), _stream())
        self.traj += self.n_rep
        if count_stats:
            self.n_total += 1
            self.n_accepted += self.accept
        return self.accept


def hmc_thermalise(hmc, x, n, dt_jitter=0.3):
    """Untimed burn-in of B chains from the reference's cold (x = 0) or random start.  A state with no potential
    energy in its stiff modes has a systematic leapfrog energy error ~ M dt^2 <omega^2> / 8 (measured: +2.65 at
    dt = 0.004, +44 at dt = 0.02 for the quartic action, M_lat = 32768, a = 1/32), so a cold chain rejects every
    trajectory at the production step size and stays cold for ever: ramp the step size up from dt / 20, then jitter
    it so that no mode sits at a leapfrog resonance.  The production dt is restored at the end."""
    dt0 = hmc.dt
    ramp = [0.05] * 6 + [0.1] * 6 + [0.2] * 6 + [0.5] * 6
    for k in range(n):
        hmc.dt = dt0 * (ramp[k] if k < len(ramp) else 1.0 + dt_jitter * (((k * 7) % 11) - 5) / 5.0)
        hmc.draw(x, count_stats=False)
    hmc.dt = dt0


def path_hmc_run(hmc, x, n_draws, qoi_kind):
    """n_draws x (HMCSampler::draw + QoI) on the device; returns (q [B, n_draws], accepted draws per chain)."""
    q = torch.empty((hmc.B, n_draws), dtype=torch.float64, device=x.device)
    cnt = torch.zeros(hmc.B, dtype=torch.int32, device=x.device)
    abi.call("mlmcpi_path_hmc_run", C.byref(hmc.act), _p(x), hmc.B, hmc.nt, float(hmc.dt), hmc.n_rep, n_draws, qoi_kind,
             hmc.seed, hmc.chain0, hmc.traj, _p(hmc.work), _p(q), _p(cnt), _stream())
    hmc.traj += n_draws * hmc.n_rep
    layout = C.c_int32(0)
    abi.call("mlmcpi_path_hmc_run_layout", C.byref(hmc.act), hmc.B, hmc.nt, C.byref(layout))
    if not layout.value:
        q = q.reshape(n_draws, hmc.B).t().contiguous()
    return q, cnt


class PathTwoLevelStep:
    """TwoLevelMetropolisStep (montecarlo/twolevelmetropolisstep.cc) on B device chains: Gaussian fill-in."""

    def __init__(self, fine, coarse, B, seed=1, chain0=0, device="cuda"):
        self.fine, self.coarse, self.B, self.seed, self.chain0 = fine, coarse, B, seed, chain0
        nbytes = C.c_size_t(0)
        abi.call("mlmcpi_path_twolevel_workspace_bytes", C.byref(fine), B, C.byref(nbytes))
        self.work = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        self.accept = torch.zeros(B, dtype=torch.int32, device=device)
        self.terms = torch.zeros((B, 3), dtype=torch.float64, device=device)
        self.theta = torch.zeros((B, fine.M), dtype=torch.float64, device=device)  # current fine state
        self.step = 0

    def set_state(self, x):
        self.theta.copy_(x)

    def draw(self, x_coarse, mask=None):
        """mask (int32 [B], optional): chains with mask == 0 are left alone and come back rejected
        (HierarchicalSampler::draw's `if (not accept) break`, sampler/hierarchicalsampler.cc:62-76)"""
        _check_state(x_coarse, self.coarse.M)
        abi.call("mlmcpi_path_twolevel_draw_masked", C.byref(self.fine), C.byref(self.coarse), _p(x_coarse), _p(self.theta),
                 self.B, self.seed, self.chain0, self.step, _p(self.work), _p(mask) if mask is not None else None,
                 _p(self.accept), _p(self.terms), _stream())
        self.step += 1
        return self.accept


class HOExactSampler:
    """HarmonicOscillatorAction::draw (the exact Gaussian sampler) on B device chains: x = L y on the fp64 matrix
    cores; the Cholesky factor is built once on the host (mlmcpi_ho_cholesky_factor)."""

    def __init__(self, act, B, seed=1, chain0=0, device="cuda"):
        import numpy as np
        self.act, self.B, self.seed, self.chain0 = act, B, seed, chain0
        lt = np.zeros((act.M, act.M))
        abi.call("mlmcpi_ho_cholesky_factor", C.byref(act), lt.ctypes.data_as(C.c_void_p))
        self.LT_host = lt
        self.LT = torch.from_numpy(lt).to(device)
        self.step = 0

    def draw(self, x=None):
        if x is None:
            x = torch.empty((self.B, self.act.M), dtype=torch.float64, device=self.LT.device)
        _check_state(x, self.act.M)
        abi.call("mlmcpi_path_exact_draw", C.byref(self.act), _p(self.LT), _p(x), self.B, self.seed, self.chain0, self.step,
                 _stream())
        self.step += 1
        return x


class GFFExactSampler:
    """GFFAction::draw (exact Gaussian sampler) on B device chains by spectral synthesis (batched 2-D FFT)."""

    def __init__(self, act, B, seed=1, chain0=0, device="cuda"):
        self.act, self.B, self.seed, self.chain0 = act, B, seed, chain0
        nbytes = C.c_size_t(0)
        abi.call("mlmcpi_lattice_exact_workspace_bytes", C.byref(act), B, C.byref(nbytes))
        self.work = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        self.step = 0

    def draw(self, phi=None):
        if phi is None:
            phi = torch.empty((self.B, self.act.Mt * self.act.Mx), dtype=torch.float64, device=self.work.device)
        _check_state(phi, self.act.Mt * self.act.Mx)
        abi.call("mlmcpi_lattice_exact_draw", C.byref(self.act), _p(phi), self.B, self.seed, self.chain0, self.step,
                 _p(self.work), _stream())
        self.step += 1
        return phi


class LatticeTwoLevelStep:
    """TwoLevelMetropolisStep on the Schwinger lattice, semi-coarsening (ExpCos fill-in), B device chains."""

    def __init__(self, fine, coarse, B, seed=1, chain0=0, device="cuda", cfa_kind=0):
        """cfa_kind 0: the conditioned fine action the reference's factory picks for the lattice; 1: the Gaussian variant
        (QuenchedSchwingerGaussianConditionedFineAction; lattices coarsened in both directions)"""
        self.fine, self.coarse, self.B, self.seed, self.chain0, self.cfa_kind = fine, coarse, B, seed, chain0, cfa_kind
        nbytes = C.c_size_t(0)
        abi.call("mlmcpi_lattice_twolevel_workspace_bytes", C.byref(fine), C.byref(coarse), B, C.byref(nbytes))
        self.work = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        self.accept = torch.zeros(B, dtype=torch.int32, device=device)
        self.terms = torch.zeros((B, 3), dtype=torch.float64, device=device)
        self.theta = torch.zeros((B, 2 * fine.Mt * fine.Mx), dtype=torch.float64, device=device)  # current fine state
        self.step = 0

    def set_state(self, x):
        self.theta.copy_(x)

    def draw(self, phi_coarse):
        _check_state(phi_coarse, 2 * self.coarse.Mt * self.coarse.Mx)
        abi.call("mlmcpi_lattice_twolevel_draw_cfa", C.byref(self.fine), C.byref(self.coarse), self.cfa_kind, _p(phi_coarse),
                 _p(self.theta), self.B, self.seed, self.chain0, self.step, _p(self.work), _p(self.accept), _p(self.terms),
                 _stream())
        self.step += 1
        return self.accept


def path_sweep_draw(act, x, scratch, n_overrelax, n_heatbath, seed, chain0, sweep0):
    _check_state(x, act.M)
    abi.call("mlmcpi_path_sweep_draw", C.byref(act), _p(x), _p(scratch), x.shape[0], n_overrelax, n_heatbath, seed,
             chain0, sweep0, _stream())


def path_sweep_draw_qoi(act, src, w0, w1, n_overrelax, n_heatbath, seed, chain0, sweep0, acc=None):
    """draw + topological susceptibility in one pass (mlmcpi_path_sweep_draw_qoi; rotor): reads `src` (w1 may be src),
    returns (result tensor, the other work tensor, chi [B]); acc [B, 5]: record_sample of chi in the same call."""
    _check_state(src, act.M)
    where = C.c_int32(0)
    q = torch.empty(src.shape[0], dtype=torch.float64, device=src.device)
    if acc is not None:
        assert acc.shape == (src.shape[0], 5) and acc.dtype == torch.float64 and acc.is_contiguous()
    abi.call("mlmcpi_path_sweep_draw_qoi", C.byref(act), _p(src), _p(w0), _p(w1), src.shape[0], n_overrelax, n_heatbath, seed,
             chain0, sweep0, _p(q), None if acc is None else _p(acc), C.byref(where), _stream())
    return (w0, w1, q) if where.value == 0 else (w1, w0, q)


def _site_args(x, sites):
    """(d_sites, n, single) for the site-update entry points: an int, or a uint32 / int32 device tensor of site indices"""
    if isinstance(sites, int):
        return None, 1, sites
    assert sites.is_cuda and sites.dtype in (torch.int32, torch.uint32) and sites.is_contiguous()
    return C.c_void_p(sites.data_ptr()), sites.numel(), 0


def path_site_updates(act, x, sites, heat, seed, chain0, step):
    """Action::heatbath_update / overrelaxation_update(state, l) (rotor): the sites in order, all chains of x in place"""
    _check_state(x, act.M)
    d, n, single = _site_args(x, sites)
    abi.call("mlmcpi_path_site_updates", C.byref(act), _p(x), x.shape[0], d, n, single, int(bool(heat)), seed, chain0, step, _stream())


def lattice_site_updates(act, x, sites, heat, seed, chain0, step):
    _check_state(x, lattice_size(act))
    d, n, single = _site_args(x, sites)
    abi.call("mlmcpi_lattice_site_updates", C.byref(act), _p(x), x.shape[0], d, n, single, int(bool(heat)), seed, chain0, step, _stream())


# ---- 2-D lattices ---------------------------------------------------------------------------------
def lattice_size(act):
    n = C.c_uint32(0)
    abi.call("mlmcpi_lattice_state_size", C.byref(act), C.byref(n))
    return n.value


def lattice_evaluate(act, phi):
    out = torch.empty(phi.shape[0], dtype=torch.float64, device=phi.device)
    abi.call("mlmcpi_lattice_evaluate", C.byref(act), _p(phi), phi.shape[0], _p(out), _stream())
    return out


def lattice_force(act, phi):
    f = torch.empty_like(phi)
    abi.call("mlmcpi_lattice_force", C.byref(act), _p(phi), _p(f), phi.shape[0], _stream())
    return f


def lattice_initialise(act, B, seed, chain0=0, device="cuda"):
    phi = torch.empty((B, lattice_size(act)), dtype=torch.float64, device=device)
    abi.call("mlmcpi_lattice_initialise", C.byref(act), _p(phi), B, seed, chain0, _stream())
    return phi


def lattice_sweep_draw(act, phi, scratch, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse=0):
    _check_state(phi, lattice_size(act))
    abi.call("mlmcpi_lattice_sweep_draw", C.byref(act), _p(phi), _p(scratch), phi.shape[0], n_overrelax, n_heatbath,
             seed, chain0, sweep0, fuse, _stream())


def lattice_sweep_draw_pingpong(act, a, b, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse=0):
    """Sweeps without the final copy; returns (state, scratch) -- the tensors swapped when needed."""
    _check_state(a, lattice_size(act))
    flag = C.c_int32(0)
    abi.call("mlmcpi_lattice_sweep_draw_pingpong", C.byref(act), _p(a), _p(b), a.shape[0], n_overrelax, n_heatbath,
             seed, chain0, sweep0, fuse, C.byref(flag), _stream())
    return (b, a) if flag.value else (a, b)


def lattice_sweep_draw_qoi(act, src, w0, w1, n_overrelax, n_heatbath, seed, chain0, sweep0, qoi_kind, fuse=0, acc=None):
    """draw + QoI in one pass (mlmcpi_lattice_sweep_draw_qoi): reads `src` (w1 may be src), returns (result tensor, the
    other work tensor, qoi [B]); qoi_kind 1 = average plaquette, 2 = Q^2 / (4 pi^2).  acc [B, 5]: record_sample of the QoI
    as well, in the same call (mlmcpi_lattice_sweep_draw_qoi_record)."""
    where = C.c_int32(0)
    q = torch.empty(src.shape[0], dtype=torch.float64, device=src.device)
    if acc is None:
        abi.call("mlmcpi_lattice_sweep_draw_qoi", C.byref(act), _p(src), _p(w0), _p(w1), src.shape[0], n_overrelax, n_heatbath, seed,
                 chain0, sweep0, fuse, qoi_kind, _p(q), C.byref(where), _stream())
    else:
        assert acc.shape == (src.shape[0], 5) and acc.dtype == torch.float64 and acc.is_contiguous()
        abi.call("mlmcpi_lattice_sweep_draw_qoi_record", C.byref(act), _p(src), _p(w0), _p(w1), src.shape[0], n_overrelax, n_heatbath,
                 seed, chain0, sweep0, fuse, qoi_kind, _p(q), _p(acc), C.byref(where), _stream())
    return (w0, w1, q) if where.value == 0 else (w1, w0, q)


def lattice_copy_from_fine(fine_act, rt, rx, fine):
    """Action::copy_from_fine on the device; returns the coarse-level state [B, n_coarse]."""
    n = lattice_size(fine_act) // (rt * rx)
    coarse = torch.empty((fine.shape[0], n), dtype=torch.float64, device=fine.device)
    abi.call("mlmcpi_lattice_copy_from_fine", C.byref(fine_act), rt, rx, _p(fine), _p(coarse), fine.shape[0], _stream())
    return coarse


def lattice_copy_from_coarse(fine_act, rt, rx, coarse, fine):
    """Action::copy_from_coarse on the device: updates the coarse-level entries of `fine` in place."""
    abi.call("mlmcpi_lattice_copy_from_coarse", C.byref(fine_act), rt, rx, _p(coarse), _p(fine), fine.shape[0], _stream())


def path_copy_from_fine(fine):
    coarse = torch.empty((fine.shape[0], fine.shape[1] // 2), dtype=torch.float64, device=fine.device)
    abi.call("mlmcpi_path_copy_from_fine", _p(fine), _p(coarse), coarse.shape[1], fine.shape[0], _stream())
    return coarse


def path_copy_from_coarse(coarse, fine):
    abi.call("mlmcpi_path_copy_from_coarse", _p(coarse), _p(fine), coarse.shape[1], fine.shape[0], _stream())


def qoi_phi_squared(phi):
    out = torch.empty(phi.shape[0], dtype=torch.float64, device=phi.device)
    abi.call("mlmcpi_qoi_phi_squared", _p(phi), phi.shape[1], phi.shape[0], _p(out), _stream())
    return out


def qoi_avg_plaquette(theta, Mt, Mx):
    out = torch.empty(theta.shape[0], dtype=torch.float64, device=theta.device)
    abi.call("mlmcpi_qoi_avg_plaquette", _p(theta), Mt, Mx, theta.shape[0], _p(out), _stream())
    return out


def qoi_2d_susceptibility(theta, Mt, Mx):
    out = torch.empty(theta.shape[0], dtype=torch.float64, device=theta.device)
    abi.call("mlmcpi_qoi_2d_susceptibility", _p(theta), Mt, Mx, theta.shape[0], _p(out), _stream())
    return out


class LatticeHMC:
    """HMCSampler::draw for a 2-D action (streaming leapfrog)."""

    def __init__(self, act, B, nt, dt, n_rep=1, seed=1, chain0=0, device="cuda"):
        self.act, self.B, self.nt, self.dt, self.n_rep, self.seed, self.chain0 = act, B, nt, dt, n_rep, seed, chain0
        nbytes = C.c_size_t(0)
        abi.call("mlmcpi_lattice_hmc_workspace_bytes", C.byref(act), B, C.byref(nbytes))
        self.work = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        self.accept = torch.zeros(B, dtype=torch.int32, device=device)
        self.energies = torch.zeros((B, 4), dtype=torch.float64, device=device)
        self.traj = 0

    def draw(self, phi):
        abi.call("mlmcpi_lattice_hmc_draw", C.byref(self.act), _p(phi), self.B, self.nt, float(self.dt), self.n_rep,
                 self.seed, self.chain0, self.traj, _p(self.work), _p(self.accept), _p(self.energies), _stream())
        self.traj += self.n_rep
        return self.accept


def stats_accumulate(acc, q):
    abi.call("mlmcpi_stats_accumulate", _p(acc), _p(q), q.shape[0], _stream())


def stats_window_state(B, window, device="cuda"):
    """zeroed state of mlmcpi_stats_window_record for B chains"""
    return torch.zeros((B, 2 * window + 3), dtype=torch.float64, device=device)


def stats_window_record(state, q):
    """Statistics::record_sample with the autocorrelation window, one chain per row of `state`"""
    window = (state.shape[1] - 3) // 2
    abi.call("mlmcpi_stats_window_record", _p(state), _p(q), q.shape[0], window, _stream())


def stats_window_tau_int(state, pooled=True):
    """Statistics::tau_int (common/statistics.cc:38-61) from the windowed state: per chain, or -- pooled -- with the
    autocovariances averaged over the chains of the batch first (one number for the batch)."""
    W = (state.shape[1] - 3) // 2
    n = state[:, 0:1]
    a1 = state[:, 1:2]
    cov = state[:, 2:2 + W] - a1 * a1                                  # [B, W]
    k = torch.arange(W, dtype=torch.float64, device=state.device)[None, :]
    wgt = 1.0 - k / torch.clamp(n, min=1.0)   # (the reference sums ALL k < window, also those the series has not reached: statistics.cc:86-88)
    if pooled:   # a 0-dim device tensor: no synchronisation here (the multilevel driver reads it one sample later)
        c = (cov * wgt).mean(dim=0)
        tau = 1.0 + 2.0 * c[1:].sum() / c[0]
        return torch.clamp(torch.nan_to_num(tau, nan=1.0, posinf=1.0, neginf=1.0), min=1.0)
    tau = 1.0 + 2.0 * (cov * wgt)[:, 1:].sum(dim=1) / cov[:, 0]
    return torch.clamp(torch.nan_to_num(tau, nan=1.0), min=1.0)


# ---- test hooks -------------------------------------------------------------------------------------
def test_random(seed, chain, step, purpose, sub, n, device="cuda"):
    out = torch.empty((n, 4), dtype=torch.float64, device=device)
    abi.call("mlmcpi_test_random", seed, chain, step, purpose, sub, n, _p(out), _stream())
    return out


def test_expcos(seed, chain, step, beta, xp, xm):
    out = torch.empty_like(xp)
    abi.call("mlmcpi_test_expcos", seed, chain, step, float(beta), _p(xp), _p(xm), xp.numel(), _p(out), _stream())
    return out


def test_vs_draw(seed, chain, step, scale, xp, xm):
    out = torch.empty_like(xp)
    abi.call("mlmcpi_test_vs_draw", seed, chain, step, float(scale), _p(xp), _p(xm), xp.numel(), _p(out), _stream())
    return out


def test_expsin2(seed, chain, step, sigma):
    out = torch.empty_like(sigma)
    abi.call("mlmcpi_test_expsin2", seed, chain, step, _p(sigma), sigma.numel(), _p(out), _stream())
    return out
