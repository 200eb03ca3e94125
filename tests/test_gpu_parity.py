"""GPU: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (north_star: bit-exact for integer/index work, stated fp tolerance otherwise):
  * Philox words and uniforms: bit-exact;
  * deterministic fp64 functions (evaluate, force, QoI, single updates, overrelaxation sweeps and
    HMC trajectories with the same counter-based random numbers): |diff| <= 1e-12 * scale, the
    slack covering libm-vs-ocml transcendentals, FMA contraction and summation order;
  * heat-bath sweeps: the accepted proposal is sigma * n with sigma = pi sqrt(2/tau) (ExpCos) or
    pi / sqrt(2 sigma) (ExpSin2); where the conditional distribution is nearly flat (tau -> 0) a
    rounding difference d in the staples or in the Box-Muller normal is amplified by
    ~ |x| tan(dx/2) / 2, so one sweep from identical inputs is compared at 2e-10 and two
    consecutive heat-bath sweeps at 1e-8 (single draws with given staples: 4e-12, see
    test_expcos_draws_match_oracle);
  * expectation values: within 4 combined standard errors of the reference chain / closed form
    (tests/test_gpu_statistics.py).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 0x1234567812345678
TOL = 1e-12
# angular tolerance (before the x4 of assert_angles_close) by number of heat-bath sweeps in the case
HB_TOL = {0: 1e-12, 1: 5e-11, 2: 2.5e-9}


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).cuda()


def seq(n):
    return np.sin(np.arange(n) + 1.0)


def assert_close(got, want, tol=TOL, scale=None, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    s = scale if scale is not None else max(1.0, float(np.max(np.abs(want))) if want.size else 1.0)
    err = float(np.max(np.abs(got - want))) if want.size else 0.0
    assert err <= tol * s, f"{what}: max |diff| = {err:.3e} > {tol:.1e} * {s:.3e}"


def assert_angles_close(got, want, tol=TOL, what=""):
    """angles in [-pi, pi): compare modulo 2 pi (a value within rounding of -pi may wrap)"""
    d = np.asarray(got) - np.asarray(want)
    d = d - 2 * np.pi * np.round(d / (2 * np.pi))
    err = float(np.max(np.abs(d)))
    assert err <= tol * 4, f"{what}: max angular diff = {err:.3e}"


# ---- RNG ---------------------------------------------------------------------------------------------
def test_philox_known_answers_on_device(gpu_ops):
    import ctypes as C
    from mlmcpathintegral_amd import abi
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "philox_kat.json")))
    for v in kat["vectors"]:
        ctr = np.array([int(x, 16) for x in v["ctr"]], dtype=np.uint32)
        key = np.array([int(x, 16) for x in v["key"]], dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        abi.call("mlmcpi_test_philox", ctr.ctypes.data_as(C.c_void_p), key.ctypes.data_as(C.c_void_p),
                 out.ctypes.data_as(C.c_void_p))
        assert [f"{x:08x}" for x in out] == v["out"]


@pytest.mark.parametrize("purpose,sub", [(1, 0), (3, 0), (4, 7), (6, 123456)])
def test_random_streams_match_oracle(gpu_ops, orc, purpose, sub):
    n, chain, step = 1000, 17, 9001
    got = gpu_ops.test_random(SEED, chain, step, purpose, sub, n).cpu().numpy()
    want = np.zeros((n, 4))
    for k in range(n):
        orc.lib().orc_dev_random(SEED, chain, step, k, purpose, sub, want[k])
    assert (got[:, :2] == want[:, :2]).all(), "uniforms must be bit-exact"
    assert ((got[:, :2] >= 0) & (got[:, :2] < 1)).all()
    assert_close(got[:, 2:], want[:, 2:], tol=1e-13, what="Box-Muller normals")


def test_expcos_draws_match_oracle(gpu_ops, orc):
    rng = np.random.default_rng(1)
    n = 4096
    xp, xm = rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n)
    for beta in (0.3, 1.0, 4.0):
        got = gpu_ops.test_expcos(SEED, 3, 5, beta, dev(xp), dev(xm)).cpu().numpy()
        want = np.array([orc.lib().orc_dev_expcos_draw(SEED, 3, 5, k, beta, xp[k], xm[k]) for k in range(n)])
        assert_angles_close(got, want, what=f"ExpCos beta={beta}")
        assert ((got >= -np.pi) & (got < np.pi + 1e-15)).all()


def test_step_envelope_draws_match_oracle(gpu_ops, orc):
    """The sampler of the heat-bath sweeps at moderate concentrations (device_common.hpp, "tabulated step envelope";
    oracle dev_vonmises_table): the accepted angle is a linear function of random bits, so device and oracle agree to
    rounding wherever they take the same accept / reject decisions -- i.e. everywhere."""
    rng = np.random.default_rng(3)
    n = 8192
    for scale in (2.0, 4.0, 0.7, 0.0):
        xp, xm = rng.uniform(-3 * np.pi, 3 * np.pi, n), rng.uniform(-3 * np.pi, 3 * np.pi, n)
        xm[:4] = xp[:4] + np.array([0.0, np.pi, 2 * np.pi, np.pi - 1e-9])   # kappa = scale, 0, scale, ~0
        got = gpu_ops.test_vs_draw(SEED, 2, 9, scale, dev(xp), dev(xm)).cpu().numpy()
        want = np.array([orc.lib().orc_dev_vs_draw(SEED, 2, 9, k, scale, xp[k], xm[k]) for k in range(n)])
        assert_angles_close(got, want, tol=1e-13, what=f"tabulated step envelope, scale {scale}")
        assert ((got >= -np.pi - 1e-15) & (got < np.pi + 1e-15)).all()


def test_expsin2_draws_match_oracle(gpu_ops, orc):
    rng = np.random.default_rng(2)
    n = 4096
    sigma = rng.uniform(0.05, 40.0, n)
    got = gpu_ops.test_expsin2(SEED, 1, 2, dev(sigma)).cpu().numpy()
    want = np.array([orc.lib().orc_dev_expsin2_draw(SEED, 1, 2, k, sigma[k]) for k in range(n)])
    assert_close(got, want, what="ExpSin2")
    assert (np.abs(got) < np.pi).all()


# ---- 1-D actions ---------------------------------------------------------------------------------------
PATH_CASES = [
    ("harmonic", dict(M=128, T_final=4.0, m0=1.0, mu2=1.0)),
    ("quartic", dict(M=16, T_final=4.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0)),
    ("quartic", dict(M=1000, T_final=125.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0)),
    ("rotor", dict(M=16, T_final=4.0, m0=0.25)),
    ("rotor", dict(M=4096, T_final=512.0, m0=0.25)),
    ("rotor", dict(M=65536, T_final=8192.0, m0=0.25)),
]
KINDS = {"harmonic": 0, "quartic": 1, "rotor": 2}


def make_path(orc, name, p):
    from mlmcpathintegral_amd import abi
    k = KINDS[name]
    act = abi.path_action(k, p["M"], p["T_final"], p.get("m0", 1.0), p.get("mu2", 1.0), p.get("lam", 0.0), p.get("x0", 0.0))
    return act, orc.Action(k, **p)


@pytest.mark.parametrize("name,p", PATH_CASES)
def test_path_evaluate_force_qoi(gpu_ops, orc, name, p):
    act, A = make_path(orc, name, p)
    M, B = p["M"], 3
    rng = np.random.default_rng(M)
    x = np.vstack([seq(M)] + [rng.uniform(-3, 3, M) for _ in range(B - 1)])
    xd = dev(x)
    S = gpu_ops.path_evaluate(act, xd).cpu().numpy()
    F = gpu_ops.path_force(act, xd).cpu().numpy()
    X2 = gpu_ops.qoi_xsquared(xd).cpu().numpy()
    chi = gpu_ops.qoi_susceptibility(xd, p["T_final"]).cpu().numpy()
    L = orc.lib()
    for b in range(B):
        assert_close(S[b], A.evaluate(x[b]), what="evaluate")
        assert_close(F[b], A.force(x[b]), what="force")
        assert_close(X2[b], L.orc_qoi_xsquared(x[b], M), what="QoIXsquared")
        assert_close(chi[b], L.orc_qoi_susceptibility(x[b], M, p["T_final"]), tol=1e-10, what="QoISusceptibility")


def test_path_golden_vectors(gpu_ops, golden):
    """The survey's known answers straight through the ABI (no oracle in between)."""
    from mlmcpathintegral_amd import abi
    g = golden["rotor_M16"]
    act = abi.path_action(2, 16, 4.0, 0.25)
    x = dev(seq(16)[None, :])
    assert_close(gpu_ops.path_evaluate(act, x).item(), g["S"])
    assert_close(gpu_ops.path_force(act, x).cpu().numpy()[0, :4], g["force_0_3"])
    g = golden["quartic_M16"]
    act = abi.path_action(1, 16, 4.0, 1.0, 1.0, 1.0, 1.0)
    assert_close(gpu_ops.path_evaluate(act, x).item(), g["S"])
    assert_close(gpu_ops.path_force(act, x).cpu().numpy()[0, :4], g["force_0_3"])
    assert_close(gpu_ops.qoi_xsquared(x).item(), g["X2"])


def test_path_initialise(gpu_ops, orc):
    act, A = make_path(orc, "rotor", dict(M=1024, T_final=128.0, m0=0.25))
    x = gpu_ops.path_initialise(act, 3, SEED, chain0=5).cpu().numpy()
    for b in range(3):
        assert (x[b] == A.dev_initialise(SEED, 5 + b)).all() or np.max(np.abs(x[b] - A.dev_initialise(SEED, 5 + b))) < 1e-15
    act, A = make_path(orc, "quartic", dict(M=64, T_final=8.0, lam=1.0, x0=1.0))
    assert (gpu_ops.path_initialise(act, 2, SEED).cpu().numpy() == 0).all()


HMC_CASES = [
    # name, params, nt, dt, B   -- covers the periodic register-resident geometry (M = NT*R), every R,
    # and the segmented geometry with halo nt+1 (M not a multiple of 64, and M > 8192)
    ("harmonic", dict(M=128, T_final=4.0, m0=1.0, mu2=1.0), 100, 0.0558, 4),
    ("quartic", dict(M=1024, T_final=128.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0), 10, 0.09, 3),
    ("quartic", dict(M=8192, T_final=1024.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0), 7, 0.05, 2),
    ("quartic", dict(M=1000, T_final=125.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0), 12, 0.08, 2),
    ("quartic", dict(M=32768, T_final=4096.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0), 20, 0.05, 2),
    ("rotor", dict(M=64, T_final=8.0, m0=0.25), 10, 0.1, 5),
    ("rotor", dict(M=4096, T_final=512.0, m0=0.25), 10, 0.1, 2),
    ("rotor", dict(M=65536, T_final=8192.0, m0=0.25), 25, 0.1, 2),
    ("rotor", dict(M=330, T_final=40.0, m0=0.25), 5, 0.1, 2),
]


@pytest.mark.parametrize("name,p,nt,dt,B", HMC_CASES)
def test_hmc_trajectories_match_oracle(gpu_ops, orc, name, p, nt, dt, B):
    """Three consecutive HMCSampler::draw calls per chain; same Philox momenta / accept uniforms."""
    act, A = make_path(orc, name, p)
    M = p["M"]
    rng = np.random.default_rng(M + nt)
    x0 = rng.uniform(-1.0, 1.0, (B, M)) if name != "rotor" else rng.uniform(-np.pi, np.pi, (B, M))
    xd = dev(x0)
    hmc = gpu_ops.PathHMC(act, B, nt, dt, n_rep=1, seed=SEED, chain0=40)
    xo = x0.copy()
    for t in range(3):
        acc = hmc.draw(xd).cpu().numpy()
        en = hmc.energies.cpu().numpy()
        for b in range(B):
            a, e, dH = A.dev_hmc_trajectory(xo[b], nt, dt, SEED, 40 + b, t)
            assert_close(en[b], e, tol=1e-11, what=f"energies t={t} b={b}")
            assert acc[b] == a, f"accept flag t={t} b={b} (dH={dH})"
        assert_close(xd.cpu().numpy(), xo, tol=1e-10, what=f"state after draw {t}")
    assert hmc.n_total == 3


@pytest.mark.parametrize("M,R_expected", [(128, 2), (256, 4), (512, 8), (1024, 16)])
def test_hmc_register_geometries(gpu_ops, orc, M, R_expected):
    """Large batches select R = M/64 sites per thread (one wave per chain): exercise every
    register-resident variant and spot-check chains of the batch against the oracle."""
    act, A = make_path(orc, "quartic", dict(M=M, T_final=M / 8.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0))
    B, nt, dt = 2100, 6, 0.1
    g = torch.Generator().manual_seed(M)
    x0 = (torch.rand((B, M), generator=g, dtype=torch.float64) * 2 - 1)
    xd = x0.cuda()
    hmc = gpu_ops.PathHMC(act, B, nt, dt, seed=SEED, chain0=0)
    acc = hmc.draw(xd).cpu().numpy()
    got = xd.cpu().numpy()
    for b in (0, 1, 63, 1050, 2099):
        xo = x0[b].numpy().copy()
        a, e, _ = A.dev_hmc_trajectory(xo, nt, dt, SEED, b, 0)
        assert acc[b] == a
        assert_close(got[b], xo, tol=1e-10, what=f"chain {b}")
        assert_close(hmc.energies[b].cpu().numpy(), e, tol=1e-11)


@pytest.mark.parametrize("name,p,nt,dt,n_rep", [
    ("harmonic", dict(M=128, T_final=4.0, m0=1.0, mu2=1.0), 100, 0.0558, 1),
    ("quartic", dict(M=1024, T_final=128.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0), 10, 0.2, 2),
    ("rotor", dict(M=512, T_final=64.0, m0=0.25), 12, 0.15, 1),
    ("rotor", dict(M=65536, T_final=8192.0, m0=0.25), 6, 0.05, 1),   # segmented: per-draw fallback inside the ABI
])
def test_hmc_run_equals_repeated_draws(gpu_ops, orc, name, p, nt, dt, n_rep):
    """mlmcpi_path_hmc_run (all draws + QoIs in one launch for short paths) reproduces repeated
    mlmcpi_path_hmc_draw + QoI calls, and matches the oracle trajectory by trajectory."""
    act, A = make_path(orc, name, p)
    M, B, n_draws = p["M"], 3, 5
    qoi_kind = 2 if name == "rotor" else 1
    rng = np.random.default_rng(M)
    x0 = rng.uniform(-1, 1, (B, M))
    xa, xb = dev(x0), dev(x0)
    run = gpu_ops.PathHMC(act, B, nt, dt, n_rep=n_rep, seed=SEED, chain0=3)
    q, cnt = gpu_ops.path_hmc_run(run, xa, n_draws, qoi_kind)
    ref = gpu_ops.PathHMC(act, B, nt, dt, n_rep=n_rep, seed=SEED, chain0=3)
    qs, total = [], torch.zeros(B, dtype=torch.int32, device="cuda")
    for d in range(n_draws):
        total += ref.draw(xb)
        qs.append(gpu_ops.qoi_susceptibility(xb, p["T_final"]) if qoi_kind == 2 else gpu_ops.qoi_xsquared(xb))
    # same arithmetic and the same Philox counters; the compiler may contract a*b+c differently in the two
    # kernels, so equality is to rounding (in practice bit-exact for most cases), the accept counts exact
    assert torch.equal(cnt, total), (cnt, total)
    assert_close(xa.cpu().numpy(), xb.cpu().numpy(), tol=1e-12, what="state after n_draws")
    assert_close(q.cpu().numpy(), torch.stack(qs, dim=1).cpu().numpy(), tol=1e-13, what="per-draw QoIs")
    if M <= 1024:  # oracle cross-check
        xo = x0.copy()
        for b in range(B):
            for d in range(n_draws):
                a = 0
                for r in range(n_rep):
                    if a:
                        break
                    a, _, _ = A.dev_hmc_trajectory(xo[b], nt, dt, SEED, 3 + b, d * n_rep + r)
        assert_close(xa.cpu().numpy(), xo, tol=1e-9, what="state vs oracle")


def test_hmc_n_rep_short_circuit(gpu_ops, orc):
    """hmcsampler.cc:10-12: `accept = accept or single_step()` stops integrating after the first
    accepted repetition; repetition r uses Philox step traj0 + r."""
    act, A = make_path(orc, "quartic", dict(M=256, T_final=32.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0))
    B, nt, dt, n_rep = 6, 10, 0.30, 3  # large dt: a mix of accepts and rejects (about 50 % each)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-1, 1, (B, 256))
    xd = dev(x0)
    hmc = gpu_ops.PathHMC(act, B, nt, dt, n_rep=n_rep, seed=SEED, chain0=0)
    xo = x0.copy()
    seen = set()
    for d in range(4):
        acc = hmc.draw(xd).cpu().numpy()
        for b in range(B):
            a = 0
            for r in range(n_rep):
                if a:
                    break
                a, _, _ = A.dev_hmc_trajectory(xo[b], nt, dt, SEED, b, d * n_rep + r)
            assert acc[b] == a
            seen.add(int(a))
        assert_close(xd.cpu().numpy(), xo, tol=1e-10)
    assert seen == {0, 1}, "test should exercise both outcomes"


@pytest.mark.parametrize("name,M,B", [("harmonic", 64, 3), ("quartic", 64, 3), ("quartic", 1000, 2), ("quartic", 32768, 2),
                                      ("rotor", 64, 3), ("rotor", 4096, 2)])
def test_twolevel_step_matches_oracle(gpu_ops, orc, name, M, B):
    """TwoLevelMetropolisStep::draw with the action's conditioned fine action (Gaussian fill-in for the
    oscillators, ExpSin2 fill-in for the rotor): trial state, the three action differences, accept flags and
    the updated fine state against the oracle's device-order restatement."""
    p = dict(M=M, T_final=M / 8.0, m0=1.0, mu2=1.0)
    if name == "rotor":
        p = dict(M=M, T_final=M / 8.0, m0=0.25)
    if name == "quartic":
        p.update(lam=1.0, x0=1.0)
    pc = dict(p, M=M // 2)
    fine, F = make_path(orc, name, p)
    coarse, Cc = make_path(orc, name, pc)
    rng = np.random.default_rng(M)
    step = gpu_ops.PathTwoLevelStep(fine, coarse, B, seed=SEED, chain0=9)
    theta0 = rng.normal(0.5, 0.6, (B, M))
    if name == "rotor":
        theta0 = rng.uniform(-np.pi, np.pi, (B, M))
    step.set_state(dev(theta0))
    theta = theta0.copy()
    seen = set()
    for t in range(6):
        # coarse proposals: the coarse points of the current fine state (accepted: the fill-in replaces a
        # rougher path) alternate with very rough ones (rejected) -> both outcomes occur
        xc = theta[:, ::2] + rng.normal(0, 1e-3 if t % 2 == 0 else 1.0, (B, M // 2))
        acc = step.draw(dev(xc)).cpu().numpy()
        terms = step.terms.cpu().numpy()
        for b in range(B):
            a, want = F.dev_twolevel_draw(Cc, xc[b], theta[b], SEED, 9 + b, t)
            assert_close(terms[b], want, tol=1e-10, scale=max(1.0, float(np.max(np.abs(want)))), what=f"action differences t={t} b={b}")
            assert acc[b] == a, (t, b, want)
            seen.add(int(a))
        assert_close(step.theta.cpu().numpy(), theta, tol=1e-12, what=f"fine state after draw {t}")
    if name in ("quartic", "rotor"):  # (for the HO the Gaussian fill-in is exact: these proposals are all accepted)
        assert seen == {0, 1}


def test_config5_hierarchy_matches_oracle(gpu_ops, orc):
    """BASELINE config 5 shape: quartic double well, 5 levels, finest M_lat = 32768 (T_final = 4096, a = 0.125),
    hierarchical sampling (sampler/hierarchicalsampler.cc:55-81): HMC on the coarsest level (M_lat = 2048), then
    one TwoLevelMetropolisStep per finer level.  Two full hierarchical draws, chain by chain against the
    oracle's device-order restatement of the same composition."""
    from mlmcpathintegral_amd import abi
    L, M0, T, B, nt, dt = 5, 32768, 4096.0, 2, 100, 0.095
    par = dict(m0=1.0, mu2=1.0, lam=1.0, x0=1.0)
    acts, oras = [], []
    for ell in range(L):
        M = M0 >> ell
        acts.append(abi.path_action(1, M, T, 1.0, 1.0, 1.0, 1.0))
        oras.append(orc.Action(orc.QUARTIC, M=M, T_final=T, **par))
    rng = np.random.default_rng(5)
    # a smooth fine-level start (coarse random walk, interpolated) so that acceptance is not degenerate
    coarse = np.cumsum(rng.normal(0, 0.2, (B, M0 >> 6)), axis=1)
    coarse -= np.linspace(0, 1, M0 >> 6)[None, :] * (coarse[:, -1:] - coarse[:, :1])  # periodic
    x_fine = 1.0 + 0.3 * np.repeat(coarse, 64, axis=1) + rng.normal(0, 0.02, (B, M0))
    hmc = gpu_ops.PathHMC(acts[-1], B, nt, dt, seed=SEED, chain0=0)
    steps = [gpu_ops.PathTwoLevelStep(acts[ell], acts[ell + 1], B, seed=SEED + ell + 1, chain0=0) for ell in range(L - 1)]
    state = [dev(x_fine[:, :: (1 << ell)]) for ell in range(L)]          # copy_from_fine down the hierarchy
    host = [x_fine[:, :: (1 << ell)].copy() for ell in range(L)]
    outcomes = []
    for draw in range(2):
        for ell in range(1, L):                                          # hierarchicalsampler.cc:57-60
            state[ell].copy_(state[ell - 1][:, ::2])
            host[ell] = host[ell - 1][:, ::2].copy()
        acc = hmc.draw(state[L - 1]).cpu().numpy()
        for b in range(B):
            a, _, _ = oras[-1].dev_hmc_trajectory(host[L - 1][b], nt, dt, SEED, b, draw)
            assert a == acc[b]
        assert_close(state[L - 1].cpu().numpy(), host[L - 1], tol=1e-9, what="coarsest level after HMC")
        for ell in range(L - 2, -1, -1):
            steps[ell].set_state(state[ell])
            acc = steps[ell].draw(state[ell + 1]).cpu().numpy()
            state[ell].copy_(steps[ell].theta)
            for b in range(B):
                a, terms = oras[ell].dev_twolevel_draw(oras[ell + 1], host[ell + 1][b], host[ell][b], SEED + ell + 1, b, draw)
                assert a == acc[b], (draw, ell, b, terms)
                outcomes.append(a)
            assert_close(state[ell].cpu().numpy(), host[ell], tol=1e-9, what=f"level {ell} after two-level step")
    assert len(outcomes) == 2 * (L - 1) * B


def test_twolevel_step_errors(gpu_ops):
    from mlmcpathintegral_amd import abi
    with pytest.raises(abi.MlmcpiError, match="half the sites"):
        gpu_ops.PathTwoLevelStep(abi.path_action(2, 64, 8.0, 0.25), abi.path_action(0, 32, 8.0, 0.25), 1).draw(
            torch.zeros((1, 32), dtype=torch.float64, device="cuda"))
    with pytest.raises(abi.MlmcpiError, match="half the sites"):
        gpu_ops.PathTwoLevelStep(abi.path_action(1, 64, 8.0), abi.path_action(1, 16, 8.0), 1).draw(
            torch.zeros((1, 16), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("M,B", [(16, 2), (128, 3), (4096, 2), (10000, 2)])
def test_rotor_sweeps_match_oracle(gpu_ops, orc, M, B):
    act, A = make_path(orc, "rotor", dict(M=M, T_final=M / 8.0, m0=0.25))
    rng = np.random.default_rng(M)
    x0 = rng.uniform(-np.pi, np.pi, (B, M))
    xd, scratch = dev(x0), torch.empty((B, M), dtype=torch.float64, device="cuda")
    xo = x0.copy()
    sweep = 0
    # (the overrelaxation sweeps of a launch are one closed form, up to 16 per launch: (10, 1) is one launch, (16, 0) the
    # deepest, (21, 1) two; MLMCPI_OR_KERNEL=block -- sweep by sweep -- is checked against it below)
    for n_or, n_hb in ((1, 0), (6, 0), (0, 1), (3, 2), (5, 1), (10, 1), (16, 0), (21, 1)):
        xd.copy_(dev(xo))  # every case starts from identical inputs on both sides
        x0_case = xo.copy()
        gpu_ops.path_sweep_draw(act, xd, scratch, n_or, n_hb, SEED, 7, sweep)
        for b in range(B):
            for s in range(n_or + n_hb):
                A.dev_sweep(xo[b], s >= n_or, SEED, 7 + b, sweep + s)
        tol = HB_TOL[min(n_hb, 2)]
        assert_angles_close(xd.cpu().numpy(), xo, tol=tol, what=f"rotor sweeps ({n_or},{n_hb})")
        if n_hb == 0:   # the closed form against the sweep-by-sweep kernel and against its statement in numpy
            from mlmcpathintegral_amd import abi
            from closed_form import angle_diff, rotor_overrelax_closed_form
            abi.set_option("MLMCPI_OR_KERNEL", "block")
            try:
                xs = dev(x0_case)
                gpu_ops.path_sweep_draw(act, xs, scratch, n_or, 0, SEED, 7, sweep)
            finally:
                abi.set_option("MLMCPI_OR_KERNEL", "")
            assert_angles_close(xs.cpu().numpy(), xd.cpu().numpy(), tol=HB_TOL[0], what=f"rotor closed form vs sweeps ({n_or},0)")
            if n_or <= 16:
                for b in range(B):
                    assert angle_diff(xd[b].cpu().numpy(), rotor_overrelax_closed_form(x0_case[b], n_or)).max() <= 1e-13
        sweep += n_or + n_hb


def test_rotor_heat_bath_behind_the_last_overrelaxation_launch(gpu_ops):
    """path_sweep_impl lets the last overrelaxation launch of a draw run the heat-bath sweep as its last sweep; the draws
    are those of separate launches (MLMCPI_OR_HEAT=split), bit for bit, at the BASELINE size too."""
    from mlmcpathintegral_amd import abi
    for M, B in ((4096, 3), (65536, 4)):
        act = abi.path_action(abi.ROTOR, M, M / 8.0, 0.25)
        x0 = gpu_ops.path_initialise(act, B, SEED, 1)
        for n_or, n_hb in ((10, 1), (8, 1), (3, 2), (1, 1), (17, 1)):
            res = {}
            for mode in ("split", "fused"):
                abi.set_option("MLMCPI_OR_HEAT", mode)
                try:
                    x = x0.clone()
                    gpu_ops.path_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 1, 40)
                    res[mode] = x
                finally:
                    abi.set_option("MLMCPI_OR_HEAT", "")
            assert torch.equal(res["split"], res["fused"]), (M, n_or, n_hb)


@pytest.mark.parametrize("M,B", [(64, 2), (4096, 3), (65536, 2)])
def test_rotor_draw_qoi_record_in_one_call_equals_the_three_steps(gpu_ops, M, B):
    """mlmcpi_path_sweep_draw_qoi: the state is the plain draw's, bit for bit; the susceptibility summed inside the last
    launch equals mlmcpi_qoi_susceptibility on it to rounding; the moments recorded in the same call are those of
    mlmcpi_stats_accumulate on that value."""
    from mlmcpathintegral_amd import abi
    act = abi.path_action(abi.ROTOR, M, M / 8.0, 0.25)
    x0 = gpu_ops.path_initialise(act, B, SEED, 9)
    for n_or, n_hb in ((10, 1), (8, 1), (3, 0), (0, 2)):
        plain = x0.clone()
        gpu_ops.path_sweep_draw(act, plain, torch.empty_like(plain), n_or, n_hb, SEED, 9, 50)
        acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        src = x0.clone()
        res, _, q = gpu_ops.path_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 9, 50, acc=acc)
        assert torch.equal(res, plain), (n_or, n_hb)
        ref = gpu_ops.qoi_susceptibility(plain, M / 8.0)
        assert_close(q.cpu().numpy(), ref.cpu().numpy(), tol=1e-10, what=f"fused susceptibility ({n_or},{n_hb})")
        want = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        gpu_ops.stats_accumulate(want, q)
        assert torch.equal(acc, want)
        src = x0.clone()   # (the first call was free to use its input as a work buffer)
        res2, _, q2 = gpu_ops.path_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 9, 50)
        assert torch.equal(q2, q) and torch.equal(res2, plain)


def test_path_sweep_unsupported_action(gpu_ops):
    """action/action.hh:73-96: heat bath / overrelaxation are errors for HO and quartic."""
    from mlmcpathintegral_amd import abi
    act = abi.path_action(1, 64, 8.0, 1.0, 1.0, 1.0, 1.0)
    x = torch.zeros((1, 64), dtype=torch.float64, device="cuda")
    with pytest.raises(abi.MlmcpiError, match="not implemented"):
        gpu_ops.path_sweep_draw(act, x, torch.empty_like(x), 1, 1, 1, 0, 0)


# ---- 2-D actions -------------------------------------------------------------------------------------------
def make_lattice(orc, kind, Mt, Mx, **kw):
    from mlmcpathintegral_amd import abi
    if kind == "gff":
        return abi.lattice_action(3, Mt, Mx, mass=kw["mass"]), orc.Action(orc.GFF, Mt=Mt, Mx=Mx, mass=kw["mass"])
    return abi.lattice_action(4, Mt, Mx, beta=kw["beta"]), orc.Action(orc.SCHWINGER, Mt=Mt, Mx=Mx, beta=kw["beta"])


@pytest.mark.parametrize("kind,Mt,Mx,kw", [("gff", 4, 4, dict(mass=10.0)), ("gff", 64, 64, dict(mass=10.0)),
                                           ("gff", 30, 30, dict(mass=2.0)), ("schwinger", 4, 4, dict(beta=1.0)),
                                           ("schwinger", 16, 6, dict(beta=2.5)), ("schwinger", 130, 70, dict(beta=1.0))])
def test_lattice_evaluate_force_qoi(gpu_ops, orc, kind, Mt, Mx, kw):
    act, A = make_lattice(orc, kind, Mt, Mx, **kw)
    n, B = A.size, 2
    rng = np.random.default_rng(n)
    x = np.vstack([seq(n), rng.uniform(-3, 3, n)])
    xd = dev(x)
    S = gpu_ops.lattice_evaluate(act, xd).cpu().numpy()
    F = gpu_ops.lattice_force(act, xd).cpu().numpy()
    L = orc.lib()
    for b in range(B):
        assert_close(S[b], A.evaluate(x[b]), what="evaluate")
        assert_close(F[b], A.force(x[b]), what="force")
    if kind == "gff":
        q = gpu_ops.qoi_phi_squared(xd).cpu().numpy()
        for b in range(B):
            assert_close(q[b], L.orc_qoi_2d_phi_squared(x[b], n))
    else:
        plaq = gpu_ops.qoi_avg_plaquette(xd, Mt, Mx).cpu().numpy()
        chi = gpu_ops.qoi_2d_susceptibility(xd, Mt, Mx).cpu().numpy()
        for b in range(B):
            assert_close(plaq[b], L.orc_qoi_avg_plaquette(x[b], Mt, Mx))
            assert_close(chi[b], L.orc_qoi_2d_susceptibility(x[b], Mt, Mx), tol=1e-10)


def test_lattice_golden_vectors(gpu_ops, golden):
    from mlmcpathintegral_amd import abi
    g = golden["schwinger_4x4"]
    act = abi.lattice_action(4, 4, 4, beta=1.0)
    x = dev(seq(32)[None, :])
    assert_close(gpu_ops.lattice_evaluate(act, x).item(), g["S"])
    assert_close(gpu_ops.qoi_avg_plaquette(x, 4, 4).item(), g["plaq"])
    assert_close(gpu_ops.lattice_force(act, x).cpu().numpy()[0, :4], g["force_0_3"])
    g = golden["gff_4x4"]
    act = abi.lattice_action(3, 4, 4, mass=10.0)
    x = dev(seq(16)[None, :])
    assert_close(gpu_ops.lattice_evaluate(act, x).item(), g["S"])
    assert_close(gpu_ops.qoi_phi_squared(x).item(), g["phi2"])
    assert_close(gpu_ops.lattice_force(act, x).cpu().numpy()[0, :4], g["force_0_3"])


def test_lattice_initialise(gpu_ops, orc):
    for kind, kw in (("schwinger", dict(beta=1.0)), ("gff", dict(mass=10.0))):
        act, A = make_lattice(orc, kind, 16, 16, **kw)
        x = gpu_ops.lattice_initialise(act, 2, SEED, chain0=3).cpu().numpy()
        for b in range(2):
            assert_close(x[b], A.dev_initialise(SEED, 3 + b), tol=1e-13)


SWEEP_CASES = [
    # kind, Mt, Mx, params, B  -- tiny lattices (buffer wraps around the torus several times), one tile,
    # several tiles with ragged edges, rectangular Schwinger lattices
    ("gff", 2, 2, dict(mass=1.0), 2),      # smallest lattice: every neighbour is the same vertex twice
    ("gff", 4, 4, dict(mass=10.0), 2),
    ("gff", 16, 16, dict(mass=10.0), 3),
    ("gff", 64, 64, dict(mass=10.0), 2),   # specialised overrelaxation kernel, 1 x 2 tiles
    ("gff", 128, 128, dict(mass=3.0), 1),  # 2 x 4 tiles
    ("gff", 130, 130, dict(mass=10.0), 1),
    ("gff", 96, 96, dict(mass=10.0), 2),   # 3 x 3 tiles of 32 x 32: register-block kernels where 64 x 64 tiles do not divide (r04)
    ("gff", 160, 160, dict(mass=3.0), 1),  # 5 x 5 tiles (the GFF action takes square lattices only, gffaction.hh:169-173)
    ("schwinger", 2, 2, dict(beta=1.0), 2),
    ("schwinger", 2, 8, dict(beta=0.7), 1),
    ("schwinger", 4, 4, dict(beta=1.0), 2),
    ("schwinger", 16, 16, dict(beta=1.0), 3),
    ("schwinger", 6, 10, dict(beta=4.0), 2),
    ("schwinger", 64, 32, dict(beta=1.0), 2),
    ("schwinger", 128, 64, dict(beta=1.0), 2),  # 2 x 2 tiles of the specialised overrelaxation kernel
    ("schwinger", 130, 70, dict(beta=1.0), 1),
    ("schwinger", 64, 64, dict(beta=2.0), 2),    # one 64 x 64 tile of the 4 x 4 register-block kernel: the buffer wraps onto itself
    ("schwinger", 192, 128, dict(beta=1.0), 1),  # 3 x 2 tiles of it
    ("schwinger", 128, 128, dict(beta=3.0), 1),  # 4 < 2 beta <= 16: the step envelope since r05 (wrapped Cauchy before)
    ("schwinger", 128, 128, dict(beta=5.0), 1),
    ("schwinger", 128, 128, dict(beta=9.0), 1),  # 2 beta > 16: the fused launch with the wrapped-Cauchy sampler (r04)
    ("schwinger", 16, 16, dict(beta=0.0), 2),    # flat conditionals: kappa is clamped, the draw is uniform
    ("schwinger", 16, 16, dict(beta=40.0), 2),   # sharply peaked conditionals (kappa up to 80)
    ("gff", 16, 16, dict(mass=0.0), 2),          # massless field: kappa = 4
]


@pytest.mark.parametrize("fuse", [1, 3, 0])  # 0 = library default (up to 6 overrelaxation sweeps per launch)
@pytest.mark.parametrize("kind,Mt,Mx,kw,B", SWEEP_CASES)
def test_lattice_sweeps_match_oracle(gpu_ops, orc, kind, Mt, Mx, kw, B, fuse):
    act, A = make_lattice(orc, kind, Mt, Mx, **kw)
    n = A.size
    rng = np.random.default_rng(n)
    x0 = rng.uniform(-np.pi, np.pi, (B, n))
    xd, scratch = dev(x0), torch.empty((B, n), dtype=torch.float64, device="cuda")
    xo = x0.copy()
    sweep = 100
    # (5, 1), (2, 1), (4, 2) on 64 x 64-divisible lattices >= 128: the last overrelaxation launch takes the heat-bath sweep along
    for n_or, n_hb in ((1, 0), (5, 0), (0, 1), (2, 1), (4, 2), (6, 1), (5, 1)):
        xd.copy_(dev(xo))  # every case starts from identical inputs on both sides
        gpu_ops.lattice_sweep_draw(act, xd, scratch, n_or, n_hb, SEED, 11, sweep, fuse=fuse)
        for b in range(B):
            for s in range(n_or + n_hb):
                A.dev_sweep(xo[b], s >= n_or, SEED, 11 + b, sweep + s)
        sweep += n_or + n_hb
        got = xd.cpu().numpy()
        if kind == "schwinger":
            assert_angles_close(got, xo, tol=HB_TOL[min(n_hb, 2)], what=f"sweeps ({n_or},{n_hb}) fuse={fuse}")
        else:
            assert_close(got, xo, tol=1e-11, what=f"sweeps ({n_or},{n_hb}) fuse={fuse}")


@pytest.mark.parametrize("Mt,Mx,B,beta", [(64, 64, 2, 1.0), (128, 128, 2, 1.0), (192, 128, 1, 2.0), (128, 256, 2, 3.0), (128, 192, 1, 4.0),
                                          (128, 128, 2, 5.5), (192, 128, 1, 8.0), # the step envelope's upper range
                                          (128, 128, 2, 9.5),                     # 2 beta > 16: wrapped Cauchy
                                          (64, 32, 2, 1.0), (192, 96, 1, 1.0),   # the last two: 64 x 32 tiles
                                          # r05: lattices no tile divides (masked edge tiles): 64 x 32 tiles + heat-bath launch
                                          # (an extent below 128), the one-launch draw on 64 x 64 tiles (both >= 128)
                                          (130, 70, 2, 1.0), (70, 130, 1, 1.0), (200, 136, 1, 1.0), (130, 198, 2, 2.5)])
def test_closed_form_overrelaxation_matches_oracle(gpu_ops, orc, Mt, Mx, B, beta):
    """schwinger_perm_kernel / schwinger_perm_heat_kernel (K <= 10 overrelaxation sweeps per launch as ONE fixed permutation
    of the plaquettes; the default where 64 x 64 tiles divide the lattice) at every depth the register-block kernels do not
    reach -- one plane (K <= 7), two planes (K >= 8), several launches (n_or > 10), with and without the heat bath behind --
    against the oracle's sweeps, against the closed form in numpy (tests/closed_form.py), and the same draw with the
    heat bath in a launch of its own, bit for bit."""
    from mlmcpathintegral_amd import abi
    from closed_form import angle_diff, schwinger_overrelax_closed_form
    act, A = make_lattice(orc, "schwinger", Mt, Mx, beta=beta)
    rng = np.random.default_rng(Mt + Mx)
    x0 = rng.uniform(-np.pi, np.pi, (B, A.size))
    sweep = 40
    for n_or, n_hb in ((7, 0), (8, 0), (9, 0), (10, 0), (10, 1), (8, 1), (13, 1), (23, 0)):
        xd = dev(x0)
        gpu_ops.lattice_sweep_draw(act, xd, torch.empty_like(xd), n_or, n_hb, SEED, 3, sweep)
        got = xd.cpu().numpy()
        xo = x0.copy()
        for b in range(B):
            for s in range(n_or + n_hb):
                A.dev_sweep(xo[b], s >= n_or, SEED, 3 + b, sweep + s)
        assert_angles_close(got, xo, tol=HB_TOL[min(n_hb, 2)], what=f"closed form ({n_or},{n_hb})")
        if n_hb == 0 and n_or <= 10:   # one launch: the numpy statement of the same sums
            for b in range(B):
                err = angle_diff(got[b], schwinger_overrelax_closed_form(x0[b], Mt, Mx, n_or)).max()
                assert err <= 1e-13, f"({n_or},0): device vs numpy closed form {err:.3e}"
        if n_hb and Mt >= 128 and Mx >= 128:
            abi.set_option("MLMCPI_OR_HEAT", "split")
            try:
                xs = dev(x0)
                gpu_ops.lattice_sweep_draw(act, xs, torch.empty_like(xs), n_or, n_hb, SEED, 3, sweep)
            finally:
                abi.set_option("MLMCPI_OR_HEAT", "")
            assert torch.equal(xs, xd), f"({n_or},{n_hb}): closed form + heat bath in one launch != two launches"
        sweep += n_or + n_hb


@pytest.mark.parametrize("Mt,Mx,B,beta", [(128, 128, 3, 1.0), (192, 128, 2, 2.0), (256, 128, 2, 0.3), (1024, 1024, 2, 1.0),
                                          # r04: beyond the step envelope's range (2 beta = 16 since r05) the fused launch draws
                                          # from the wrapped-Cauchy envelope
                                          (128, 192, 2, 3.0), (128, 128, 2, 4.0), (192, 128, 2, 5.0), (128, 128, 2, 8.0), (128, 192, 2, 10.0),
                                          (256, 256, 2, 40.0),
                                          (200, 136, 2, 1.0), (130, 262, 1, 1.0)])   # r05: masked edge tiles of the fused launch
def test_overrelaxation_and_heat_bath_in_one_launch_equal_two_launches(gpu_ops, Mt, Mx, B, beta):
    """schwinger_or_heat_kernel<K> (the last K <= 5 overrelaxation sweeps of a draw, the heat-bath sweep behind them and
    the QoI in one launch) against the same draw with the heat bath in a launch of its own (MLMCPI_OR_HEAT=split): states
    bit for bit, for every depth K, with a second heat-bath sweep behind, and the fused QoI to rounding (its partial sums
    run over 64 x 64 instead of 64 x 32 tiles)."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=beta)
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 2)
    cases = [(1, 1), (2, 1), (3, 1), (4, 2), (5, 1), (10, 1)] if Mt < 1024 else [(5, 1), (10, 1)]
    for n_or, n_hb in cases:
        res = {}
        for mode in ("split", "fused"):
            abi.set_option("MLMCPI_OR_HEAT", mode)
            try:
                x = x0.clone()
                gpu_ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 2, 31)
                src = x0.clone()
                xq, _, q = gpu_ops.lattice_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 2, 31, 1)
                res[mode] = (x, xq.clone(), q.clone())
            finally:
                abi.set_option("MLMCPI_OR_HEAT", "")
        assert torch.equal(res["split"][0], res["fused"][0]), f"({n_or},{n_hb}): state differs"
        assert torch.equal(res["split"][1], res["fused"][1]) and torch.equal(res["fused"][0], res["fused"][1])
        assert_close(res["fused"][2].cpu().numpy(), res["split"][2].cpu().numpy(), tol=1e-13, what=f"fused QoI ({n_or},{n_hb})")
        assert_close(res["fused"][2].cpu().numpy(), gpu_ops.qoi_avg_plaquette(res["fused"][0], Mt, Mx).cpu().numpy(), tol=1e-12, what="QoI")
    with pytest.raises(abi.MlmcpiError):
        abi.set_option("MLMCPI_OR_HEAT", "sideways")


@pytest.mark.parametrize("Mt,Mx,B", [(128, 128, 2), (256, 192, 3), (1024, 1024, 1)])
def test_wide_workgroups_of_the_fused_launch_change_nothing(gpu_ops, Mt, Mx, B):
    """Launches with at most one workgroup per CU give the heat-bath part of schwinger_or_heat_kernel sixteen waves
    (1024-thread workgroups; the register-block part stays on its 7 or 8): same draws, bit for bit, at every depth."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 4)
    # "" at these sizes: the library's own plan -- wide workgroups, and the whole draw in one launch for 6 <= n_or <= 10
    for n_or, n_hb in ([(1, 1), (2, 1), (3, 1), (4, 2), (5, 1), (6, 1), (7, 2), (8, 1), (9, 1), (10, 1), (11, 1)] if Mt < 1024 else [(10, 1), (7, 1)]):
        res = {}
        for mode in ("narrow", "wide", ""):
            abi.set_option("MLMCPI_OR_HEAT", mode)
            try:
                src = x0.clone()
                x, _, q = gpu_ops.lattice_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 4, 17, 1)
                res[mode] = (x.clone(), q.clone())
            finally:
                abi.set_option("MLMCPI_OR_HEAT", "")
        assert torch.equal(res["narrow"][0], res["wide"][0]) and torch.equal(res["narrow"][0], res[""][0]), f"({n_or},{n_hb})"
        assert_close(res["wide"][1].cpu().numpy(), res["narrow"][1].cpu().numpy(), tol=1e-13, what="QoI")


@pytest.mark.parametrize("M,B,mass", [(128, 3, 3.0), (192, 2, 10.0), (512, 2, 10.0),
                                      (96, 3, 10.0), (64, 2, 3.0), (160, 2, 10.0),    # r04: 32 x 32 tiles
                                      (130, 2, 10.0), (200, 2, 3.0)])                  # r05: ... with masked edge tiles
def test_gff_overrelaxation_and_heat_bath_in_one_launch_equal_two_launches(gpu_ops, M, B, mass):
    """gff_or_heat_kernel<K> against the two launches it replaces (MLMCPI_OR_HEAT=split): field bit for bit, phi^2 to rounding.
    (Lattices of 96, 64, 160 sites: the register-block kernels on 32 x 32 tiles, gff_or_heat_kernel<K, 32>.)"""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.GFF, M, M, mass=mass)
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 2)
    for n_or, n_hb in ([(1, 1), (2, 1), (3, 1), (4, 2), (5, 1), (10, 1)] if M < 512 else [(10, 1)]):
        res = {}
        for mode in ("split", "fused"):
            abi.set_option("MLMCPI_OR_HEAT", mode)
            try:
                x = x0.clone()
                gpu_ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 2, 31)
                src = x0.clone()
                xq, _, q = gpu_ops.lattice_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 2, 31, 3)
                res[mode] = (x, xq.clone(), q.clone())
            finally:
                abi.set_option("MLMCPI_OR_HEAT", "")
        assert torch.equal(res["split"][0], res["fused"][0]), f"({n_or},{n_hb}): field differs"
        assert torch.equal(res["split"][1], res["fused"][1]) and torch.equal(res["fused"][0], res["fused"][1])
        assert_close(res["fused"][2].cpu().numpy(), res["split"][2].cpu().numpy(), tol=1e-13, what=f"fused phi^2 ({n_or},{n_hb})")
        assert_close(res["fused"][2].cpu().numpy(), gpu_ops.qoi_phi_squared(res["fused"][0]).cpu().numpy(), tol=1e-12, what="phi^2")


@pytest.mark.parametrize("tile", ["64x64x256", "128x32x256", "128x64x256"])
def test_heatbath_retry_pool_with_several_passes_per_phase(gpu_ops, tile):
    """tiles with more than 5 x 256 cells per colour phase make heatbath_cells take several passes, i.e.
    several uses of the LDS retry pool per phase (counters and entry arrays are reused): the result must be the one
    of the default geometry (one pass per phase), bit for bit, at the small kappa (many retries) of beta = 0.3 too."""
    from mlmcpathintegral_amd import abi
    for beta, Mt in ((1.0, 256), (0.3, 128)):
        act = abi.lattice_action(4, Mt, Mt, beta=beta)
        x = gpu_ops.lattice_initialise(act, 3, SEED, chain0=0)
        want, scratch = x.clone(), torch.empty_like(x)
        abi.set_option("MLMCPI_OR_KERNEL", "block")   # (the sweep-by-sweep family: the tile option below selects its generic kernels)
        try:
            gpu_ops.lattice_sweep_draw(act, want, scratch, 1, 3, SEED, 0, 7, fuse=1)
        finally:
            abi.set_option("MLMCPI_OR_KERNEL", "")
        abi.set_option("MLMCPI_SWEEP_TILE", tile)
        try:
            got = x.clone()
            gpu_ops.lattice_sweep_draw(act, got, scratch, 1, 3, SEED, 0, 7, fuse=1)
        finally:
            abi.set_option("MLMCPI_SWEEP_TILE", "")
        assert torch.equal(want, got), f"tile {tile}, beta {beta}: heat-bath result depends on the tile geometry"


@pytest.mark.parametrize("kind,Mt,Mx,kw", [("schwinger", 8, 6, dict(beta=1.0)), ("schwinger", 6, 6, dict(beta=3.0)), ("schwinger", 6, 8, dict(beta=6.0)), ("schwinger", 8, 8, dict(beta=10.0)),
                                           ("gff", 8, 8, dict(mass=3.0)), ("rotor", 32, 0, dict(T_final=4.0, m0=0.25)),
                                           ("rotor", 16, 0, dict(T_final=1.0, m0=1.0))])
def test_site_at_a_time_updates_match_oracle(gpu_ops, orc, kind, Mt, Mx, kw):
    """Action::heatbath_update / overrelaxation_update(state, l) (action/action.hh:73-96) through
    mlmcpi_{path,lattice}_site_updates: a shuffled index list walked sequentially per chain -- the loop of
    overrelaxedheatbathsampler.cc:8-31 with random_order -- against the oracle's single-site updates applied in the
    same order; both samplers (beta = 1, 3, 6: tabulated step envelope, beta = 10 / m0/a = 16: wrapped Cauchy)."""
    from mlmcpathintegral_amd import abi
    B = 3
    if kind == "rotor":
        act = abi.path_action(abi.ROTOR, Mt, kw["T_final"], kw["m0"])
        A = orc.Action(orc.ROTOR, M=Mt, T_final=kw["T_final"], m0=kw["m0"])
        update = gpu_ops.path_site_updates
    else:
        act, A = make_lattice(orc, kind, Mt, Mx, **kw)
        update = gpu_ops.lattice_site_updates
    n = A.size
    rng = np.random.default_rng(n)
    x0 = rng.uniform(-np.pi, np.pi, (B, n))
    for heat in (False, True):
        sites = rng.permutation(n).astype(np.uint32)
        xd = dev(x0)
        update(act, xd, torch.from_numpy(sites.view(np.int32)).cuda(), heat, SEED, 5, 21)
        update(act, xd, int(sites[3]), heat, SEED, 5, 22)   # the single-site form
        want = x0.copy()
        for b in range(B):
            for l in sites:
                A.dev_site_update(want[b], l, heat, SEED, 5 + b, 21)
            A.dev_site_update(want[b], sites[3], heat, SEED, 5 + b, 22)
        got = xd.cpu().numpy()
        if kind == "gff":
            assert_close(got, want, tol=1e-11, what=f"site updates heat={heat}")
        else:
            assert_angles_close(got, want, tol=HB_TOL[2] if heat else 1e-12, what=f"site updates heat={heat}")


def test_site_updates_over_the_colour_classes_equal_a_sweep(gpu_ops):
    """The random-number contract of the site-at-a-time entry points: Philox (site, chain, step), as in the sweeps -- the
    links of the four colour classes visited one class after the other, with the sweep's step, ARE the sweep."""
    from mlmcpathintegral_amd import abi
    Mt = 64
    act = abi.lattice_action(4, Mt, Mt, beta=1.0)
    x = gpu_ops.lattice_initialise(act, 2, SEED, 0)
    for heat in (False, True):
        a, b = x.clone(), x.clone()
        gpu_ops.lattice_sweep_draw(act, a, torch.empty_like(a), 0 if heat else 1, 1 if heat else 0, SEED, 0, 9, fuse=1)
        l = np.arange(2 * Mt * Mt)
        mu, v = l & 1, l >> 1
        colour = np.where(mu == 0, (v // Mt) & 1, 2 + ((v % Mt) & 1))
        order = np.concatenate([l[colour == c] for c in range(4)]).astype(np.int32)
        gpu_ops.lattice_site_updates(act, b, torch.from_numpy(order).cuda(), heat, SEED, 0, 9)
        d = (a - b).abs()
        d = torch.minimum(d, (d - 2 * np.pi).abs())
        assert float(d.max()) < 1e-12, f"heat={heat}: {float(d.max())}"


@pytest.mark.parametrize("rt,rx", [(2, 2), (2, 1), (1, 2)])
def test_level_transfers_match_oracle(gpu_ops, orc, golden, rt, rx):
    """Action::copy_from_fine / copy_from_coarse (Schwinger, GFF, 1-D paths) against the oracle, plus the
    survey's recorded coarse links of the 4 x 4 Schwinger state."""
    from mlmcpathintegral_amd import abi
    L = orc.lib()
    Mt, Mx, B = 12 * rt, 10 * rx, 2          # fine extents
    rng = np.random.default_rng(rt * 10 + rx)
    fine_act = abi.lattice_action(4, Mt, Mx, beta=1.0)
    fine = rng.uniform(-np.pi, np.pi, (B, 2 * Mt * Mx))
    got = gpu_ops.lattice_copy_from_fine(fine_act, rt, rx, dev(fine)).cpu().numpy()
    for b in range(B):
        want = np.zeros(2 * (Mt // rt) * (Mx // rx))
        L.orc_schwinger_copy_from_fine(Mt // rt, Mx // rx, rt, rx, fine[b], want)
        assert_angles_close(got[b], want, tol=1e-15, what="Schwinger copy_from_fine")
    coarse = rng.uniform(-np.pi, np.pi, got.shape)
    fd = dev(fine)
    gpu_ops.lattice_copy_from_coarse(fine_act, rt, rx, dev(coarse), fd)
    for b in range(B):
        want = fine[b].copy()
        L.orc_schwinger_copy_from_coarse(Mt // rt, Mx // rx, rt, rx, coarse[b], want)
        assert (fd[b].cpu().numpy() == want).all(), "Schwinger copy_from_coarse (untouched links must stay)"
    if (rt, rx) == (2, 2):
        # GFF lives on square lattices only (gffaction.hh:169-173), i.e. CoarsenBoth hierarchies
        Mg = 24
        gff = abi.lattice_action(3, Mg, Mg, mass=1.0)
        phi = rng.normal(size=(B, Mg * Mg))
        gotc = gpu_ops.lattice_copy_from_fine(gff, 2, 2, dev(phi)).cpu().numpy()
        pd = dev(phi)
        cnew = rng.normal(size=gotc.shape)
        gpu_ops.lattice_copy_from_coarse(gff, 2, 2, dev(cnew), pd)
        for b in range(B):
            want = np.zeros(gotc.shape[1]); f = phi[b].copy()
            L.orc_gff_transfer(Mg // 2, Mg // 2, 2, 2, f, want, 1)
            assert (gotc[b] == want).all()
            L.orc_gff_transfer(Mg // 2, Mg // 2, 2, 2, f, cnew[b].copy(), 0)
            assert (pd[b].cpu().numpy() == f).all()
        g = golden["schwinger_4x4"]
        x = seq(32)
        x[0:2] = g["after_overrelax_0_then_1"]
        x[2:4] = g["after_heatbath_2_then_3"]
        c = gpu_ops.lattice_copy_from_fine(abi.lattice_action(4, 4, 4, beta=1.0), 2, 2, dev(x[None, :])).cpu().numpy()[0]
        assert_close(c[:4], g["copy_from_fine_first_four_coarse_links"], tol=1e-15)
        xf = rng.normal(size=(B, 64))
        assert (gpu_ops.path_copy_from_fine(dev(xf)).cpu().numpy() == xf[:, ::2]).all()
        tgt = dev(xf)
        newc = rng.normal(size=(B, 32))
        gpu_ops.path_copy_from_coarse(dev(newc), tgt)
        want = xf.copy(); want[:, ::2] = newc
        assert (tgt.cpu().numpy() == want).all()


def test_pingpong_sweeps_equal_copy_form(gpu_ops):
    from mlmcpathintegral_amd import abi
    for act, n in ((abi.lattice_action(4, 64, 32, beta=1.0), 2 * 64 * 32), (abi.lattice_action(3, 32, 32, mass=5.0), 32 * 32)):
        x = gpu_ops.lattice_initialise(act, 3, SEED)
        for n_or, n_hb in ((1, 0), (2, 1), (3, 3), (10, 1)):
            a, b = x.clone(), x.clone()
            gpu_ops.lattice_sweep_draw(act, a, torch.empty_like(a), n_or, n_hb, SEED, 0, 5)
            res, other = gpu_ops.lattice_sweep_draw_pingpong(act, b, torch.empty_like(b), n_or, n_hb, SEED, 0, 5)
            assert torch.equal(res, a) and res.data_ptr() != other.data_ptr()


def test_invalid_arguments_are_errors(gpu_ops):
    """Bad sizes, null pointers and unsupported geometries return an error status (never a fault)."""
    import ctypes as C
    from mlmcpathintegral_amd import abi
    z = C.c_void_p(0)
    x = torch.zeros((1, 64), dtype=torch.float64, device="cuda")
    p = C.c_void_p(x.data_ptr())
    act = abi.path_action(1, 64, 8.0, 1.0, 1.0, 1.0, 1.0)
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_evaluate", C.byref(act), p, 0, p, z)          # B = 0
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_evaluate", C.byref(act), z, 1, p, z)          # null state
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_force", C.byref(act), p, p, 1, z)             # in-place force
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_evaluate", C.byref(abi.path_action(1, 1, 8.0)), p, 1, p, z)   # M_lat < 2
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_evaluate", C.byref(abi.path_action(1, 64, -1.0)), p, 1, p, z)  # T_final <= 0
    with pytest.raises(abi.MlmcpiError):
        abi.call("mlmcpi_path_evaluate", C.byref(abi.path_action(4, 64, 8.0)), p, 1, p, z)   # 2-D kind on a path call
    with pytest.raises(abi.MlmcpiError, match="too long"):
        gpu_ops.PathHMC(abi.path_action(2, 65536, 8192.0, 0.25), 1, 5000, 0.1)              # nt beyond the fused halo
    with pytest.raises(abi.MlmcpiError):
        y = torch.zeros((1, 128), dtype=torch.float64, device="cuda")
        q = C.c_void_p(y.data_ptr())
        abi.call("mlmcpi_lattice_sweep_draw", C.byref(abi.lattice_action(4, 8, 8, beta=1.0)), q, q, 1, 1, 0, 1, 0, 0, 0, z)  # state == scratch


def test_sweep_rejects_odd_lattice(gpu_ops):
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(4, 5, 4, beta=1.0)
    x = torch.zeros((1, 40), dtype=torch.float64, device="cuda")
    with pytest.raises(abi.MlmcpiError, match="even"):
        gpu_ops.lattice_sweep_draw(act, x, torch.empty_like(x), 1, 0, 1, 0, 0)
    with pytest.raises(abi.MlmcpiError, match="squared"):
        gpu_ops.lattice_evaluate(abi.lattice_action(3, 8, 4, mass=1.0), x)


@pytest.mark.parametrize("kind,Mt,Mx,kw", [("gff", 16, 16, dict(mass=10.0)), ("schwinger", 16, 8, dict(beta=1.0))])
def test_lattice_hmc_matches_oracle(gpu_ops, orc, kind, Mt, Mx, kw):
    act, A = make_lattice(orc, kind, Mt, Mx, **kw)
    n, B, nt, dt = A.size, 3, 8, 0.05
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1, 1, (B, n))
    xd = dev(x0)
    hmc = gpu_ops.LatticeHMC(act, B, nt, dt, seed=SEED, chain0=2)
    xo = x0.copy()
    for t in range(3):
        acc = hmc.draw(xd).cpu().numpy()
        en = hmc.energies.cpu().numpy()
        for b in range(B):
            a, e, _ = A.dev_hmc_trajectory(xo[b], nt, dt, SEED, 2 + b, t)
            assert_close(en[b], e, tol=1e-11)
            assert acc[b] == a
        assert_close(xd.cpu().numpy(), xo, tol=1e-10)


# ---- full-size properties (BASELINE sizes; the oracle would take too long) -----------------------------
def test_schwinger_1024_properties(gpu_ops, orc, golden):
    """1024 x 1024, beta = 1: (i) overrelaxation sweeps conserve the action; (ii) the result is
    independent of how many sweeps are fused per launch (halo logic at scale, bit-exact); (iii) chain b
    of a batch equals the same chain run alone; (iv) QoIs of the uniform start agree with the oracle."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(4, 1024, 1024, beta=1.0)
    B = 2
    x = gpu_ops.lattice_initialise(act, B, SEED, chain0=0)
    A = orc.Action(orc.SCHWINGER, Mt=1024, Mx=1024, beta=1.0)
    x0 = x[0].cpu().numpy()
    assert_close(gpu_ops.lattice_evaluate(act, x)[0].item(), A.evaluate(x0), tol=1e-12)
    assert_close(gpu_ops.qoi_2d_susceptibility(x, 1024, 1024)[0].item(),
                 orc.lib().orc_qoi_2d_susceptibility(x0, 1024, 1024), tol=1e-9)
    S0 = gpu_ops.lattice_evaluate(act, x).cpu().numpy()
    a, b1 = x.clone(), x.clone()
    scratch = torch.empty_like(x)
    # The sweep-by-sweep kernels first (MLMCPI_OR_KERNEL=block and the others below): one arithmetic, bit-identical results
    # whatever the launch plan.  The default -- overrelaxation in closed form -- is compared with them further down.
    abi.set_option("MLMCPI_OR_KERNEL", "block")
    try:
        gpu_ops.lattice_sweep_draw(act, a, scratch, 4, 2, SEED, 0, 0, fuse=1)
        gpu_ops.lattice_sweep_draw(act, b1, scratch, 4, 2, SEED, 0, 0, fuse=3)
        assert torch.equal(a, b1), "fused and unfused sweeps must agree bit for bit"
        # library default of that family (up to 6 overrelaxation sweeps per launch, 4 x 4 register-block kernel)
        d4 = x.clone()
        gpu_ops.lattice_sweep_draw(act, d4, scratch, 4, 2, SEED, 0, 0, fuse=0)
        assert torch.equal(a, d4), "the default fusion depth must not change the result"
        for n in (5, 6, 10):  # every depth of the 4 x 4 kernel (10 sweeps: 5 + 5) against single-sweep launches
            u, v = x.clone(), x.clone()
            gpu_ops.lattice_sweep_draw(act, u, scratch, n, 0, SEED, 0, 0, fuse=1)
            gpu_ops.lattice_sweep_draw(act, v, scratch, n, 0, SEED, 0, 0, fuse=0)
            assert torch.equal(u, v), f"{n} overrelaxation sweeps: default launch plan differs from single sweeps"
        single = x[1:2].clone()
        gpu_ops.lattice_sweep_draw(act, single, torch.empty_like(single), 4, 2, SEED, 1, 0, fuse=2)
        assert torch.equal(single[0], a[1]), "a chain's result must not depend on the batch it runs in"
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    # the specialised overrelaxation kernel (compile-time tile geometry) against the generic one
    abi.set_option("MLMCPI_SWEEP_TILE", "64x32x256")
    try:
        gen = x.clone()
        gpu_ops.lattice_sweep_draw(act, gen, scratch, 4, 2, SEED, 0, 0, fuse=2)
    finally:
        abi.set_option("MLMCPI_SWEEP_TILE", "")
    assert torch.equal(a, gen), "specialised and generic sweep kernels must agree bit for bit"
    # the LDS-resident kernel
    abi.set_option("MLMCPI_OR_KERNEL", "lds")
    try:
        lds4 = x.clone()
        gpu_ops.lattice_sweep_draw(act, lds4, scratch, 4, 2, SEED, 0, 0, fuse=4)
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    assert torch.equal(a, lds4), "register-tiled and LDS-resident overrelaxation kernels must agree bit for bit"
    abi.set_option("MLMCPI_OR_KERNEL", "patch")  # 2 x 2 register blocks on 64 x 32 tiles (the default is 4 x 4 on 64 x 64)
    try:
        p4 = x.clone()
        gpu_ops.lattice_sweep_draw(act, p4, scratch, 4, 2, SEED, 0, 0, fuse=4)
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    assert torch.equal(a, p4), "4 x 4 and 2 x 2 register-block overrelaxation kernels must agree bit for bit"
    # The default: K overrelaxation sweeps in closed form (schwinger_perm_kernel / schwinger_perm_heat_kernel).  The same map
    # up to the rounding of 4 K additions: to 1e-12 against the sweep-by-sweep kernels for every launch plan (the two
    # heat-bath sweeps behind take the same decisions); bit for bit among launches of the same depth.
    plans = {}
    for fuse in (1, 2, 3, 0):
        plans[fuse] = x.clone()
        gpu_ops.lattice_sweep_draw(act, plans[fuse], scratch, 4, 2, SEED, 0, 0, fuse=fuse)
        assert_angles_close(plans[fuse].cpu().numpy(), a.cpu().numpy(), tol=HB_TOL[0], what=f"closed form, fuse={fuse}")
    for n in (5, 6, 10, 13):  # one launch (13: 7 + 6) against single-sweep launches of the register-block kernel
        u, v = x.clone(), x.clone()
        abi.set_option("MLMCPI_OR_KERNEL", "block")
        try:
            gpu_ops.lattice_sweep_draw(act, u, scratch, n, 0, SEED, 0, 0, fuse=1)
        finally:
            abi.set_option("MLMCPI_OR_KERNEL", "")
        gpu_ops.lattice_sweep_draw(act, v, scratch, n, 0, SEED, 0, 0, fuse=0)
        assert_angles_close(v.cpu().numpy(), u.cpu().numpy(), tol=HB_TOL[0], what=f"closed form, {n} sweeps")
    single = x[1:2].clone()
    gpu_ops.lattice_sweep_draw(act, single, torch.empty_like(single), 4, 2, SEED, 1, 0, fuse=2)
    assert torch.equal(single[0], plans[2][1]), "a chain's result must not depend on the batch it runs in"
    # the reference's draw, 10 + 1 sweeps: ONE launch (closed form + heat bath) against the register-block plan (5; 5 + heat bath)
    one, two = x.clone(), x.clone()
    gpu_ops.lattice_sweep_draw(act, one, scratch, 10, 1, SEED, 0, 50)
    abi.set_option("MLMCPI_OR_KERNEL", "block")
    try:
        gpu_ops.lattice_sweep_draw(act, two, scratch, 10, 1, SEED, 0, 50)
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    assert_angles_close(one.cpu().numpy(), two.cpu().numpy(), tol=HB_TOL[1], what="10 + 1 draw: one launch vs the register-block plan")
    c = x.clone()
    gpu_ops.lattice_sweep_draw(act, c, scratch, 5, 0, SEED, 0, 0, fuse=2)
    S1 = gpu_ops.lattice_evaluate(act, c).cpu().numpy()
    assert_close(S1, S0, tol=1e-11, what="action conservation under overrelaxation")
    assert ((a >= -np.pi - 1e-12) & (a < np.pi + 1e-12)).all()


def test_rotor_65536_and_gff_512_properties(gpu_ops):
    from mlmcpathintegral_amd import abi
    act = abi.path_action(2, 65536, 8192.0, 0.25)
    x = gpu_ops.path_initialise(act, 2, SEED)
    S0 = gpu_ops.path_evaluate(act, x).cpu().numpy()
    scratch = torch.empty_like(x)
    gpu_ops.path_sweep_draw(act, x, scratch, 6, 0, SEED, 0, 0)
    assert_close(gpu_ops.path_evaluate(act, x).cpu().numpy(), S0, tol=1e-11, what="rotor OR conserves S")
    act = abi.lattice_action(3, 512, 512, mass=10.0)
    phi = gpu_ops.lattice_initialise(act, 2, SEED)
    S0 = gpu_ops.lattice_evaluate(act, phi).cpu().numpy()
    a, b = phi.clone(), phi.clone()
    scratch = torch.empty_like(phi)
    gpu_ops.lattice_sweep_draw(act, a, scratch, 6, 1, SEED, 0, 0, fuse=1)
    gpu_ops.lattice_sweep_draw(act, b, scratch, 6, 1, SEED, 0, 0, fuse=4)
    assert torch.equal(a, b)
    abi.set_option("MLMCPI_SWEEP_TILE", "64x32x256")  # generic kernels
    try:
        gen = phi.clone()
        gpu_ops.lattice_sweep_draw(act, gen, scratch, 6, 1, SEED, 0, 0, fuse=2)
    finally:
        abi.set_option("MLMCPI_SWEEP_TILE", "")
    assert torch.equal(a, gen), "specialised and generic GFF kernels must agree bit for bit"
    abi.set_option("MLMCPI_OR_KERNEL", "lds")
    try:
        lds4 = phi.clone()
        gpu_ops.lattice_sweep_draw(act, lds4, scratch, 6, 1, SEED, 0, 0, fuse=4)
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    assert torch.equal(a, lds4), "register-tiled and LDS-resident GFF overrelaxation kernels must agree bit for bit"
    abi.set_option("MLMCPI_OR_KERNEL", "patch")  # 2 x 2 register blocks on 64 x 32 tiles (the default is 4 x 4 on 64 x 64)
    try:
        p4 = phi.clone()
        gpu_ops.lattice_sweep_draw(act, p4, scratch, 6, 1, SEED, 0, 0, fuse=4)
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
    assert torch.equal(a, p4), "4 x 4 and 2 x 2 register-block GFF overrelaxation kernels must agree bit for bit"
    for n in (5, 6, 10):  # every depth of the 4 x 4 kernel against single-sweep launches
        u, v = phi.clone(), phi.clone()
        gpu_ops.lattice_sweep_draw(act, u, scratch, n, 0, SEED, 0, 0, fuse=1)
        gpu_ops.lattice_sweep_draw(act, v, scratch, n, 0, SEED, 0, 0, fuse=0)
        assert torch.equal(u, v), f"GFF, {n} overrelaxation sweeps: default launch plan differs from single sweeps"
    c = phi.clone()
    gpu_ops.lattice_sweep_draw(act, c, scratch, 6, 0, SEED, 0, 0, fuse=3)
    assert_close(gpu_ops.lattice_evaluate(act, c).cpu().numpy(), S0, tol=1e-11, what="GFF OR conserves S")


@pytest.mark.parametrize("Mt,Mx,rt,rx,beta,B", [(16, 8, 2, 1, 2.0, 3), (8, 16, 1, 2, 2.0, 3), (64, 32, 2, 1, 1.0, 2),
                                                (32, 64, 1, 2, 6.0, 2), (256, 256, 2, 1, 1.0, 1), (4, 4, 1, 2, 0.7, 2),
                                                (8, 8, 2, 2, 2.0, 3), (16, 12, 2, 2, 0.8, 2), (8, 8, 2, 2, 10.0, 3),
                                                (64, 64, 2, 2, 4.0, 1), (128, 128, 2, 2, 12.0, 1)])
def test_schwinger_twolevel_step_matches_oracle(gpu_ops, orc, Mt, Mx, rt, rx, beta, B):
    """TwoLevelMetropolisStep::draw on the Schwinger lattice (copy_from_coarse, conditioned fine action, copy_from_fine,
    the three action differences, Metropolis test) against the oracle's device-order restatement: accept flags, action
    differences and the fine state after every draw.  Semi-coarsening: uniform pair shifts + ExpCos fill-in; both
    directions: Bessel-product fill-in up to beta = 8, its Gaussian-mixture approximation beyond."""
    from mlmcpathintegral_amd import abi
    fine, F = make_lattice(orc, "schwinger", Mt, Mx, beta=beta)
    coarse, Cc = make_lattice(orc, "schwinger", Mt // rt, Mx // rx, beta=beta / (rt * rx))   # quenchedschwingeraction.hh coarse_action
    rng = np.random.default_rng(Mt * 7 + Mx)
    step = gpu_ops.LatticeTwoLevelStep(fine, coarse, B, seed=SEED, chain0=4)
    theta0 = rng.uniform(-np.pi, np.pi, (B, 2 * Mt * Mx)) * (0.15 if Mt * Mx <= 256 else 1.0)
    step.set_state(dev(theta0))
    theta = theta0.copy()
    seen = set()
    for t in range(5):
        # proposals: the coarse image of the current state, perturbed a little (mostly accepted on small
        # lattices) or a lot (mostly rejected)
        base = gpu_ops.lattice_copy_from_fine(fine, rt, rx, dev(theta)).cpu().numpy()
        pc = base + rng.normal(0, 0.02 if t % 2 == 0 else 1.5, base.shape)
        acc = step.draw(dev(pc)).cpu().numpy()
        terms = step.terms.cpu().numpy()
        for b in range(B):
            a, want = F.dev_lattice_twolevel_draw(Cc, pc[b], theta[b], SEED, 4 + b, t)
            assert_close(terms[b], want, tol=2e-10, scale=max(1.0, float(np.max(np.abs(want)))), what=f"action differences t={t} b={b}")
            assert acc[b] == a, (t, b, want)
            seen.add(int(a))
        assert_angles_close(step.theta.cpu().numpy(), theta, tol=1e-10, what=f"fine state after draw {t}")
        theta = step.theta.cpu().numpy().copy()   # resync (angle wrap at +-pi may differ by 2 pi)
    print("outcomes seen:", seen)
    if Mt * Mx == 128:
        assert seen == {0, 1}


@pytest.mark.parametrize("Mt,Mx,beta,B", [(8, 8, 4.0, 3), (16, 12, 2.0, 2), (8, 8, 80.0, 2), (64, 64, 9.0, 1)])
def test_schwinger_gaussian_cfa_twolevel_step_matches_oracle(gpu_ops, orc, Mt, Mx, beta, B):
    """The two-level step with QuenchedSchwingerGaussianConditionedFineAction (quenchedschwingerconditionedfineaction.cc:
    81-134, 293-327; GaussianFillinDistribution): fill-in, the three action differences and the accept flags against the
    oracle's restatement.  beta = 80 takes the n_offsets = 0 branch of the peak construction."""
    fine, F = make_lattice(orc, "schwinger", Mt, Mx, beta=beta)
    coarse, Cc = make_lattice(orc, "schwinger", Mt // 2, Mx // 2, beta=beta / 4)
    rng = np.random.default_rng(Mt * 11 + Mx)
    step = gpu_ops.LatticeTwoLevelStep(fine, coarse, B, seed=SEED, chain0=2, cfa_kind=1)
    theta0 = rng.uniform(-np.pi, np.pi, (B, 2 * Mt * Mx)) * 0.1
    step.set_state(dev(theta0))
    theta = theta0.copy()
    for t in range(4):
        base = gpu_ops.lattice_copy_from_fine(fine, 2, 2, dev(theta)).cpu().numpy()
        pc = base + rng.normal(0, 0.02 if t % 2 == 0 else 0.8, base.shape)
        acc = step.draw(dev(pc)).cpu().numpy()
        terms = step.terms.cpu().numpy()
        for b in range(B):
            a, want = F.dev_lattice_twolevel_draw(Cc, pc[b], theta[b], SEED, 2 + b, t, cfa_kind=1)
            assert_close(terms[b], want, tol=5e-10, scale=max(1.0, float(np.max(np.abs(want)))), what=f"action differences t={t} b={b}")
            assert acc[b] == a, (t, b, want)
        assert_angles_close(step.theta.cpu().numpy(), theta, tol=1e-10, what=f"fine state after draw {t}")
        theta = step.theta.cpu().numpy().copy()


@pytest.mark.parametrize("Mt,Mx,B,n_or,n_hb", [(1024, 1024, 2, 2, 1), (64, 64, 3, 0, 1), (48, 20, 2, 3, 2), (128, 64, 2, 1, 1),
                                               (200, 136, 2, 10, 1), (1000, 1000, 1, 10, 1), (130, 70, 2, 10, 1)])   # r05: masked edge tiles
def test_fused_qoi_equals_separate_evaluation(gpu_ops, Mt, Mx, B, n_or, n_hb):
    """mlmcpi_lattice_sweep_draw_qoi: the QoI summed inside the draw's last launch equals the stand-alone QoI kernels on
    the same result (average plaquette and Q^2 / 4 pi^2), and the state is the one the plain draw produces, bit for bit;
    a draw that does not end with a heat-bath sweep is refused."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 3)
    plain = x0.clone()
    gpu_ops.lattice_sweep_draw(act, plain, torch.empty_like(plain), n_or, n_hb, SEED, 3, 11)
    for kind, ref in ((1, gpu_ops.qoi_avg_plaquette(plain, Mt, Mx)), (2, gpu_ops.qoi_2d_susceptibility(plain, Mt, Mx))):
        src = x0.clone()
        res, _, q = gpu_ops.lattice_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 3, 11, kind)
        assert torch.equal(res, plain)
        assert_close(q.cpu().numpy(), ref.cpu().numpy(), tol=1e-12 if kind == 1 else 1e-10, what=f"fused QoI {kind}")
    with pytest.raises(abi.MlmcpiError):
        gpu_ops.lattice_sweep_draw_qoi(act, x0, torch.empty_like(x0), x0.clone(), 2, 0, SEED, 3, 11, 1)
    with pytest.raises(abi.MlmcpiError):  # phi^2 is the GFF's QoI
        gpu_ops.lattice_sweep_draw_qoi(act, x0, torch.empty_like(x0), x0.clone(), n_or, n_hb, SEED, 3, 11, 3)


@pytest.mark.parametrize("kind,M,B", [("schwinger", 256, 3), ("schwinger", 48, 2), ("gff", 128, 2)])
def test_draw_qoi_record_in_one_call_equals_the_three_steps(gpu_ops, kind, M, B):
    """mlmcpi_lattice_sweep_draw_qoi_record = mlmcpi_lattice_sweep_draw_qoi + mlmcpi_stats_accumulate: same sample, same
    QoI, same per-chain moments after several samples, bit for bit (the recurrence is the same, on the same value)."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.SCHWINGER, M, M, beta=1.0) if kind == "schwinger" else abi.lattice_action(abi.GFF, M, M, mass=3.0)
    qk = 1 if kind == "schwinger" else 3
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 6)
    out = {}
    for mode in ("three", "one"):
        x, w = x0.clone(), torch.empty_like(x0)
        acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        for t in range(4):
            if mode == "one":
                x, w, q = gpu_ops.lattice_sweep_draw_qoi(act, x, w, x, 3, 1, SEED, 6, 4 * t, qk, acc=acc)
            else:
                x, w, q = gpu_ops.lattice_sweep_draw_qoi(act, x, w, x, 3, 1, SEED, 6, 4 * t, qk)
                gpu_ops.stats_accumulate(acc, q)
        out[mode] = (x.clone(), q.clone(), acc.clone())
    for a, b in zip(out["three"], out["one"]):
        assert torch.equal(a, b)
    assert float(out["one"][2][:, 0].min()) == 4.0
    with pytest.raises(abi.MlmcpiError):   # a draw that does not end with a heat-bath sweep cannot carry the QoI
        gpu_ops.lattice_sweep_draw_qoi(act, x0, torch.empty_like(x0), x0.clone(), 2, 0, SEED, 6, 0, qk, acc=torch.zeros((B, 5), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("M,B,n_or,n_hb", [(512, 3, 5, 1), (64, 2, 0, 1), (20, 2, 3, 2), (192, 2, 1, 1), (96, 2, 10, 1),
                                           (130, 2, 10, 1), (1000, 1, 10, 1)])   # r05: masked edge tiles
def test_fused_gff_qoi_equals_separate_evaluation(gpu_ops, M, B, n_or, n_hb):
    """The same for the GFF action: QoI2DPhiSquared (qoi kind 3) summed inside the heat-bath launch."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.GFF, M, M, mass=10.0)
    x0 = gpu_ops.lattice_initialise(act, B, SEED, 3)
    plain = x0.clone()
    gpu_ops.lattice_sweep_draw(act, plain, torch.empty_like(plain), n_or, n_hb, SEED, 3, 11)
    src = x0.clone()
    res, _, q = gpu_ops.lattice_sweep_draw_qoi(act, src, torch.empty_like(src), src, n_or, n_hb, SEED, 3, 11, 3)
    assert torch.equal(res, plain)
    assert_close(q.cpu().numpy(), gpu_ops.qoi_phi_squared(plain).cpu().numpy(), tol=1e-12, what="fused phi^2")
    with pytest.raises(abi.MlmcpiError):  # the plaquette QoIs belong to the Schwinger action
        gpu_ops.lattice_sweep_draw_qoi(act, x0, torch.empty_like(x0), x0.clone(), n_or, n_hb, SEED, 3, 11, 1)


def test_schwinger_twolevel_step_errors(gpu_ops):
    from mlmcpathintegral_amd import abi
    f = abi.lattice_action(4, 16, 16, beta=1.0)
    with pytest.raises(abi.MlmcpiError, match="invalid coarsening"):
        gpu_ops.LatticeTwoLevelStep(f, abi.lattice_action(4, 16, 16, beta=1.0), 1)
    with pytest.raises(abi.MlmcpiError, match="only the quenched Schwinger"):
        gpu_ops.LatticeTwoLevelStep(abi.lattice_action(3, 16, 16, mass=1.0), abi.lattice_action(3, 8, 8, mass=1.0), 1)


@pytest.mark.parametrize("M,B", [(128, 40), (100, 5), (16, 16), (512, 33)])
def test_ho_exact_sampler_matches_oracle(gpu_ops, orc, M, B):
    """HarmonicOscillatorAction::draw (x = L y, harmonicoscillatoraction.cc:59-66) on the fp64 matrix cores
    (v_mfma_f64_16x16x4_f64; ragged chain and site counts included) against the oracle's device-order draw."""
    from mlmcpathintegral_amd import abi
    p = dict(M=M, T_final=M / 32.0, m0=1.0, mu2=1.0)
    act, A = make_path(orc, "harmonic", p)
    sampler = gpu_ops.HOExactSampler(act, B, seed=SEED, chain0=3)
    Lo = np.zeros((M, M))
    assert orc.lib().orc_ho_cholesky(A.h, Lo.reshape(-1)) == 0
    for step in range(2):
        x = sampler.draw().cpu().numpy()
        for b in range(B):
            want = np.zeros(M)
            orc.lib().orc_dev_exact_draw(A.h, Lo.reshape(-1), want, SEED, 3 + b, step)
            assert_close(x[b], want, tol=1e-11, scale=1.0, what=f"exact draw step {step} chain {b}")
    with pytest.raises(abi.MlmcpiError, match="only for the harmonic oscillator"):
        gpu_ops.HOExactSampler(abi.path_action(abi.QUARTIC, 64, 8.0, 1.0, 1.0, 1.0, 1.0), 4)


@pytest.mark.parametrize("Mt,mass,B", [(4, 2.0, 3), (8, 10.0, 2), (16, 1.0, 2)])
def test_gff_exact_sampler_matches_oracle(gpu_ops, orc, Mt, mass, B):
    """GFFAction::draw by spectral synthesis (batched 2-D FFT) against the oracle's direct O(N^2) evaluation of the
    same Fourier sum with the same Philox normals."""
    from mlmcpathintegral_amd import abi
    act, A = make_lattice(orc, "gff", Mt, Mt, mass=mass)
    s = gpu_ops.GFFExactSampler(act, B, seed=SEED, chain0=2)
    for step in range(2):
        phi = s.draw().cpu().numpy()
        for b in range(B):
            want = np.zeros(Mt * Mt)
            orc.lib().orc_dev_gff_exact_draw(A.h, want, SEED, 2 + b, step)
            assert_close(phi[b], want, tol=1e-12, scale=float(np.max(np.abs(want))), what=f"GFF exact draw {step} chain {b}")
    with pytest.raises(abi.MlmcpiError, match="only for the GFF"):
        gpu_ops.GFFExactSampler(abi.lattice_action(4, 8, 8, beta=1.0), 1)


# ---- the device's angle samplers against the analytic law, with no oracle in between ---------------------------------
# (oracle.cc::dev_vonmises_table was written in lockstep with device_common.hpp, so device == oracle is
# close to a self-comparison there.  Here 2 10^5 DEVICE draws per case go straight against the quadrature CDF of
# p(x) ~ exp(kappa cos(x - centre)) (expcosdistribution.cc:7-21) and the closed form <cos> = I1 / I0.)
def _vonmises_cdf(kappa):
    from scipy import integrate
    grid = np.linspace(-np.pi, np.pi, 8001)
    cdf = integrate.cumulative_trapezoid(np.exp(kappa * (np.cos(grid) - 1.0)), grid, initial=0.0)
    cdf /= cdf[-1]
    return lambda x: np.interp(x, grid, cdf)


_VS_CASES = [(2.0, 0.0), (2.0, 0.45), (2.0, 0.85), (2.0, 1.25), (2.0, 1.65), (2.0, 2.05), (2.0, 2.45), (2.0, 2.8), (2.0, 3.1),
             (4.0, 0.2), (4.0, 1.9), (4.0, 4.5), (0.6, 1.0), (3.0, -7.0), (0.0, 1.0),
             (6.0, 0.3), (6.0, 2.3), (8.0, 0.0), (8.0, 0.9), (8.0, 2.0), (8.0, 2.8), (8.0, 3.3),
             (12.0, 0.5), (12.0, 2.4), (16.0, 0.0), (16.0, 1.0), (16.0, 2.6), (16.0, 3.2)]   # r05: concentrations up to 16


@pytest.mark.gpu
@pytest.mark.parametrize("scale,d", _VS_CASES)
def test_device_step_envelope_draws_follow_the_von_mises_law(gpu_ops, scale, d):
    """mlmcpi_test_vs_draw (the sweeps' sampler at 2 beta, 2 m0 / a <= 16; r05, 4 before) over all eight concentration classes
    (class = floor(32 | |d / 4 pi| mod 1/2 - 1/4 |): d = 0 ... pi walks from class 7 down to class 0)."""
    from scipy import stats
    from scipy.special import i0e, i1e
    from conftest import zcheck
    n = 200000
    x_p = torch.full((n,), 0.4, dtype=torch.float64, device="cuda")
    x_m = x_p + d
    draws = gpu_ops.test_vs_draw(20261004, 3, 11, scale, x_p, x_m).cpu().numpy()
    assert (np.abs(draws) <= np.pi + 1e-12).all()
    kappa = scale * abs(np.cos(0.5 * d))
    centre = 0.4 + 0.5 * d + (np.pi if np.cos(0.5 * d) < 0 else 0.0)
    rel = draws - centre
    rel -= 2 * np.pi * np.round(rel / (2 * np.pi))
    ks = stats.kstest(rel, _vonmises_cdf(kappa))
    assert ks.pvalue > 1e-3, (scale, d, ks)
    # symmetry about the centre (the sign bit) and the first trigonometric moment
    zcheck(f"device vs_draw scale={scale} d={d}: <sin>", float(np.sin(rel).mean()), float(np.sin(rel).std() / np.sqrt(n)), 0.0, gate=4.5)
    zcheck(f"device vs_draw scale={scale} d={d}: <cos>", float(np.cos(rel).mean()), float(np.cos(rel).std() / np.sqrt(n)),
           float(i1e(kappa) / i0e(kappa)), gate=4.5)


@pytest.mark.gpu
@pytest.mark.parametrize("beta,d", [(1.0, 0.7), (1.0, 2.9), (2.5, 0.3), (2.5, 2.2), (6.0, 1.5), (40.0, 0.1), (0.05, 1.0)])
def test_device_expcos_draws_follow_the_von_mises_law(gpu_ops, beta, d):
    """mlmcpi_test_expcos: the wrapped-Cauchy sampler the sweeps use beyond 2 beta = 16 (and the two-level fill-ins always)"""
    from scipy import stats
    from scipy.special import i0e, i1e
    from conftest import zcheck
    n = 200000
    x_p = torch.full((n,), -0.3, dtype=torch.float64, device="cuda")
    x_m = x_p + d
    draws = gpu_ops.test_expcos(77, 1, 5, beta, x_p, x_m).cpu().numpy()
    kappa = 2.0 * beta * abs(np.cos(0.5 * d))
    centre = -0.3 + 0.5 * d + (np.pi if np.cos(0.5 * d) < 0 else 0.0)
    rel = draws - centre
    rel -= 2 * np.pi * np.round(rel / (2 * np.pi))
    ks = stats.kstest(rel, _vonmises_cdf(kappa))
    assert ks.pvalue > 1e-3, (beta, d, ks)
    zcheck(f"device expcos beta={beta} d={d}: <cos>", float(np.cos(rel).mean()), float(np.cos(rel).std() / np.sqrt(n)),
           float(i1e(kappa) / i0e(kappa)), gate=4.5)
