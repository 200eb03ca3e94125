"""GPU: expectation values of the device chains against closed forms and the reference's own runs
(SURVEY.md 8(c)).  Errors are the scatter of independent per-chain means (robust against
autocorrelation).  Every comparison goes through conftest.zcheck: it prints its z-score, records it (the session
writes gpurun_out/zscores.json; the round's copy is under profiles/) and gates it at 3 combined standard errors; the
headline pair (test_schwinger_headline_*) is gated at the north star's 2 sigma."""
import math

import numpy as np
import pytest
import torch

from conftest import HEADLINE_SIGMA, zcheck

pytestmark = pytest.mark.gpu
SEED = 20240607


def _where():
    import inspect
    return inspect.stack()[1].function


def chain_mean_and_error(samples):
    """samples [n_draws, B] -> mean, standard error from the B independent chain means"""
    m = samples.mean(dim=0)
    return float(m.mean()), float(m.std(unbiased=True)) / math.sqrt(m.numel())


def test_schwinger_16x16_expectation_values(gpu_ops, golden):
    from mlmcpathintegral_amd import abi
    ref = golden["reference_runs"]["schwinger_16x16_beta1"]
    act = abi.lattice_action(4, 16, 16, beta=1.0)
    B, burn, n = 256, 200, 1500
    x = gpu_ops.lattice_initialise(act, B, SEED)
    scratch = torch.empty_like(x)
    plaq, q2 = [], []
    sweep = 0
    for k in range(burn + n):
        gpu_ops.lattice_sweep_draw(act, x, scratch, 1, 1, SEED, 0, sweep)  # 1 OR + 1 HB as the reference run
        sweep += 2
        if k >= burn:
            plaq.append(gpu_ops.qoi_avg_plaquette(x, 16, 16))
            q2.append(gpu_ops.qoi_2d_susceptibility(x, 16, 16))
    mp, ep = chain_mean_and_error(torch.stack(plaq))
    mq, eq = chain_mean_and_error(torch.stack(q2))
    print(f"plaq {mp:.6f} +- {ep:.6f} (I1/I0 {ref['plaq_analytic']}, reference {ref['plaq']} +- {ref['plaq_err']})")
    print(f"Q2   {mq:.4f} +- {eq:.4f} (analytic {ref['Q2_analytic']}, reference {ref['Q2']} +- {ref['Q2_err']})")
    zcheck(_where() + ": mp vs ref['plaq_analytic']", mp, ep, ref["plaq_analytic"])
    zcheck(_where() + ": mp vs ref['plaq']", mp, ep, ref["plaq"], ref["plaq_err"])
    zcheck(_where() + ": mq vs ref['Q2_analytic']", mq, eq, ref["Q2_analytic"])
    zcheck(_where() + ": mq vs ref['Q2']", mq, eq, ref["Q2"], ref["Q2_err"])


def test_gff_16x16_expectation_value(gpu_ops, golden):
    from mlmcpathintegral_amd import abi
    ref = golden["reference_runs"]["gff_16x16_mass10"]
    act = abi.lattice_action(3, 16, 16, mass=10.0)
    B, burn, n = 256, 500, 2000
    phi = gpu_ops.lattice_initialise(act, B, SEED)
    scratch = torch.empty_like(phi)
    vals, sweep = [], 0
    for k in range(burn + n):
        gpu_ops.lattice_sweep_draw(act, phi, scratch, 1, 1, SEED, 0, sweep)
        sweep += 2
        if k >= burn:
            vals.append(gpu_ops.qoi_phi_squared(phi))
    m, e = chain_mean_and_error(torch.stack(vals))
    print(f"phi2 {m:.6f} +- {e:.6f} (analytic {ref['phi2_analytic']}, reference {ref['phi2']} +- {ref['phi2_err']})")
    zcheck(_where() + ": m vs ref['phi2_analytic']", m, e, ref["phi2_analytic"])
    zcheck(_where() + ": m vs ref['phi2']", m, e, ref["phi2"], ref["phi2_err"])


def test_harmonic_oscillator_hmc_config1(gpu_ops, golden):
    """BASELINE config 1 (HO, M_lat=128, T=4, nt=100, tuned dt=0.0558) on the device chains."""
    from mlmcpathintegral_amd import abi
    ref = golden["reference_runs"]["config1_ho_M128"]
    act = abi.path_action(0, 128, 4.0, 1.0, 1.0)
    B, burn, n = 512, 300, 400
    x = gpu_ops.path_initialise(act, B, SEED)
    hmc = gpu_ops.PathHMC(act, B, 100, ref["tuned_dt"], seed=SEED)
    vals = []
    for k in range(burn + n):
        # Burn-in with a jittered step size: at fixed trajectory length nt*dt the harmonic modes with
        # omega_k * nt * dt close to a multiple of pi barely move (x -> +-x), so from the reference's
        # all-zero start they would keep zero amplitude.  The reference equilibrates them during its
        # constructor's auto-tuning (100 x 1000 trajectories at varying dt, hmcsampler.cc:77-113).
        hmc.dt = ref["tuned_dt"] * (1.0 + (0.35 * ((k * 7) % 11 - 5) / 5.0 if k < burn else 0.0))
        hmc.draw(x, count_stats=k >= burn)
        if k >= burn:
            vals.append(gpu_ops.qoi_xsquared(x))
    m, e = chain_mean_and_error(torch.stack(vals))
    p_acc = float(hmc.n_accepted.double().mean()) / hmc.n_total
    print(f"x2 {m:.6f} +- {e:.6f} (analytic {ref['analytic_x2']}, reference {ref['numerical']} +- {ref['error']}); "
          f"p_accept {p_acc:.4f} (reference {ref['p_accept']})")
    zcheck(_where() + ": m vs ref['analytic_x2']", m, e, ref["analytic_x2"])
    assert abs(p_acc - ref["p_accept"]) < 0.02


def test_rotor_heatbath_matches_reference_order_chain(gpu_ops, orc):
    """Rotor M=32: device multicolour chains vs the oracle's reference-order (lexicographic,
    mt19937_64) OverrelaxedHeatBathSampler chain: topological susceptibility and <cos dx>."""
    from mlmcpathintegral_amd import abi
    M, T, m0 = 32, 4.0, 0.25
    act = abi.path_action(2, M, T, m0)
    B, burn, n = 256, 200, 1000
    x = gpu_ops.path_initialise(act, B, SEED)
    scratch = torch.empty_like(x)
    chi, cosd, sweep = [], [], 0
    for k in range(burn + n):
        gpu_ops.path_sweep_draw(act, x, scratch, 1, 1, SEED, 0, sweep)
        sweep += 2
        if k >= burn:
            chi.append(gpu_ops.qoi_susceptibility(x, T))
            cosd.append(torch.cos(x[:, 1] - x[:, 0]))
    mchi, echi = chain_mean_and_error(torch.stack(chi))
    mcos, ecos = chain_mean_and_error(torch.stack(cosd))
    L = orc.lib()
    A = orc.Action(orc.ROTOR, M=M, T_final=T, m0=m0)
    hb = L.orc_heatbath_new(A.h, 1, 1, 500, 0)
    y = np.zeros(M)
    rc, rcos = [], []
    for _ in range(40000):
        L.orc_heatbath_draw(hb, y)
        rc.append(L.orc_qoi_susceptibility(y, M, T))
        rcos.append(math.cos(y[1] - y[0]))
    L.orc_heatbath_free(hb)
    rc, rcos = np.array(rc), np.array(rcos)
    # reference chain errors from 40 batch means
    def batch_err(v):
        bm = v.reshape(40, -1).mean(axis=1)
        return bm.std(ddof=1) / math.sqrt(40)
    print(f"chi {mchi:.5f} +- {echi:.5f} vs reference-order {rc.mean():.5f} +- {batch_err(rc):.5f}; "
          f"cos {mcos:.5f} +- {ecos:.5f} vs {rcos.mean():.5f} +- {batch_err(rcos):.5f}")
    zcheck(_where() + ": mchi vs rc.mean()", mchi, echi, rc.mean(), batch_err(rc))
    zcheck(_where() + ": mcos vs rcos.mean()", mcos, ecos, rcos.mean(), batch_err(rcos))


def test_hierarchical_twolevel_chain_samples_fine_distribution(gpu_ops):
    """Coarse-level HMC + TwoLevelMetropolisStep (delayed acceptance, sampler/hierarchicalsampler.cc:55-81)
    must sample the FINE-level distribution: HO M_lat=128 <x^2> against the fine-level closed form."""
    from mlmcpathintegral_amd import abi
    import oracle
    M, T, B = 128, 4.0, 512
    fine, coarse = abi.path_action(0, M, T, 1.0, 1.0), abi.path_action(0, M // 2, T, 1.0, 1.0)
    xc = gpu_ops.path_initialise(coarse, B, SEED)
    hmc = gpu_ops.PathHMC(coarse, B, 50, 0.11, seed=SEED)
    step = gpu_ops.PathTwoLevelStep(fine, coarse, B, seed=SEED)
    vals = []
    for k in range(300 + 300):
        hmc.dt = 0.11 * (1.0 + (0.3 * ((k * 7) % 11 - 5) / 5.0 if k < 300 else 0.0))  # see the HO test above
        hmc.draw(xc)
        # hierarchicalsampler.cc:57-60: the coarse sampler restarts from the coarse points of the fine state
        step.draw(xc)
        xc.copy_(step.theta[:, ::2])
        if k >= 300:
            vals.append(gpu_ops.qoi_xsquared(step.theta))
    m, e = chain_mean_and_error(torch.stack(vals))
    exact = oracle.lib().orc_ho_xsquared_analytical(M, T, 1.0, 1.0)
    print(f"hierarchical two-level <x^2> = {m:.6f} +- {e:.6f} (fine-level analytic {exact:.6f})")
    zcheck(_where() + ": m vs exact", m, e, exact)


def test_rotor_hierarchical_chain_samples_fine_distribution(gpu_ops, orc):
    """Rotor: coarse-level overrelaxed heat bath + TwoLevelMetropolisStep with the ExpSin2 conditioned fine
    action (action/qm/rotorconditionedfineaction.cc) must sample the FINE-level distribution: topological
    susceptibility and <cos(x_1 - x_0)> against a direct fine-level heat-bath chain."""
    from mlmcpathintegral_amd import abi
    M, T, m0, B = 32, 4.0, 0.25, 512
    fine, coarse = abi.path_action(2, M, T, m0), abi.path_action(2, M // 2, T, m0)
    xc = gpu_ops.path_initialise(coarse, B, SEED)
    scratch = torch.empty_like(xc)
    step = gpu_ops.PathTwoLevelStep(fine, coarse, B, seed=SEED + 1)
    step.set_state(gpu_ops.path_initialise(fine, B, SEED + 2))
    chi, cosd, sweep, n_acc = [], [], 0, 0
    burn, n = 200, 1500
    for k in range(burn + n):
        gpu_ops.path_sweep_draw(coarse, xc, scratch, 1, 1, SEED, 0, sweep)
        sweep += 2
        acc = step.draw(xc)
        xc.copy_(step.theta[:, ::2])                                     # hierarchicalsampler.cc:57-60
        if k >= burn:
            n_acc += float(acc.double().mean())
            chi.append(gpu_ops.qoi_susceptibility(step.theta, T))
            cosd.append(torch.cos(step.theta[:, 1] - step.theta[:, 0]))
    mchi, echi = chain_mean_and_error(torch.stack(chi))
    mcos, ecos = chain_mean_and_error(torch.stack(cosd))
    # direct fine-level chain
    x = gpu_ops.path_initialise(fine, B, SEED + 3)
    sc = torch.empty_like(x)
    dcos, dchi = [], []
    for k in range(burn + n):
        gpu_ops.path_sweep_draw(fine, x, sc, 1, 1, SEED + 3, 0, 2 * k)
        if k >= burn:
            dcos.append(torch.cos(x[:, 1] - x[:, 0]))
            dchi.append(gpu_ops.qoi_susceptibility(x, T))
    rcos, rerr = chain_mean_and_error(torch.stack(dcos))
    rchi, rchierr = chain_mean_and_error(torch.stack(dchi))
    print(f"rotor two-level: p_accept {n_acc / n:.3f}; chi_t {mchi:.5f} +- {echi:.5f} vs direct {rchi:.5f} +- {rchierr:.5f}; "
          f"cos {mcos:.5f} +- {ecos:.5f} vs direct {rcos:.5f} +- {rerr:.5f}")
    assert n_acc / n > 0.3
    zcheck(_where() + ": mchi vs rchi", mchi, echi, rchi, rchierr)
    zcheck(_where() + ": mcos vs rcos", mcos, ecos, rcos, rerr)


@pytest.mark.parametrize("rt,rx", [(2, 1), (1, 2), (2, 2)])
def test_schwinger_hierarchical_chain_samples_fine_distribution(gpu_ops, rt, rx):
    """Schwinger 8 x 8, beta = 1.5: coarse-level overrelaxed heat bath on the coarsened lattice (beta/2 for
    semi-coarsening, beta/4 for both directions) + TwoLevelMetropolisStep with the conditioned fine action (ExpCos,
    Bessel product) must sample the FINE-level distribution:
    average plaquette and Q^2 against a direct fine-level heat-bath chain."""
    from mlmcpathintegral_amd import abi
    Mt = Mx = 8
    beta, B = 1.5, 512
    fine, coarse = abi.lattice_action(4, Mt, Mx, beta=beta), abi.lattice_action(4, Mt // rt, Mx // rx, beta=beta / (rt * rx))
    pc = gpu_ops.lattice_initialise(coarse, B, SEED)
    scratch = torch.empty_like(pc)
    step = gpu_ops.LatticeTwoLevelStep(fine, coarse, B, seed=SEED + 1)
    step.set_state(gpu_ops.lattice_initialise(fine, B, SEED + 2))
    burn, n = 200, 1200
    plaq, chi, sweep, n_acc = [], [], 0, 0.0
    for k in range(burn + n):
        gpu_ops.lattice_sweep_draw(coarse, pc, scratch, 1, 1, SEED, 0, sweep)
        sweep += 2
        acc = step.draw(pc)
        pc.copy_(gpu_ops.lattice_copy_from_fine(fine, rt, rx, step.theta))   # hierarchicalsampler.cc:57-60
        if k >= burn:
            n_acc += float(acc.double().mean())
            plaq.append(gpu_ops.qoi_avg_plaquette(step.theta, Mt, Mx))
            chi.append(gpu_ops.qoi_2d_susceptibility(step.theta, Mt, Mx))
    mp, ep = chain_mean_and_error(torch.stack(plaq))
    mc, ec = chain_mean_and_error(torch.stack(chi))
    x = gpu_ops.lattice_initialise(fine, B, SEED + 3)
    sc = torch.empty_like(x)
    dplaq, dchi = [], []
    for k in range(burn + n):
        gpu_ops.lattice_sweep_draw(fine, x, sc, 1, 1, SEED + 3, 0, 2 * k)
        if k >= burn:
            dplaq.append(gpu_ops.qoi_avg_plaquette(x, Mt, Mx))
            dchi.append(gpu_ops.qoi_2d_susceptibility(x, Mt, Mx))
    rp, erp = chain_mean_and_error(torch.stack(dplaq))
    rc, erc = chain_mean_and_error(torch.stack(dchi))
    print(f"Schwinger two-level ({rt},{rx}): p_accept {n_acc / n:.3f}; plaquette {mp:.5f} +- {ep:.5f} vs direct {rp:.5f} +- {erp:.5f}; "
          f"Q^2/(4 pi^2) {mc:.4f} +- {ec:.4f} vs direct {rc:.4f} +- {erc:.4f}")
    assert n_acc / n > 0.05
    zcheck(_where() + ": mp vs rp", mp, ep, rp, erp)
    zcheck(_where() + ": mc vs rc", mc, ec, rc, erc)


def test_batched_mlmc_harmonic_oscillator_matches_closed_form(gpu_ops):
    """Telescoping estimator on device chains (mlmcpathintegral_amd/mlmc.py, the batched twin of MonteCarloMultiLevel):
    HO, 3 levels M_lat = 128 / 64 / 32, <x^2> against the fine-level closed form."""
    from mlmcpathintegral_amd import abi, mlmc
    import oracle
    est = mlmc.PathMLMC(abi.HARMONIC, 128, 4.0, 3, B=512, nt=50, dt0=0.08, seed=SEED, n_sub=3)
    est.thermalise(300)
    est.pass_(300)
    q, e, table = est.estimate()
    exact = oracle.lib().orc_ho_xsquared_analytical(128, 4.0, 1.0, 1.0)
    print(f"HO MLMC <x^2> = {q:.6f} +- {e:.6f} (closed form {exact:.6f}); level means {table[:, 1].tolist()}, "
          f"variances {table[:, 2].tolist()}; acceptance {est.p_accept()}")
    zcheck(_where() + ": q vs exact", q, e, exact)
    assert table[0, 2] < table[1, 2] < table[2, 2], "variance of Y_l decays towards the fine levels"


def test_batched_mlmc_quartic_five_levels_matches_single_level(gpu_ops):
    """BASELINE config 5 hierarchy (quartic double well, 5 levels, finest M_lat = 32768, coarsest 2048) at a finer
    lattice spacing (T_final = 1024, a = 1/32 ... 1/2) where the two-level steps accept often enough for a short test:
    the telescoping sum of the level estimators against a single-level HMC estimate on the finest lattice.  (At the
    survey's a = 0.125 ... 2 the two-level acceptance of the unrenormalised quartic action drops to a few per cent on
    the coarse levels -- parity for that shape is test_config5_hierarchy_matches_oracle.)"""
    from mlmcpathintegral_amd import abi, mlmc
    par = dict(lam=1.0, x0=1.0)
    M0, T, B = 32768, 1024.0, 64
    est = mlmc.PathMLMC(abi.QUARTIC, M0, T, 5, B=B, nt=50, dt0=0.02, seed=SEED, n_sub=2, params=par)
    est.thermalise(400)
    est.pass_(200)
    q, e, table = est.estimate()
    fine = abi.path_action(abi.QUARTIC, M0, T, 1.0, 1.0, 1.0, 1.0)
    x = gpu_ops.path_initialise(fine, B, SEED + 5)
    hmc = gpu_ops.PathHMC(fine, B, 100, 0.02, seed=SEED + 5)   # long trajectories: the slow modes have period ~ 2 pi
    gpu_ops.hmc_thermalise(hmc, x, 400)
    vals = []
    for k in range(300):
        hmc.draw(x)
        vals.append(gpu_ops.qoi_xsquared(x))
    m, em = chain_mean_and_error(torch.stack(vals))
    p_fine = float(hmc.n_accepted.double().mean()) / hmc.n_total
    print(f"quartic 5-level MLMC <x^2> = {q:.6f} +- {e:.6f}; single-level fine HMC {m:.6f} +- {em:.6f} (p_accept {p_fine:.2f}); "
          f"level means {table[:, 1].tolist()}; acceptance {est.p_accept()}")
    assert p_fine > 0.3
    zcheck(_where() + ": q vs m", q, e, m, em)


def test_hierarchical_mlmc_matches_single_level(gpu_ops):
    """MonteCarloMultiLevel with sampler = 'hierarchical', as the reference composes it (montecarlomultilevel.cc:118-146,
    170-190; sampler/hierarchicalsampler.cc:55-81): HMC only on the coarsest level, masked two-level steps up (a chain
    rejected on a coarse level does not move on the finer ones), coarse samplers sub-sampled ceil(2 tau_int) draws
    apart.  Quartic double well, 3 levels 512 / 256 / 128 sites at a = 1/32 ... 1/8; telescoping sum against a
    single-level HMC estimate on the finest lattice."""
    from mlmcpathintegral_amd import abi, mlmc
    par = dict(lam=1.0, x0=1.0)
    M0, T, B = 512, 16.0, 256
    est = mlmc.PathMLMC(abi.QUARTIC, M0, T, 3, B=B, nt=40, seed=SEED, params=par, hierarchical=True, dt_coarse=0.12)
    est.thermalise(200)
    est.pass_(60)
    q, e, table = est.estimate()
    hier = {l: lv.sampler.p_accept() for l, lv in est.levels.items()}
    fine = abi.path_action(abi.QUARTIC, M0, T, 1.0, 1.0, 1.0, 1.0)
    x = gpu_ops.path_initialise(fine, B, SEED + 5)
    hmc = gpu_ops.PathHMC(fine, B, 100, 0.03, seed=SEED + 5)
    gpu_ops.hmc_thermalise(hmc, x, 300)
    vals = []
    for k in range(200):
        hmc.draw(x)
        vals.append(gpu_ops.qoi_xsquared(x))
    m, em = chain_mean_and_error(torch.stack(vals))
    print(f"hierarchical 3-level MLMC <x^2> = {q:.6f} +- {e:.6f}; single-level fine HMC {m:.6f} +- {em:.6f}; level means "
          f"{table[:, 1].tolist()}; sub-sampling {[lv.n_sub for lv in est.levels.values()]}; acceptance of the hierarchical samplers {hier}; "
          f"two-level steps {est.p_accept()}")
    # every level of every hierarchical sampler moves, and the break is honoured: acceptance can only drop towards the fine levels
    for l, acc in hier.items():
        ks = sorted(acc, reverse=True)
        assert all(acc[k] > 0.05 for k in ks) and all(acc[ks[i]] >= acc[ks[i + 1]] for i in range(len(ks) - 1)), (l, acc)
    zcheck(_where() + ": q vs m", q, e, m, em)


def test_ho_exact_sampler_covariance(gpu_ops):
    """The exact sampler draws independent paths from N(0, Q^-1): <x^2> against the closed form and the two-point
    function <x_0 x_k> against the first column of L L^T, from 8192 chains x 8 draws."""
    from mlmcpathintegral_amd import abi
    import oracle
    M, T, B = 128, 4.0, 8192
    act = abi.path_action(abi.HARMONIC, M, T, 1.0, 1.0)
    s = gpu_ops.HOExactSampler(act, B, seed=SEED)
    cov = s.LT_host.T @ s.LT_host          # L L^T
    x2, c0k = [], []
    for _ in range(8):
        x = s.draw()
        x2.append(gpu_ops.qoi_xsquared(x))
        c0k.append((x[:, :1] * x).mean(dim=0))
    x2 = torch.stack(x2).reshape(-1)
    m, e = float(x2.mean()), float(x2.std(unbiased=True)) / math.sqrt(x2.numel())
    exact = oracle.lib().orc_ho_xsquared_analytical(M, T, 1.0, 1.0)
    print(f"exact sampler <x^2> = {m:.6f} +- {e:.6f} (closed form {exact:.6f})")
    zcheck(_where() + ": m vs exact", m, e, exact)
    c = torch.stack(c0k).mean(dim=0).cpu().numpy()
    err = 4 * math.sqrt(2.0) * cov[0, 0] / math.sqrt(8 * B)   # generous bound on the sampling error of a covariance entry
    assert np.max(np.abs(c - cov[0])) < err


@pytest.mark.parametrize("M,mass,B", [(16, 10.0, 4096), (64, 10.0, 256), (512, 10.0, 16)])
def test_gff_exact_sampler_and_sweeps_share_the_distribution(gpu_ops, M, mass, B):
    """<phi^2> of exact draws against gff_phi_squared_analytical (the reference's closed form, restated in the oracle),
    up to the BASELINE size 512^2 where the reference's own exact sampler cannot be built -- and the
    overrelaxed heat-bath sweeps must leave that distribution invariant: <phi^2> after 3 x (2 OR + 1 HB) sweeps on the
    exact draws agrees as well (a check of the sweep kernels at 512^2 that needs no thermalisation)."""
    from mlmcpathintegral_amd import abi
    import oracle
    act = abi.lattice_action(abi.GFF, M, M, mass=mass)
    s = gpu_ops.GFFExactSampler(act, B, seed=SEED)
    exact = oracle.lib().orc_gff_phi_squared_analytical(mass, M, M)
    draws = torch.stack([gpu_ops.qoi_phi_squared(s.draw()) for _ in range(8)]).reshape(-1)
    m, e = float(draws.mean()), float(draws.std(unbiased=True)) / math.sqrt(draws.numel())
    phi = s.draw()
    scratch = torch.empty_like(phi)
    gpu_ops.lattice_sweep_draw(act, phi, scratch, 2, 1, SEED, 0, 0)
    gpu_ops.lattice_sweep_draw(act, phi, scratch, 2, 1, SEED, 0, 3)
    gpu_ops.lattice_sweep_draw(act, phi, scratch, 2, 1, SEED, 0, 6)
    q = gpu_ops.qoi_phi_squared(phi)
    ms, es = float(q.mean()), float(q.std(unbiased=True)) / math.sqrt(q.numel())
    print(f"GFF {M}^2: <phi^2> exact draws {m:.6f} +- {e:.6f}, after sweeps {ms:.6f} +- {es:.6f}, closed form {exact:.6f}")
    zcheck(_where() + ": m vs exact", m, e, exact)
    zcheck(_where() + ": ms vs exact", ms, es, exact)


def test_schwinger_headline_1024_plaquette_matches_closed_form(gpu_ops):
    """The headline shape itself (BASELINE configs[3]): quenched Schwinger 1024 x 1024, beta = 1, the default sampler
    (10 overrelaxation + 1 heat-bath sweep per draw), 64 chains x 1500 draws after 60 burn-in draws (1.3 10^17 link
    updates' worth of plaquettes: standard error 2 10^-6, from 64 independent chain means): <cos theta_P> against
    I1(1)/I0(1), at the north star's 2 sigma; Q^2/(4 pi^2) against V chi_t(beta = 1, P = 1024^2) of the reference's
    closed form (42610.18) at the session gate."""
    from mlmcpathintegral_amd import abi
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    B, burn, n = 64, 60, 1500
    x = gpu_ops.lattice_initialise(act, B, 99)
    s = torch.empty_like(x)
    plaq, q2, sweep = [], [], 0
    for k in range(burn + n):
        x, s = gpu_ops.lattice_sweep_draw_pingpong(act, x, s, 10, 1, 99, 0, sweep)
        sweep += 11
        if k >= burn:
            plaq.append(gpu_ops.qoi_avg_plaquette(x, 1024, 1024))
            q2.append(gpu_ops.qoi_2d_susceptibility(x, 1024, 1024))
    mp, ep = chain_mean_and_error(torch.stack(plaq))
    mq, eq = chain_mean_and_error(torch.stack(q2))
    zcheck("headline 1024^2: <plaquette> vs I1(1)/I0(1)", mp, ep, 0.44638996, gate=HEADLINE_SIGMA)
    zcheck("headline 1024^2: Q^2/(4 pi^2) vs V chi_t closed form", mq, eq, 42610.18)


def test_schwinger_headline_sampler_matches_cpu_chain(gpu_ops, orc):
    """QoI means within 2 sigma of the CPU chain (north star): the default sampler (10 OR + 1 HB per draw, beta = 1) on
    a 64 x 64 lattice -- a size the reference-order CPU chain (lexicographic sweeps, mt19937_64, Gaussian-envelope
    rejection sampler: the reference's own algorithm in the oracle) affords in seconds -- device multicolour chains
    against that chain: average plaquette and Q^2/(4 pi^2)."""
    from mlmcpathintegral_amd import abi
    Mt = 64
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mt, beta=1.0)
    B, burn, n = 128, 50, 300
    x = gpu_ops.lattice_initialise(act, B, SEED + 64)
    s = torch.empty_like(x)
    plaq, q2, sweep = [], [], 0
    for k in range(burn + n):
        x, s = gpu_ops.lattice_sweep_draw_pingpong(act, x, s, 10, 1, SEED + 64, 0, sweep)
        sweep += 11
        if k >= burn:
            plaq.append(gpu_ops.qoi_avg_plaquette(x, Mt, Mt))
            q2.append(gpu_ops.qoi_2d_susceptibility(x, Mt, Mt))
    mp, ep = chain_mean_and_error(torch.stack(plaq))
    mq, eq = chain_mean_and_error(torch.stack(q2))
    L = orc.lib()
    A = orc.Action(orc.SCHWINGER, Mt=Mt, Mx=Mt, beta=1.0)
    hb = L.orc_heatbath_new(A.h, 1, 10, 100, 0)   # n_sweep_heatbath = 1, n_sweep_overrelax = 10, 100 burn-in draws
    y = np.zeros(A.size)
    rp, rq = [], []
    for _ in range(2000):
        L.orc_heatbath_draw(hb, y)
        rp.append(L.orc_qoi_avg_plaquette(y, Mt, Mt))
        rq.append(L.orc_qoi_2d_susceptibility(y, Mt, Mt))
    L.orc_heatbath_free(hb)
    rp, rq = np.array(rp), np.array(rq)

    def batch_err(v):
        bm = v.reshape(40, -1).mean(axis=1)
        return bm.std(ddof=1) / math.sqrt(40)
    zcheck("headline sampler 64^2: <plaquette> GPU chain vs CPU chain", mp, ep, rp.mean(), batch_err(rp), gate=HEADLINE_SIGMA)
    zcheck("headline sampler 64^2: Q^2/(4 pi^2) GPU chain vs CPU chain", mq, eq, rq.mean(), batch_err(rq), gate=HEADLINE_SIGMA)
    zcheck("headline sampler 64^2: <plaquette> GPU chain vs I1(1)/I0(1)", mp, ep, 0.44638996)


def test_windowed_statistics_on_the_device_equal_the_oracle(gpu_ops, orc):
    """mlmcpi_stats_window_record (Statistics::record_sample with its autocorrelation window, one chain per thread) against
    the oracle's Statistics restatement (itself bit for bit the compiled reference, tests/test_abi_host.py): average,
    variance and tau_int per chain after 7, 20 and 150 samples of correlated series, window 20 and 5."""
    L = orc.lib()
    rng = np.random.default_rng(5)
    for window in (20, 5):
        B, n = 6, 150
        series = np.zeros((n, B))
        for b in range(B):   # AR(1) with different correlations
            rho, v = 0.15 * b, 0.0
            for j in range(n):
                v = rho * v + rng.normal()
                series[j, b] = 0.3 + v
        state = gpu_ops.stats_window_state(B, window)
        handles = [L.orc_stats_new(window) for _ in range(B)]
        for j in range(n):
            gpu_ops.stats_window_record(state, torch.tensor(series[j], dtype=torch.float64, device="cuda"))
            for b in range(B):
                L.orc_stats_record(handles[b], np.ascontiguousarray(series[j, b:b + 1]), 1)
            if j + 1 in (7, 20, 150):
                st = state.cpu().numpy()
                tau = gpu_ops.stats_window_tau_int(state, pooled=False).cpu().numpy()
                for b in range(B):
                    out = np.zeros(6)
                    L.orc_stats_get(handles[b], out)   # average, variance, variance_error, tau_int, error, samples
                    nn, a1, S0 = st[b, 0], st[b, 1], st[b, 2]
                    assert nn == j + 1 == out[5]
                    assert abs(a1 - out[0]) < 1e-13
                    assert abs(nn / (nn - 1.0) * (S0 - a1 * a1) - out[1]) < 1e-12
                    assert abs(tau[b] - out[3]) < 1e-10, (window, j, b, tau[b], out[3])
        for h in handles:
            L.orc_stats_free(h)


def test_device_reproduces_the_bias_of_the_reference_scheme_at_low_hierarchical_acceptance(gpu_ops):
    """VERDICT r04 item 4 / ADVICE r04 (medium).  The multilevel estimator of the reference feeds its two-level steps with
    coarse samples taken ceil(2 tau_int) draws of a HierarchicalSampler apart (montecarlomultilevel.cc:170-190).  When
    that sampler accepts rarely (12 % here) its sojourn times are long-tailed, consecutive coarse samples are not
    independent, and the two-level chain -- exact for independent proposals -- is BIASED: the reference-order restatement
    in the oracle (MlmcRefO, mt19937_64, sequential; tools/exp_hier_bias.py, profiles/r05_hier_bias_reference_order_level0.json),
    quartic double well, 3 levels 1024 / 512 / 256 sites, T = 256, 8 x 150000 samples, finds for level 0
        fine part  <x^2> = 0.585904 +- 0.000189   against single-level HMC 0.591391 +- 0.000054   (-27.9 sigma)
        coarse part        0.568512 +- 0.000115   against HMC on 512 sites 0.568467 +- 0.000041   (+0.4 sigma: the sampler is exact)
        Y_0              = 0.017392 +- 0.000102   against 0.022924 expected                        (-45 sigma).
    Here: the same level on the device (Philox streams, multicolour order of nothing -- 1-D paths --, running tau_int pooled
    over the batch) lands on the ORACLE's biased value, not on the unbiased one: the device composition is the reference's."""
    from mlmcpathintegral_amd import abi, mlmc
    par = dict(lam=1.0, x0=1.0)
    M0, T, B = 1024, 256.0, 384
    est = mlmc.PathMLMC(abi.QUARTIC, M0, T, 3, B=B, nt=100, dt0=0.05, seed=SEED + 11, params=par, hierarchical=True, dt_coarse=0.1)
    lv = est.levels[0]
    lv.thermalise(300, est.dts)
    fine_part, coarse_part = [], []
    n_samples = 260
    for _ in range(n_samples):
        y = lv.sample()
        fine_part.append(gpu_ops.qoi_xsquared(lv.step.theta).clone())
        coarse_part.append(fine_part[-1] - y)
    f, ef = chain_mean_and_error(torch.stack(fine_part[20:]))
    c, ec = chain_mean_and_error(torch.stack(coarse_part[20:]))
    hier_acc = lv.sampler.p_accept()
    two_acc = float(lv.step_accepted.double().mean()) / max(1, lv.n_draws)
    print(f"device level 0: fine part {f:.6f} +- {ef:.6f}, coarse part {c:.6f} +- {ec:.6f}, draws between coarse samples "
          f"{lv.n_sub_sum / n_samples:.1f}, hierarchical acceptance {hier_acc}, two-level acceptance {two_acc:.3f}")
    assert 0.05 < min(hier_acc.values()) < 0.25 and 0.3 < two_acc < 0.6            # the regime of the oracle run (0.126, 0.449)
    assert 30 < lv.n_sub_sum / n_samples < 60                                      # the oracle: 44.2
    zcheck(_where() + ": coarse part vs single-level HMC on 512 sites (the hierarchical sampler is exact)", c, ec, 0.5684667, 0.0000407)
    zcheck(_where() + ": fine part vs the reference-order oracle's (biased) value", f, ef, 0.5859039, 0.0001892)
    z_unbiased = (f - 0.5913911) / math.hypot(ef, 0.0000538)
    print(f"[z] fine part vs single-level HMC on 1024 sites: z = {z_unbiased:+.2f} (the reference scheme's bias: -27.9 sigma in the oracle run)")
    assert z_unbiased < -4.0, "the two-level chain fed by the sub-sampled hierarchical sampler should show the scheme's bias"
