"""CPU: the oracle restatement against the known answers recorded from the compiled reference
(SURVEY.md 8(c) -> tests/golden/survey_known_answers.json) and published Philox vectors."""
import json
import math
import os

import numpy as np
import pytest


def seq(n):
    return np.sin(np.arange(n) + 1.0)


def test_philox_known_answers(orc):
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "philox_kat.json")))
    for v in kat["vectors"]:
        ctr = np.array([int(x, 16) for x in v["ctr"]], dtype=np.uint32)
        key = np.array([int(x, 16) for x in v["key"]], dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        orc.lib().orc_philox4x32_10(ctr, key, out)
        assert [f"{x:08x}" for x in out] == v["out"]


def test_rotor_known_answers(orc, golden):
    g = golden["rotor_M16"]
    A = orc.Action(orc.ROTOR, **g["params"])
    x = seq(16)
    assert A.evaluate(x) == g["S"]
    assert list(A.force(x)[:4]) == g["force_0_3"]
    A.overrelaxation_update(x, 0)
    A.overrelaxation_update(x, 15)
    assert (x[0], x[15]) == (g["after_overrelax_0_then_15"]["x0"], g["after_overrelax_0_then_15"]["x15"])
    # heat bath with the action's own engine (seed 21172817), bit-exact incl. libstdc++ distributions
    A.heatbath_update(x, 3)
    A.heatbath_update(x, 4)
    assert (x[3], x[4]) == (g["after_heatbath_3_then_4"]["x3"], g["after_heatbath_3_then_4"]["x4"])


def test_quartic_known_answers(orc, golden):
    g = golden["quartic_M16"]
    A = orc.Action(orc.QUARTIC, **g["params"])
    x = seq(16)
    L = orc.lib()
    assert A.evaluate(x) == g["S"]
    assert L.orc_qoi_xsquared(x, 16) == g["X2"]
    assert list(A.force(x)[:4]) == g["force_0_3"]
    assert L.orc_action_wminimum(A.h, 0.3, 0.7) == g["Wminimum_0.3_0.7"]
    assert L.orc_action_wcurvature(A.h, 0.3, 0.7) == g["Wcurvature_0.3_0.7"]


def test_schwinger_known_answers(orc, golden):
    g = golden["schwinger_4x4"]
    A = orc.Action(orc.SCHWINGER, **g["params"])
    x = seq(32)
    assert A.evaluate(x) == g["S"]
    assert orc.lib().orc_qoi_avg_plaquette(x, 4, 4) == g["plaq"]
    assert list(A.force(x)[:4]) == g["force_0_3"]
    A.overrelaxation_update(x, 0)
    A.overrelaxation_update(x, 1)
    assert list(x[:2]) == g["after_overrelax_0_then_1"]
    A.heatbath_update(x, 2)
    A.heatbath_update(x, 3)
    assert list(x[2:4]) == g["after_heatbath_2_then_3"]


def test_schwinger_copy_from_fine_known_answers(orc, golden):
    """coarse_action()->copy_from_fine on the 4 x 4 state after the recorded updates; the coarse
    links are mod_2pi sums of the fine links they span (quenchedschwingeraction.cc:147-162)."""
    g = golden["schwinger_4x4"]
    x = seq(32)
    x[0:2] = g["after_overrelax_0_then_1"]
    x[2:4] = g["after_heatbath_2_then_3"]
    coarse = np.zeros(8)
    orc.lib().orc_schwinger_copy_from_fine(2, 2, 2, 2, x, coarse)
    assert np.max(np.abs(coarse[:4] - np.array(g["copy_from_fine_first_four_coarse_links"]))) < 1e-15


def test_gff_known_answers(orc, golden):
    g = golden["gff_4x4"]
    A = orc.Action(orc.GFF, **g["params"])
    x = seq(16)
    L = orc.lib()
    assert L.orc_action_gff_mu2(A.h) == g["mu2"]
    assert A.evaluate(x) == g["S"]
    assert L.orc_qoi_2d_phi_squared(x, 16) == g["phi2"]
    assert list(A.force(x)[:4]) == g["force_0_3"]
    A.overrelaxation_update(x, 5)
    assert x[5] == g["after_overrelax_5"]
    A.heatbath_update(x, 6)
    A.heatbath_update(x, 7)
    assert list(x[6:8]) == g["after_heatbath_6_then_7"]


def test_seeded_initial_states(orc, golden):
    g = golden["seeded_initial_states"]
    L = orc.lib()
    A = orc.Action(orc.ROTOR, M=65536, T_final=4.0, m0=0.25)
    x = A.initialise_state()
    e = g["rotor_M65536_T4_m0_0.25"]
    assert (A.evaluate(x), x[0], A.force(x)[0]) == (e["S"], e["x0"], e["force0"])
    assert L.orc_qoi_susceptibility(x, 65536, 4.0) == e["chi"]
    A = orc.Action(orc.SCHWINGER, Mt=256, Mx=256, beta=1.0)
    x = A.initialise_state()
    e = g["schwinger_256_beta1"]
    assert (A.evaluate(x), L.orc_qoi_2d_susceptibility(x, 256, 256)) == (e["S"], e["Q2"])


def test_lattice_known_answers(orc, golden):
    g = golden["lattice2d_4x4"]
    nb = np.zeros(16 * 8, dtype=np.uint32)
    orc.lib().orc_neighbours2d(4, 4, 0, nb)
    assert list(nb[:8]) == g["neighbours_of_vertex_0"]
    assert orc.lib().orc_link_cart2lin(4, 4, -1, -1, 1) == g["link_cart2lin_m1_m1_1"]
    nbr = np.zeros(8 * 8, dtype=np.uint32)
    orc.lib().orc_neighbours2d(4, 4, 1, nbr)
    assert list(nbr[:8]) == g["rotated_level1_of_CoarsenRotate"]["neighbours_of_vertex_0"]


def test_hmc_autotune_matches_reference(orc, golden):
    """sampler/hmcsampler.cc:77-113 with the reference's seeds: the tuned step size is a chain-exact
    quantity (100 x 1000 trajectories through mt19937_64 + libstdc++ normal/uniform)."""
    g = golden["hmc_quartic_M128"]
    p = g["params"]
    L = orc.lib()
    A = orc.Action(orc.QUARTIC, M=p["M"], T_final=p["T_final"], m0=p["m0"], mu2=p["mu2"], lam=p["lam"], x0=p["x0"])
    h = L.orc_hmc_new(A.h, p["nt"], p["dt0"], p["n_rep"], p["n_burnin"], 1, 100, 1000)
    assert L.orc_hmc_tuned(h) == 1
    assert round(L.orc_hmc_dt(h), 4) == g["tuned_dt_4dp"]
    x = np.zeros(p["M"])
    for _ in range(5000):
        L.orc_hmc_draw(h, x)
    assert abs(L.orc_hmc_p_accept(h) - g["p_accept_approx"]) < 0.02
    L.orc_hmc_free(h)


def test_analytic_expectation_values(orc, golden):
    L = orc.lib()
    r = golden["reference_runs"]
    assert abs(L.orc_ho_xsquared_analytical(128, 4.0, 1.0, 1.0) - r["config1_ho_M128"]["analytic_x2"]) < 5e-7
    for n, v in r["gff_phi2_analytic"].items():
        assert abs(L.orc_gff_phi_squared_analytical(10.0, int(n), int(n)) - v) < 5e-9


def test_device_order_sweep_preserves_action(orc):
    """Overrelaxation leaves the action invariant (each update reflects about the conditional
    mode); holds for the multicolour order exactly as for the reference order."""
    rng = np.random.default_rng(5)
    for A in (orc.Action(orc.SCHWINGER, Mt=8, Mx=6, beta=1.3), orc.Action(orc.GFF, Mt=8, Mx=8, mass=3.0),
              orc.Action(orc.ROTOR, M=32, T_final=4.0, m0=0.25)):
        x = rng.uniform(-3, 3, A.size)
        S0 = A.evaluate(x)
        A.dev_sweep(x, False, 1, 0, 0)
        assert abs(A.evaluate(x) - S0) < 1e-10 * max(1.0, abs(S0))


@pytest.mark.parametrize("kind", ["schwinger", "gff", "rotor"])
def test_reference_and_device_order_agree_statistically(orc, kind):
    """The multicolour/Philox chain and the reference-order (lexicographic, mt19937_64) chain sample
    the same distribution: means of a QoI agree within 4 combined standard errors."""
    L = orc.lib()
    if kind == "schwinger":
        A = orc.Action(orc.SCHWINGER, Mt=8, Mx=8, beta=1.0)
        q = lambda x: L.orc_qoi_avg_plaquette(x, 8, 8)
    elif kind == "gff":
        A = orc.Action(orc.GFF, Mt=8, Mx=8, mass=4.0)
        q = lambda x: L.orc_qoi_2d_phi_squared(x, 64)
    else:
        A = orc.Action(orc.ROTOR, M=16, T_final=2.0, m0=1.0)
        q = lambda x: np.cos(x[1] - x[0])
    n = 4000
    hb = L.orc_heatbath_new(A.h, 1, 1, 200, 0)
    x = np.zeros(A.size)
    ref = []
    for _ in range(n):
        L.orc_heatbath_draw(hb, x)
        ref.append(q(x))
    L.orc_heatbath_free(hb)
    y = A.dev_initialise(3, 0)
    dev = []
    step = 0
    for k in range(n + 200):
        A.dev_sweep(y, False, 3, 0, step)
        A.dev_sweep(y, True, 3, 0, step + 1)
        step += 2
        if k >= 200:
            dev.append(q(y))
    ref, dev = np.array(ref), np.array(dev)
    # sweeps decorrelate quickly at these couplings; inflate the naive error by a safe factor 2
    err = 2.0 * np.sqrt(ref.var() / n + dev.var() / n)
    assert abs(ref.mean() - dev.mean()) < 4 * err, (ref.mean(), dev.mean(), err)


def test_golden_vectors_are_the_surveys_numbers():
    """tests/golden/check_provenance.py: every value of survey_known_answers.json occurs in SURVEY.md section 8(c),
    where the survey recorded it from the compiled reference."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(__file__), "golden", "check_provenance.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:]


def test_rotor_closed_form_equals_reference_formula(orc):
    """The device-order rotor update uses getWminimum in closed form (mean angle, shifted by pi when
    cos((x+ - x-)/2) < 0) instead of atan2(sin x+ + sin x-, cos x+ + cos x-) (rotoraction.hh:206-213): both forms
    must give the same overrelaxation and heat-bath updates, up to the conditioning of the atan2 form."""
    L = orc.lib()
    M, T, m0 = 64, 8.0, 0.25
    A = orc.Action(orc.ROTOR, M=M, T_final=T, m0=m0)
    rng = np.random.default_rng(17)
    wrap = lambda v: v - 2 * np.pi * np.floor((v + np.pi) / (2 * np.pi))
    for trial in range(20):
        x = rng.uniform(-np.pi, np.pi, M)
        # overrelaxation, even sites then odd sites: reference formula site by site vs the device-order sweep
        ref = x.copy()
        for colour in (0, 1):
            for l in range(colour, M, 2):
                A.overrelaxation_update(ref, l)
        dev = x.copy()
        A.dev_sweep(dev, False, 5, 0, trial)
        d = wrap(dev - ref)
        cond = 1.0 / np.maximum(1e-6, np.abs(np.cos(0.5 * (np.roll(x, -1) - np.roll(x, 1)))))
        assert np.all(np.abs(d) < 1e-14 * cond + 1e-13), np.max(np.abs(d))
        # heat bath: atan2 centre + the same von Mises stream vs the device-order sweep
        ref = x.copy()
        a = T / M
        for colour in (0, 1):
            for l in range(colour, M, 2):
                xm, xp = ref[(l - 1) % M], ref[(l + 1) % M]
                x_min = math.atan2(math.sin(xp) + math.sin(xm), math.cos(xp) + math.cos(xm))
                sigma = 2.0 * (2.0 * m0 / a) * abs(math.cos(0.5 * (xp - xm)))
                # 2 m0 / a = 4 <= kVsKappaMax: the sweeps of this action draw from the step envelope (kappa = sigma / 2)
                ref[l] = L.orc_dev_vs_draw(5, 0, trial, l, 2.0 * m0 / a, xp, xm)
        dev = x.copy()
        A.dev_sweep(dev, True, 5, 0, trial)
        assert np.max(np.abs(wrap(dev - ref))) < 1e-9
