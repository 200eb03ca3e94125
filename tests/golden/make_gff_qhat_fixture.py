#!/usr/bin/env python3
"""Generates tests/golden/gff_qhat.json from the reference author's own numpy construction of the Gibbs-smoothed coarse
GFF action.  Runs in the build container only (it imports /root/reference/python/*.py; nothing of those files is copied:
the fixture holds numbers).

    python tests/golden/make_gff_qhat_fixture.py

What is imported and how:
  * /root/reference/python/gibbs_smoother.py -- class CoarseGibbsSmoother (:1-55): stencils of the rotated level,
    Sigma, Sigma_initial, G = 1 - tril(Q)^-1 Q, Sigma_iter = Sigma + G^n (Sigma_initial - Sigma) G^n^T, Q_prec_iter.
    This is the construction of src/action/qft/gffaction.cc:133-166 (Q_precision_hat) for the first rotated level:
    stencil_initial = {4 + 2 mu2_f, -1} = the level's plain stencil (its mu2 is twice the fine one), stencil = the
    9-point marginal {4 + mu2_f - 4/(4 + mu2_f), -2/(4 + mu2_f), -1/(4 + mu2_f)}.  The file has no `import numpy`:
    numpy is put into the module's namespace as `np` before it is executed.
  * /root/reference/python/gff_twolevel_coarse.py -- class GFFAction (:8-152): draw_5pt / draw_9pt / evaluate_5pt /
    evaluate_9pt, the exact samplers and actions of the plain coarse stencil (4 + 2 mu2) and of the 9-point marginal.
    Importing it runs the author's 64 x 64 demonstration first (about a minute); its TwoLevelSampler.step hard-wires a
    low-mode projection, so the acceptance recorded here is the same Metropolis test WITHOUT the projection, written
    with the class's own draw_* / evaluate_* calls (see independence_acceptance below).

  * /root/reference/python/gff_twolevel.py -- class GFFAction (:7-165): evaluate (the plain fine action), evaluate_fillin
    (the Gaussian fill-in action of the odd vertices = GFFConditionedFineAction::evaluate, gffconditionedfineaction.cc:7-49),
    class TwoLevelSampler.step (:206-229: the three action differences of TwoLevelMetropolisStep::draw), class
    QoISquaredField.exact_value (:186-193 = gff_phi_squared_analytical, auxilliary.cc:197-209).  Importing it runs the
    author's 32 x 32 demonstration and writes covariance.pdf into the working directory: it is imported from a temporary
    directory.  Recorded: S_fine, S_fillin, S_coarse (the smoothed coarse action, Lattice2D sweep order) of two
    deterministic fields and the step's Delta S for the pair (current, proposal).

Index order.  The Gibbs iteration matrix G depends on the ORDER of the sweep, so for nsmooth > 0 Q_prec_iter depends on
the linear numbering of the rotated vertices.  The Python class numbers them (i outer, j inner, i + j even); the C++
Lattice2D numbers them even-even vertices first, then odd-odd (lattice/lattice2d.hh:230-241), and GFFAction sweeps in
that order.  Both are recorded: "order": "python" (the class as it is) and "order": "lattice2d" (a subclass that
replaces ONLY _build_index_maps by the C++ numbering; the matrices are still built by the reference's _build_matrices).
For nsmooth = 0 the result does not depend on the order.

Quantities (all permutation-free: fields and entries are addressed by Cartesian coordinates):
  energies[k] = CoarseGibbsSmoother.evaluate(phi_k) = 1/2 phi_k^T Q_prec_iter phi_k for three deterministic fields
  sigma_iter_diag = [[i, j, Sigma_iter[l, l]], ...], sigma_iter_row0 = [[i, j, Sigma_iter[l(0,0), l(i,j)]], ...]
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/python"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_gibbs_smoother():
    spec = importlib.util.spec_from_file_location("gibbs_smoother", os.path.join(REF, "gibbs_smoother.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.np = np   # the file uses np without importing it
    spec.loader.exec_module(mod)
    sys.modules["gibbs_smoother"] = mod
    return mod


def fields(Mlat):
    """three deterministic fields on the Mlat x Mlat grid (only entries with i + j even are read)"""
    i, j = np.meshgrid(np.arange(Mlat), np.arange(Mlat), indexing="ij")
    return [np.sin(0.7 * i + 1.3 * j + 0.2),
            np.cos(2.0 * np.pi * i / Mlat) * (1.0 + 0.1 * j),
            ((3 * i + 5 * j) % 7) / 7.0 - 0.4]


def lattice2d_rotated_index(Mlat, i, j):
    """Lattice2D::vertex_cart2lin for a rotated lattice (lattice/lattice2d.hh:230-241), restated"""
    half = Mlat // 2
    ish, jsh = ((i + Mlat) - (i & 1)) // 2, ((j + Mlat) - (j & 1)) // 2
    return half * (jsh % half) + ish % half + (Mlat * Mlat // 4) * (i & 1)


def qhat_cases(gs):
    class Lattice2DOrder(gs.CoarseGibbsSmoother):
        def _build_index_maps(self):
            self.cart2lin_idx, self.lin2cart_idx = {}, {}
            for i in range(self.Mlat):
                for j in range(self.Mlat):
                    if (i + j) % 2 == 0:
                        ell = lattice2d_rotated_index(self.Mlat, i, j)
                        self.cart2lin_idx[(i, j)] = ell
                        self.lin2cart_idx[ell] = (i, j)

    cases = []
    mass = 10.0
    for Mlat in (8, 16):
        action = types.SimpleNamespace(Mlat=Mlat, alat=1.0 / Mlat, mass=mass)
        for nsmooth in (0, 1, 2):
            for order, cls in (("python", gs.CoarseGibbsSmoother), ("lattice2d", Lattice2DOrder)):
                sm = cls(action, nsmooth=nsmooth)
                l00 = sm.cart2lin_idx[(0, 0)]
                coords = sorted(sm.cart2lin_idx)
                cases.append({
                    "Mlat": Mlat, "mass": mass, "nsmooth": nsmooth, "order": order,
                    "energies": [float(sm.evaluate(f)) for f in fields(Mlat)],
                    "sigma_iter_diag": [[i, j, float(sm.Sigma_iter[sm.cart2lin_idx[(i, j)], sm.cart2lin_idx[(i, j)]])] for i, j in coords],
                    "sigma_iter_row0": [[i, j, float(sm.Sigma_iter[l00, sm.cart2lin_idx[(i, j)]])] for i, j in coords],
                })
    return cases


def independence_acceptance(tl, Mlat, mass, nsamples, seed):
    """TwoLevelSampler.step of gff_twolevel_coarse.py:155-186 with project = False: state ~ 9-point marginal (the target),
    proposal ~ plain stencil with 2 mu2 (what the coarse level samples), Metropolis test on
    dS = S_9(prop) - S_9(state) + S_5(state) - S_5(prop).  Returns (acceptance rate, <min(1, exp(-dS))>)."""
    np.random.seed(seed)
    action = tl.GFFAction(Mlat, mass)
    state, prop = np.zeros((Mlat, Mlat)), np.zeros((Mlat, Mlat))
    accepted, expected = 0, 0.0
    for _ in range(nsamples):
        action.draw_5pt(prop)
        action.draw_9pt(state)
        dS = action.evaluate_9pt(prop) - action.evaluate_9pt(state) + action.evaluate_5pt(state) - action.evaluate_5pt(prop)
        p = 1.0 if dS < 0 else float(np.exp(-dS))
        expected += p
        accepted += dS < 0 or np.random.uniform() < p
    return accepted / nsamples, expected / nsamples


def twolevel_terms(gs, tw, Lattice2DOrder):
    """The terms of one two-level step between two deterministic fields, by the reference author's own methods."""
    out = []
    for Mlat, mass, nsmooth in ((8, 10.0, 2), (16, 10.0, 2), (16, 3.0, 1), (16, 10.0, 0)):
        action = tw.GFFAction(Mlat, mass)
        sm = Lattice2DOrder(action, nsmooth=nsmooth)
        cur, prop = fields(Mlat)[0], fields(Mlat)[2]
        terms = {}
        for name, phi in (("current", cur), ("proposal", prop)):
            terms[name] = {"S_fine": float(action.evaluate(phi)), "S_fillin": float(action.evaluate_fillin(phi)),
                           "S_coarse": float(sm.evaluate(phi))}
        # TwoLevelSampler.step, gff_twolevel.py:213-218
        dS = (terms["proposal"]["S_fine"] - terms["current"]["S_fine"]) + (terms["current"]["S_coarse"] - terms["proposal"]["S_coarse"]) \
            + (terms["current"]["S_fillin"] - terms["proposal"]["S_fillin"])
        out.append({"Mlat": Mlat, "mass": mass, "nsmooth": nsmooth, "fields": ["fields()[0]", "fields()[2]"], "terms": terms, "DeltaS": dS})
    exact = [{"Mlat": M, "mass": m, "phi_squared": float(tw.QoISquaredField(M, m).exact_value())} for M, m in ((16, 10.0), (64, 10.0), (32, 1.0))]
    return out, exact


def expected_acceptance(cls, Mlat, mass, nsmooth, n, seed):
    """Stationary acceptance probability of the two-level step 16 x 16 <- rotated level, from the reference author's matrices
    (numpy, vectorised restatement of TwoLevelSampler.step: current state ~ the fine action, coarse proposal ~ Sigma_iter,
    Gaussian fill-in of the odd vertices): E[min(1, exp(-Delta S))].  cls fixes the order of the Gibbs sweep."""
    act = types.SimpleNamespace(Mlat=Mlat, alat=1.0 / Mlat, mass=mass)
    mu2, N = (mass / Mlat) ** 2, Mlat * Mlat
    idx = lambda i, j: (i % Mlat) + Mlat * (j % Mlat)
    Q = np.zeros((N, N))
    for i in range(Mlat):
        for j in range(Mlat):
            Q[idx(i, j), idx(i, j)] = 4 + mu2
            for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                Q[idx(i, j), idx(i + di, j + dj)] = -1
    Lf = np.linalg.cholesky(np.linalg.inv(Q))
    sm = cls(act, nsmooth=nsmooth)
    c_lin = np.zeros(sm.ndof, dtype=int)
    for (i, j), l in sm.cart2lin_idx.items():
        c_lin[l] = idx(i, j)
    Lc, Qh, kap = np.linalg.cholesky(sm.Sigma_iter), sm.Q_prec_iter, 4 + mu2
    odd = [(i, j) for i in range(Mlat) for j in range(Mlat) if (i + j) % 2 == 1]
    nbr = np.array([[idx(i + 1, j), idx(i - 1, j), idx(i, j + 1), idx(i, j - 1)] for i, j in odd])
    oddl = np.array([idx(i, j) for i, j in odd])

    def terms(phi):
        pc = phi[c_lin]
        return 0.5 * phi @ Q @ phi, 0.5 * pc @ Qh @ pc, 0.5 * kap * np.sum((phi[oddl] - phi[nbr].sum(1) / kap) ** 2)

    rng = np.random.default_rng(seed)
    acc = []
    for _ in range(n):
        th = Lf @ rng.standard_normal(N)
        pr = np.zeros(N)
        pr[c_lin] = Lc @ rng.standard_normal(sm.ndof)
        pr[oddl] = pr[nbr].sum(1) / kap + rng.standard_normal(len(oddl)) / np.sqrt(kap)
        a, b, c = terms(th)
        d, e, f = terms(pr)
        acc.append(min(1.0, float(np.exp(-((d - a) + (b - e) + (c - f))))))
    return float(np.mean(acc)), float(np.std(acc) / np.sqrt(n))


def main():
    gs = load_gibbs_smoother()
    out = {"generated_by": "tests/golden/make_gff_qhat_fixture.py",
           "source": ["/root/reference/python/gibbs_smoother.py:1-55 (CoarseGibbsSmoother)",
                      "/root/reference/python/gff_twolevel_coarse.py:8-152 (GFFAction)"],
           "qhat": qhat_cases(gs)}
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    import gff_twolevel_coarse as tl   # runs the author's 64 x 64 demonstration on import
    acc = []
    # mu2 of the fine level of the product's two-level test (16 x 16, mass 10): (10 / 16)^2; the Python class takes
    # mu2 = (mass / Mlat)^2, so mass = 0.625 Mlat keeps it
    for Mlat in (8, 12, 16):
        rate, mean_p = independence_acceptance(tl, Mlat, 0.625 * Mlat, 4000, 20260 + Mlat)
        acc.append({"Mlat": Mlat, "ndof": Mlat * Mlat, "mu2_fine": 0.390625, "nsamples": 4000, "accepted_fraction": rate,
                    "mean_acceptance_probability": mean_p})
    out["independence_acceptance_5pt_vs_9pt"] = acc
    # gff_twolevel.py: imported from a scratch directory (it saves a plot where it runs)
    import tempfile

    class Lattice2DOrder(gs.CoarseGibbsSmoother):
        def _build_index_maps(self):
            self.cart2lin_idx, self.lin2cart_idx = {}, {}
            for i in range(self.Mlat):
                for j in range(self.Mlat):
                    if (i + j) % 2 == 0:
                        ell = lattice2d_rotated_index(self.Mlat, i, j)
                        self.cart2lin_idx[(i, j)] = ell
                        self.lin2cart_idx[ell] = (i, j)

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            import gff_twolevel as tw   # runs the author's 32 x 32 demonstration on import
        finally:
            os.chdir(cwd)
    out["source"].append("/root/reference/python/gff_twolevel.py:7-229 (GFFAction, QoISquaredField, TwoLevelSampler.step)")
    out["twolevel_terms"], out["phi_squared_exact"] = twolevel_terms(gs, tw, Lattice2DOrder)
    # the author's TwoLevelSampler as it is (gff_twolevel.py:195-232), at the shape of the product's two-level chain test
    # (16 x 16 <- rotated 16 x 16, mass 10, two Gibbs sweeps): its acceptance rate, for comparison with the device chain's
    rho = []
    for Mlat, mass, nsmooth, n in ((16, 10.0, 2, 4000), (16, 10.0, 0, 4000)):
        np.random.seed(4242 + nsmooth)
        sampler = tw.TwoLevelSampler(tw.GFFAction(Mlat, mass), tw.QoISquaredField(Mlat, mass), nsmooth=nsmooth)
        for _ in range(n):
            sampler.step()
        rho.append({"Mlat": Mlat, "mass": mass, "nsmooth": nsmooth, "nsamples": n, "rho_accept": sampler.rho_accept()})
    out["twolevel_rho_accept"] = rho
    out["twolevel_expected_acceptance"] = []
    for order, cls in (("python", gs.CoarseGibbsSmoother), ("lattice2d", Lattice2DOrder)):
        m, e = expected_acceptance(cls, 16, 10.0, 2, 20000, 1)
        out["twolevel_expected_acceptance"].append({"Mlat": 16, "mass": 10.0, "nsmooth": 2, "order": order, "mean": m, "error": e, "n": 20000})
    with open(os.path.join(HERE, "gff_qhat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote gff_qhat.json:", len(out["qhat"]), "Q-hat cases;", acc, out["twolevel_rho_accept"])


if __name__ == "__main__":
    main()
