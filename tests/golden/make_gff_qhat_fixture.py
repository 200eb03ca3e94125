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
    with open(os.path.join(HERE, "gff_qhat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote gff_qhat.json:", len(out["qhat"]), "Q-hat cases;", acc)


if __name__ == "__main__":
    main()
