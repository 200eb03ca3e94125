"""Gaussian free field multilevel glue: GFFAction on the levels of a coarsening hierarchy (rotated levels,
Gibbs-smoothed coarse actions), GFFConditionedFineAction and the two-level step between GFF levels.
CPU: vertex lists against the compiled reference (oracle/_ref), the library's dense matrices against the oracle's
independent construction and against the defining properties (n_gibbs = 0 gives the plain precision matrix, many sweeps
give the exact marginal of the finer level).  GPU: evaluate / draw / fill-in / two-level step against the oracle, and the
hierarchical chain against the closed form of <phi^2>."""
import ctypes as C
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROTATE, BOTH = 4, 0


class Level:
    """library-side level handle (host functions work without a GPU)"""

    def __init__(self, Mt, ctype, level, mass, n_gibbs=0, omega=1.0):
        from mlmcpathintegral_amd import abi
        self.abi = abi
        self.h = C.c_void_p()
        abi.call("mlmcpi_gff_level_create", Mt, Mt, ctype, level, float(mass), n_gibbs, float(omega), C.byref(self.h))
        n, nc, mtc, mxc, mu2 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
        abi.call("mlmcpi_gff_level_info", self.h, C.byref(n), C.byref(nc), C.byref(mtc), C.byref(mxc), C.byref(mu2))
        self.N, self.n_coarse, self.Mt_c, self.mu2 = n.value, nc.value, mtc.value, mu2.value

    def tables(self):
        pairs = np.zeros(2 * self.n_coarse, dtype=np.uint32)
        fineonly = np.zeros(self.N - self.n_coarse, dtype=np.uint32)
        self.abi.call("mlmcpi_gff_level_tables", self.h, pairs.ctypes.data_as(C.c_void_p), fineonly.ctypes.data_as(C.c_void_p))
        return pairs, fineonly

    def matrix(self, which):
        out = np.zeros((self.N, self.N))
        self.abi.call("mlmcpi_gff_level_matrix", self.h, which, out.ctypes.data_as(C.c_void_p))
        return out

    def __del__(self):
        try:
            self.abi.call("mlmcpi_gff_level_destroy", self.h)
        except Exception:
            pass


class OLevel:
    def __init__(self, orc, Mt, ctype, level, mass, n_gibbs=0, omega=1.0):
        self.L = orc.lib()
        self.h = self.L.orc_gff_level_new(Mt, Mt, ctype, level, float(mass), n_gibbs, float(omega))
        self.N, self.n_coarse = self.L.orc_gff_level_size(self.h), self.L.orc_gff_level_n_coarse(self.h)

    def tables(self):
        pairs = np.zeros(2 * self.n_coarse, dtype=np.uint32)
        fineonly = np.zeros(self.N - self.n_coarse, dtype=np.uint32)
        self.L.orc_gff_level_tables(self.h, pairs.ctypes.data_as(C.c_void_p), fineonly.ctypes.data_as(C.c_void_p))
        return pairs, fineonly

    def matrix(self, which):
        out = np.zeros((self.N, self.N))
        self.L.orc_gff_level_matrix(self.h, which, out)
        return out

    def __del__(self):
        self.L.orc_gff_level_free(self.h)


@pytest.fixture(scope="module")
def ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    lib = C.CDLL(path)
    lib.ref_lattice2d_new.restype = C.c_void_p
    lib.ref_lattice2d_new.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_int]
    for f in ("ref_lattice2d_free", "ref_lattice2d_fineonly", "ref_lattice2d_fine2coarse"):
        getattr(lib, f).restype = None
    lib.ref_lattice2d_free.argtypes = [C.c_void_p]
    lib.ref_lattice2d_fineonly.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_lattice2d_fine2coarse.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_lattice2d_n_fineonly.argtypes = [C.c_void_p]
    lib.ref_lattice2d_n_coarse.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("Mt,ctype,level", [(8, ROTATE, 0), (8, ROTATE, 1), (16, ROTATE, 0), (16, ROTATE, 1), (8, BOTH, 0), (4, ROTATE, 0),
                                            (32, ROTATE, 1)])
def test_vertex_lists_match_compiled_reference(ref, orc, Mt, ctype, level):
    """coarse / fine-only vertices and the fine -> coarse map of Lattice2D (lattice2d.cc:82-134), rotated levels included:
    library == oracle == the reference's own class, bit for bit"""
    h = ref.ref_lattice2d_new(Mt, Mt, ctype, level)
    nf, nc = ref.ref_lattice2d_n_fineonly(h), ref.ref_lattice2d_n_coarse(h)
    want_f, want_p = np.zeros(nf, dtype=np.uint32), np.zeros(2 * nc, dtype=np.uint32)
    ref.ref_lattice2d_fineonly(h, want_f.ctypes.data_as(C.c_void_p))
    ref.ref_lattice2d_fine2coarse(h, want_p.ctypes.data_as(C.c_void_p))
    ref.ref_lattice2d_free(h)
    for lv in (Level(Mt, ctype, level, 1.0), OLevel(orc, Mt, ctype, level, 1.0)):
        assert lv.n_coarse == nc and lv.N - lv.n_coarse == nf
        pairs, fineonly = lv.tables()
        assert np.array_equal(pairs, want_p) and np.array_equal(fineonly, want_f)


@pytest.mark.parametrize("Mt,level,n_gibbs,omega", [(4, 0, 2, 1.0), (4, 1, 2, 1.0), (8, 1, 2, 1.0), (8, 1, 1, 1.3), (8, 0, 3, 0.8), (8, 1, 0, 1.0)])
def test_dense_matrices_match_oracle(orc, Mt, level, n_gibbs, omega):
    """Qhat of gffaction.cc:126-173 and the factor of the exact sampler: the library (Cholesky-based inverses) against the
    oracle (Gauss-Jordan), independent constructions of the same dense algebra"""
    a, b = Level(Mt, ROTATE, level, 10.0, n_gibbs, omega), OLevel(orc, Mt, ROTATE, level, 10.0, n_gibbs, omega)
    for which in (0, 1):
        A, Bm = a.matrix(which), b.matrix(which)
        assert np.max(np.abs(A - Bm)) < 1e-11 * np.max(np.abs(Bm)), which
    Q = a.matrix(0)
    assert np.allclose(Q, Q.T, rtol=0, atol=1e-12 * np.abs(Q).max()) and np.all(np.linalg.eigvalsh(Q) > 0)
    Linv = a.matrix(1)
    # L^-T L^-1 = Q^-1 with Q the plain stencil of the level
    nb = _neighbours(Mt, level % 2 == 1)
    Qp = _stencil_matrix(nb, [4.0 + a.mu2, -1.0])
    assert np.allclose(Linv.T @ Linv @ Qp, np.eye(a.N), atol=1e-10)


def _neighbours(Mt, rotated):
    from mlmcpathintegral_amd import abi
    n = Mt * Mt // 2 if rotated else Mt * Mt
    out = np.zeros(8 * n, dtype=np.uint32)
    abi.call("mlmcpi_neighbours_2d", Mt, Mt, int(rotated), out.ctypes.data_as(C.c_void_p))
    return out.reshape(n, 8)


def _stencil_matrix(nb, stencil):
    n = nb.shape[0]
    Q = np.zeros((n, n))
    for l in range(n):
        Q[l, l] += stencil[0]
        for s in range(len(stencil) - 1):
            for k in range(4):
                Q[l, nb[l, 4 * s + k]] += stencil[s + 1]
    return Q


def test_smoothed_coarse_action_interpolates_between_plain_and_exact_marginal():
    """What the coarse GFF action IS: with no smoothing it is the plain stencil of the coarse lattice; with many Gibbs
    sweeps it becomes the exact marginal of the fine action on the coarse vertices (the Schur complement, which is what the
    fill-in of GFFConditionedFineAction assumes) -- checked against the marginal computed from the FINE level's precision
    matrix, so the factor sqrt(2) in the lattice spacing of the rotated level, the 9-point stencil and the vertex maps all
    have to be right."""
    Mt, mass = 8, 10.0
    fine = Level(Mt, ROTATE, 0, mass)
    pairs, _ = fine.tables()
    Qf = _stencil_matrix(_neighbours(Mt, False), [4.0 + fine.mu2, -1.0])
    cov = np.linalg.inv(Qf)
    fidx, cidx = pairs[0::2], pairs[1::2]
    marg = np.zeros((fine.n_coarse, fine.n_coarse))
    marg[np.ix_(cidx, cidx)] = cov[np.ix_(fidx, fidx)]      # covariance of the coarse vertices, in coarse numbering
    Q_marginal = np.linalg.inv(marg)
    plain = Level(Mt, ROTATE, 1, mass, 0).matrix(0)
    assert np.allclose(plain, _stencil_matrix(_neighbours(Mt, True), [4.0 + 2.0 * fine.mu2, -1.0]), atol=1e-12)
    err = [np.max(np.abs(Level(Mt, ROTATE, 1, mass, n).matrix(0) - Q_marginal)) for n in (0, 2, 8, 60)]
    assert err[0] > err[1] > err[2] > err[3] and err[3] < 1e-9 * np.abs(Q_marginal).max(), err


# ---- the reference author's own numpy construction (tests/golden/gff_qhat.json, made by make_gff_qhat_fixture.py) --------
def _qhat_fixture():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "gff_qhat.json")) as f:
        return json.load(f)


def _fixture_fields(Mlat):
    """the three deterministic fields of make_gff_qhat_fixture.py (Cartesian coordinates: permutation-free)"""
    i, j = np.meshgrid(np.arange(Mlat), np.arange(Mlat), indexing="ij")
    return [np.sin(0.7 * i + 1.3 * j + 0.2), np.cos(2.0 * np.pi * i / Mlat) * (1.0 + 0.1 * j), ((3 * i + 5 * j) % 7) / 7.0 - 0.4]


def _rotated_coords(Mt, n):
    from mlmcpathintegral_amd import abi
    out = []
    for ell in range(n):
        i, j = C.c_int(), C.c_int()
        abi.load().mlmcpi_vertex_lin2cart(Mt, Mt, 1, ell, C.byref(i), C.byref(j))
        out.append((i.value, j.value))
    return out


@pytest.mark.parametrize("case", [c for c in _qhat_fixture()["qhat"] if c["order"] == "lattice2d" or c["nsmooth"] == 0],
                         ids=lambda c: f"M{c['Mlat']}-n{c['nsmooth']}-{c['order']}")
def test_qhat_matches_the_reference_authors_numpy_construction(orc, case):
    """Q-hat of the first rotated level, Sigma_iter = Sigma + G^n (Sigma_initial - Sigma) G^n^T inverted
    (gffaction.cc:133-166), against /root/reference/python/gibbs_smoother.py:41-55 run by the fixture generator -- the
    reference author's own construction of the same object, with the same stencils (coarse mu2 = 2 x fine).  Library
    (mlmcpi_gff_level_matrix) and oracle both reproduce 1/2 phi^T Q-hat phi for three fields and the diagonal and one row
    of Sigma_iter = Q-hat^-1.  For nsmooth > 0 the matrix depends on the order of the Gibbs sweep: the cases compared are
    the ones the generator built in Lattice2D's numbering of the rotated vertices (order = lattice2d)."""
    Mlat, n = case["Mlat"], case["nsmooth"]
    for lv in (Level(Mlat, ROTATE, 1, case["mass"], n), OLevel(orc, Mlat, ROTATE, 1, case["mass"], n)):
        Q = lv.matrix(0)
        coords = _rotated_coords(Mlat, lv.N)
        for f, want in zip(_fixture_fields(Mlat), case["energies"]):
            phi = np.array([f[i, j] for i, j in coords])
            assert abs(0.5 * phi @ Q @ phi - want) < 1e-10 * max(1.0, abs(want))
        Sigma = np.linalg.inv(Q)
        lin = {c: l for l, c in enumerate(coords)}
        for i, j, v in case["sigma_iter_diag"]:
            assert abs(Sigma[lin[(i, j)], lin[(i, j)]] - v) < 1e-10 * abs(v)
        for i, j, v in case["sigma_iter_row0"]:
            assert abs(Sigma[lin[(0, 0)], lin[(i, j)]] - v) < 1e-11


def test_gibbs_sweep_order_matters_for_qhat():
    """why only the lattice2d-ordered cases can be compared for nsmooth > 0: the same construction swept in the Python
    class's own vertex order gives a different Q-hat (energies differ in the 3rd-4th digit), and the same one for nsmooth = 0"""
    cases = {(c["Mlat"], c["nsmooth"], c["order"]): c for c in _qhat_fixture()["qhat"]}
    for Mlat in (8, 16):
        a, b = cases[(Mlat, 0, "python")], cases[(Mlat, 0, "lattice2d")]
        assert np.allclose(a["energies"], b["energies"], rtol=1e-12)
        a, b = cases[(Mlat, 2, "python")], cases[(Mlat, 2, "lattice2d")]
        assert not np.allclose(a["energies"], b["energies"], rtol=1e-6)


@pytest.mark.parametrize("case", _qhat_fixture()["twolevel_terms"], ids=lambda c: f"M{c['Mlat']}-m{c['mass']}-n{c['nsmooth']}")
def test_two_level_terms_match_the_reference_authors_python(orc, case):
    """The three actions of TwoLevelMetropolisStep::draw for the GFF (twolevelmetropolisstep.cc:35-89 with GFFAction::evaluate,
    gffaction.cc:8-30, the smoothed coarse action, :133-166, and GFFConditionedFineAction::evaluate,
    gffconditionedfineaction.cc:7-49), against /root/reference/python/gff_twolevel.py (GFFAction.evaluate, evaluate_fillin,
    CoarseGibbsSmoother.evaluate, TwoLevelSampler.step), run by the fixture generator on two deterministic fields: the
    oracle reproduces S_fine, S_fillin, S_coarse of each field and Delta S of the pair to 1e-10 -- the vertex maps, the
    restriction to the rotated level, the fill-in's neighbour sums and variance included."""
    L = orc.lib()
    Mlat, mass, n = case["Mlat"], case["mass"], case["nsmooth"]
    F, Cl = OLevel(orc, Mlat, ROTATE, 0, mass, 0), OLevel(orc, Mlat, ROTATE, 1, mass, n)
    f = _fixture_fields(Mlat)
    got = {}
    for name, field in (("current", f[0]), ("proposal", f[2])):
        phi = np.array([field[l % Mlat, l // Mlat] for l in range(Mlat * Mlat)])   # vertex l = Mt j + i
        coarse = np.zeros(Cl.N)
        L.orc_gff_copy(F.h, phi, coarse, 1)
        got[name] = {"S_fine": L.orc_gff_level_evaluate(F.h, phi), "S_fillin": L.orc_gff_cfa_evaluate(F.h, phi),
                     "S_coarse": L.orc_gff_level_evaluate(Cl.h, coarse)}
        for k, v in case["terms"][name].items():
            assert abs(got[name][k] - v) < 1e-10 * max(1.0, abs(v)), (name, k, got[name][k], v)
    dS = (got["proposal"]["S_fine"] - got["current"]["S_fine"]) + (got["current"]["S_coarse"] - got["proposal"]["S_coarse"]) \
        + (got["current"]["S_fillin"] - got["proposal"]["S_fillin"])
    assert abs(dS - case["DeltaS"]) < 1e-9


def test_phi_squared_closed_form_matches_the_reference_authors_python(orc):
    """gff_phi_squared_analytical (auxilliary.cc:197-209) against QoISquaredField.exact_value of gff_twolevel.py:186-193"""
    for c in _qhat_fixture()["phi_squared_exact"]:
        got = orc.lib().orc_gff_phi_squared_analytical(c["mass"], c["Mlat"], c["Mlat"])
        assert abs(got - c["phi_squared"]) < 1e-12 * c["phi_squared"], (c, got)


def test_three_level_acceptance_is_the_plain_vs_marginal_mismatch():
    """with three GFF levels (16 x 16 -> rotated -> 8 x 8) the lower two-level step accepts about 8 %.  The
    reading: the fill-in of the rotated level (gffconditionedfineaction.cc:7-25) is exact for the PLAIN stencil there, so
    a proposal on that level is distributed (nearly) like the plain stencil's Gaussian, while the level's action is the
    Gibbs-smoothed Q-hat, close to the 9-point marginal of the finer level: an independence Metropolis test between the two.
    (a) numpy, with the library's own matrices: the expected acceptance of exactly that step is 0.05 ... 0.12;
    (b) the reference author's Python experiment of the same test (GFFAction.draw_5pt / draw_9pt / evaluate_*,
        gff_twolevel_coarse.py:8-152, run by the fixture generator at the same mu2 on square lattices of 64 and 144
        vertices) brackets it: 0.20 at 64, 0.055 at 144 vertices; the rotated level has 128."""
    rng = np.random.default_rng(7)
    Mt, mass, n_gibbs = 16, 10.0, 2
    # (a level handle takes the extents of ITS lattice: level 2 of the 16 x 16 hierarchy is the unrotated 8 x 8 lattice)
    l1, l2 = Level(Mt, ROTATE, 1, mass, n_gibbs), Level(Mt // 2, ROTATE, 2, mass, n_gibbs)
    assert l2.N == l1.n_coarse and abs(l2.mu2 - 2.0 * l1.mu2) < 1e-14
    Q1, Q2 = l1.matrix(0), l2.matrix(0)
    pairs, fineonly = l1.tables()
    fidx, cidx = pairs[0::2], pairs[1::2]
    nb = _neighbours(Mt, True)
    kappa = 4.0 + l1.mu2
    C1, C2 = np.linalg.cholesky(np.linalg.inv(Q1)), np.linalg.cholesky(np.linalg.inv(Q2))

    def cfa(theta):  # -log density of the fine-only vertices given the coarse ones (up to a constant)
        mean = theta[nb[fineonly, :4]].sum(axis=1) / kappa
        return 0.5 * kappa * np.sum((theta[fineonly] - mean) ** 2)

    acc = []
    for _ in range(4000):
        theta = C1 @ rng.standard_normal(l1.N)                      # current state ~ the level's action
        x_new = C2 @ rng.standard_normal(l2.N)                      # coarse proposal ~ the coarser level's action
        prop = np.zeros(l1.N)
        prop[fidx] = x_new[cidx]
        prop[fineonly] = prop[nb[fineonly, :4]].sum(axis=1) / kappa + rng.standard_normal(len(fineonly)) / math.sqrt(kappa)
        x_old = np.zeros(l2.N)
        x_old[cidx] = theta[fidx]
        dS = (0.5 * prop @ Q1 @ prop - 0.5 * theta @ Q1 @ theta) - (0.5 * x_new @ Q2 @ x_new - 0.5 * x_old @ Q2 @ x_old) \
            - (cfa(prop) - cfa(theta))
        acc.append(min(1.0, math.exp(-dS)))
    p = float(np.mean(acc))
    ref = {a["ndof"]: a["mean_acceptance_probability"] for a in _qhat_fixture()["independence_acceptance_5pt_vs_9pt"]}
    assert ref[144] < p < ref[64], (p, ref)
    assert 0.05 < p < 0.12, p


# ---- GPU -----------------------------------------------------------------------------------------------------------------
def _dev(abi, name, *args):
    abi.call(name, *args)


@pytest.mark.gpu
@pytest.mark.parametrize("Mt,level,n_gibbs", [(8, 0, 0), (8, 1, 2), (16, 1, 2), (16, 0, 0), (8, 1, 0)])
def test_level_evaluate_and_draw_match_oracle(gpu_ops, orc, Mt, level, n_gibbs):
    import torch
    from mlmcpathintegral_amd import abi
    a, b = Level(Mt, ROTATE, level, 10.0, n_gibbs), OLevel(orc, Mt, ROTATE, level, 10.0, n_gibbs)
    B, seed = 5, 31
    phi = torch.zeros((B, a.N), dtype=torch.float64, device="cuda")
    abi.call("mlmcpi_gff_level_draw", a.h, C.c_void_p(phi.data_ptr()), B, seed, 3, 7, None)
    S = torch.zeros(B, dtype=torch.float64, device="cuda")
    abi.call("mlmcpi_gff_level_evaluate", a.h, C.c_void_p(phi.data_ptr()), B, C.c_void_p(S.data_ptr()), None)
    got, gotS = phi.cpu().numpy(), S.cpu().numpy()
    for c in range(B):
        want = np.zeros(a.N)
        b.L.orc_gff_level_dev_draw(b.h, want, seed, 3 + c, 7)
        assert np.max(np.abs(got[c] - want)) < 1e-11 * max(1.0, np.abs(want).max())
        assert abs(gotS[c] - b.L.orc_gff_level_evaluate(b.h, want)) < 1e-10 * max(1.0, abs(gotS[c]))


@pytest.mark.gpu
@pytest.mark.parametrize("Mt,level", [(8, 0), (8, 1), (16, 0), (16, 1)])
def test_gff_twolevel_step_matches_oracle(gpu_ops, orc, Mt, level):
    """fill-in, conditioned fine action, the three action differences and the accept decision of the two-level step
    between a GFF level and its coarsening (unrotated -> rotated and rotated -> unrotated), against the oracle"""
    import torch
    from mlmcpathintegral_amd import abi
    mass = 10.0
    fine, coarse = Level(Mt, ROTATE, level, mass, 0 if level == 0 else 2), None
    Mt_c = fine.Mt_c
    coarse = Level(Mt_c, ROTATE, level + 1, mass, 2)
    ofine, ocoarse = OLevel(orc, Mt, ROTATE, level, mass, 0 if level == 0 else 2), OLevel(orc, Mt_c, ROTATE, level + 1, mass, 2)
    B, seed = 6, 77
    rng = np.random.default_rng(5)
    theta0 = 0.3 * rng.normal(size=(B, fine.N))
    phic0 = 0.3 * rng.normal(size=(B, coarse.N))
    theta = torch.tensor(theta0, device="cuda")
    phic = torch.tensor(phic0, device="cuda")
    nbytes = C.c_size_t()
    abi.call("mlmcpi_gff_twolevel_workspace_bytes", fine.h, B, C.byref(nbytes))
    work = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(B, dtype=torch.int32, device="cuda")
    terms = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
    abi.call("mlmcpi_gff_twolevel_draw", fine.h, coarse.h, C.c_void_p(phic.data_ptr()), C.c_void_p(theta.data_ptr()), B, seed, 2, 9,
             C.c_void_p(work.data_ptr()), C.c_void_p(acc.data_ptr()), C.c_void_p(terms.data_ptr()), None)
    got, gacc, gterms = theta.cpu().numpy(), acc.cpu().numpy(), terms.cpu().numpy()
    n_acc = 0
    for c in range(B):
        want = theta0[c].copy()
        t = np.zeros(3)
        a = ofine.L.orc_gff_dev_twolevel_draw(ofine.h, ocoarse.h, np.ascontiguousarray(phic0[c]), want, seed, 2 + c, 9, t)
        assert np.allclose(gterms[c], t, rtol=1e-9, atol=1e-9), (gterms[c], t)
        assert a == gacc[c]
        assert np.max(np.abs(got[c] - want)) < 1e-11
        n_acc += a
    # conditioned fine action on its own
    S = torch.zeros(B, dtype=torch.float64, device="cuda")
    abi.call("mlmcpi_gff_cfa_evaluate", fine.h, C.c_void_p(theta.data_ptr()), B, C.c_void_p(S.data_ptr()), None)
    for c in range(B):
        assert abs(S[c].item() - ofine.L.orc_gff_cfa_evaluate(ofine.h, np.ascontiguousarray(got[c]))) < 1e-10


@pytest.mark.gpu
def test_gff_two_level_chain_samples_the_fine_distribution(gpu_ops, orc):
    """16 x 16 GFF (mass 10, CoarsenRotate): coarse samples from the smoothed coarse action's own exact sampler
    (GFFAction::draw with n_gibbs_smooth = 2, the reference's GFFSamplerFactory), filled in and accepted by the two-level
    step: <phi^2> of the fine chain against gff_phi_squared_analytical, and a high acceptance rate (the coarse action is
    two Gibbs sweeps away from the exact marginal)."""
    import torch
    from conftest import zcheck
    from mlmcpathintegral_amd import abi
    Mt, mass, B, n_burn, n = 16, 10.0, 512, 20, 200
    fine, coarse = Level(Mt, ROTATE, 0, mass, 0), Level(Mt, ROTATE, 1, mass, 2)
    theta = torch.zeros((B, fine.N), dtype=torch.float64, device="cuda")
    phic = torch.zeros((B, coarse.N), dtype=torch.float64, device="cuda")
    nbytes = C.c_size_t()
    abi.call("mlmcpi_gff_twolevel_workspace_bytes", fine.h, B, C.byref(nbytes))
    work = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(B, dtype=torch.int32, device="cuda")
    vals, n_acc = [], 0
    for k in range(n_burn + n):
        abi.call("mlmcpi_gff_level_draw", coarse.h, C.c_void_p(phic.data_ptr()), B, 5, 0, k, None)
        abi.call("mlmcpi_gff_twolevel_draw", fine.h, coarse.h, C.c_void_p(phic.data_ptr()), C.c_void_p(theta.data_ptr()), B, 5, 0, k,
                 C.c_void_p(work.data_ptr()), C.c_void_p(acc.data_ptr()), None, None)
        if k >= n_burn:
            vals.append(gpu_ops.qoi_phi_squared(theta))
            n_acc += int(acc.sum().item())
    v = torch.stack(vals).mean(dim=0)
    m, e = float(v.mean()), float(v.std(unbiased=True)) / math.sqrt(B)
    exact = orc.lib().orc_gff_phi_squared_analytical(mass, Mt, Mt)
    p_acc = n_acc / (n * B)
    print(f"GFF two-level chain 16^2: <phi^2> = {m:.6f} +- {e:.6f} (exact {exact:.6f}), acceptance {p_acc:.3f}")
    zcheck("GFF two-level chain 16^2: <phi^2> vs closed form", m, e, exact)
    # the acceptance rate expected from the reference author's matrices with Lattice2D's order of the Gibbs sweep
    # (tests/golden/gff_qhat.json: 0.9863 +- 0.0001; the Python class's own sweep order would give 0.9910)
    want = next(a for a in _qhat_fixture()["twolevel_expected_acceptance"] if a["order"] == "lattice2d")
    zcheck("GFF two-level chain 16^2: acceptance vs the reference author's matrices", p_acc,
           math.sqrt(p_acc * (1.0 - p_acc) / (n * B)) * 3.0, want["mean"], want["error"])   # (x 3: successive steps of a chain are correlated)
    assert p_acc > 0.5
