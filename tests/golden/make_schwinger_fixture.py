#!/usr/bin/env python3
"""Generates tests/golden/schwinger_ref_python.json and tests/golden/reference_printed_outputs.json from the reference
author's own Python restatements of the quenched Schwinger plaquette / link maps and of the heat-bath and fill-in
densities.  Runs in the build container only (it imports /root/reference/tools/*.py; nothing of those files is copied:
the fixtures hold numbers).

    python tests/golden/make_schwinger_fixture.py

What is imported and what it pins:
  * /root/reference/tools/plot_schwinger_configuration.py
      mod_2pi (:35-37)                     = common/auxilliary.hh:42-44
      Lattice2d.lin2cart / cart2lin (:45-56) = Lattice2D::link_lin2cart / link_cart2lin (lattice/lattice2d.hh:348-375)
      Configuration.get_plaquette (:69-76) = mod_2pi of theta_P of QuenchedSchwingerAction::evaluate
                                             (action/qft/quenchedschwingeraction.cc:13-17), the summand of
                                             QoIAvgPlaquette / QoI2DSusceptibility (qoi/qft/*.cc)
    The class reads the module global `m_lat` in its constructor (:41-42): it is set before each construction.
    Fields are given by Cartesian coordinates, so the fixture does not depend on any index convention of ours.
  * /root/reference/tools/plot_distribution.py  (f_analytical of each wrapper class)
      ExpSin2Distribution (:103-122)                  = ExpSin2Distribution::evaluate (distribution/expsin2distribution.cc:20-24)
      ExpCosDistribution (:154-168)                   = ExpCosDistribution::evaluate (expcosdistribution.cc:7-21)
      BesselProductDistribution (:170-193)            = BesselProductDistribution::evaluate (besselproductdistribution.cc:6-25);
                                                        normalised by quadrature here, by the cosine series in the C++
      ApproximateBesselProductDistribution (:195-259) = ApproximateBesselProductDistribution::evaluate
                                                        (approximatebesselproductdistribution.cc:6-54)
    The last one DIFFERS from the C++ in the weight of the two Gaussian combs: the Python has
    rho = (s_m/s_p)^1.5 exp(-4 (s_p - s_m)) (:213), the C++ pow(s_p/s_m, 1.5) exp(-4 (s_p - s_m)) (.cc:50-51) -- and
    sums 33 images where the C++ sums 9, and has no small-x0 branch (x0 < pi/8: one comb of width 1/beta, .cc:46-49).  The fixture therefore records, next to the
    class's own f_analytical and N_p, the two combs separately (the class's f_analytical with the instance attributes
    N_p, N_m set to (1, 0) and (0, 1)); the oracle follows the C++ weight and is compared comb by comb.
  * The docstring of plot_distribution.py (:16-37) and the doc comment of src/test_distribution.cc (:433-449) hold
    OUTPUT of the reference binary: BesselProductDistribution beta = 4, x_p = 2.82743, x_m = 0 at four points.  Those
    four (x, y) pairs are transcribed verbatim into reference_printed_outputs.json (6 significant digits).  The three
    sample values printed beside them are not used: they depend on the engine history of that run (the distribution
    keeps a normal_distribution with a cached variate), which the file does not record.
"""
import importlib.util
import json
import math
import os
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np  # noqa: E402

REF = "/root/reference/tools"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def link_field(kind, m, i, j, mu):
    """deterministic link angles in [-pi, pi), given by Cartesian coordinates"""
    if kind == 0:
        return 3.0 * math.sin(1.0 + 0.9 * i + 1.7 * j + 0.4 * mu)
    if kind == 1:
        return math.pi * (((7 * i + 11 * j + 3 * mu) % 13) / 6.5 - 1.0)
    return 2.5 * math.cos(2.0 * math.pi * (i + 2 * j) / m + 0.3 * mu) - 0.6 * mu


def schwinger_cases(ps):
    cases = []
    for m in (4, 6, 16):
        ps.m_lat = m  # Lattice2d.__init__ reads this module global
        lat = ps.Lattice2d(m)
        maps = [[i, j, mu, int(lat.cart2lin(i, j, mu))] for j in range(m) for i in range(m) for mu in range(2)]
        wrapped = [[i, j, mu, int(lat.cart2lin(i, j, mu))] for (i, j) in ((-1, -1), (m, 0), (0, m), (m + 1, -2), (-m, 2 * m - 1))
                   for mu in range(2)]
        inverse = [[ell] + [int(v) for v in lat.lin2cart(ell)] for ell in range(2 * m * m)]
        for kind in range(3):
            data = np.zeros(2 * m * m)
            links = []
            for j in range(m):
                for i in range(m):
                    for mu in range(2):
                        v = link_field(kind, m, i, j, mu)
                        data[lat.cart2lin(i, j, mu)] = v
                        links.append([i, j, mu, v])
            cfg = ps.Configuration(lat, data)
            plaq = [[i, j, float(cfg.get_plaquette(i, j))] for j in range(m) for i in range(m)]
            cases.append({"m": m, "field": kind, "links": links, "plaquettes": plaq,
                          "cart2lin": maps if kind == 0 else None, "cart2lin_wrapped": wrapped if kind == 0 else None,
                          "lin2cart": inverse if kind == 0 else None})
    return cases


def mod_2pi_cases(ps):
    pi = math.pi
    xs = [0.0, 1.0, -1.0, pi, -pi, 3 * pi, -3 * pi, 2 * pi, -2 * pi, pi - 1e-12, -pi + 1e-12, pi + 1e-12, -pi - 1e-12,
          5 * pi, -5 * pi, 6.0, -6.0, 7.5, -7.5, 12.566, -12.566, 100.0, -100.0, 1e-300, -1e-300, 3.0, -3.0, 3.2, -3.2,
          9.42, -9.42, 9.43, -9.43, 0.5 * pi, -0.5 * pi, 1.5 * pi, -1.5 * pi, 31.4, -31.4, 2.0 * pi - 1e-9]
    assert len(xs) == 40
    return [[x, float(ps.mod_2pi(x))] for x in xs]


POINTS = [-math.pi + 2.0 * math.pi * k / 128.0 for k in range(129)]
TRIPLES = [(4.0, 2.82743, 0.0), (1.0, 0.3, -2.9), (8.0, 3.1, -3.1), (8.0, 1.2, -0.7)]  # the third: x0 < pi/8, the C++'s small-x0 branch


def distribution_cases(pd):
    out = {"points": POINTS, "expsin2": [], "expcos": [], "bessel_product": [], "approx_bessel_product": []}
    for sigma in (0.1, 4.0, 64.0):
        d = pd.ExpSin2Distribution(None, None, None, sigma)
        out["expsin2"].append({"sigma": sigma, "pdf": [float(d.f_analytical(x)) for x in POINTS]})
    for (beta, x_p, x_m) in TRIPLES:
        d = pd.ExpCosDistribution(None, None, None, beta, x_p, x_m)
        out["expcos"].append({"beta": beta, "x_p": x_p, "x_m": x_m, "pdf": [float(d.f_analytical(x)) for x in POINTS]})
        d = pd.BesselProductDistribution(None, None, None, beta, x_p, x_m)
        out["bessel_product"].append({"beta": beta, "x_p": x_p, "x_m": x_m, "Znorm": float(d.Znorm),
                                      "pdf": [float(d.f_analytical(x)) for x in POINTS]})
        d = pd.ApproximateBesselProductDistribution(None, None, None, beta, x_p, x_m)
        rec = {"beta": beta, "x_p": x_p, "x_m": x_m, "x0": float(d.x0), "sign_flip": int(d.sign_flip),
               "sigma2_inv_p": float(d.sigma2_inv_p), "sigma2_inv_m": float(d.sigma2_inv_m), "N_p_python": float(d.N_p),
               "pdf_python": [float(d.f_analytical(x)) for x in POINTS]}
        d.N_p, d.N_m = 1.0, 0.0
        rec["comb_p"] = [float(d.f_analytical(x)) for x in POINTS]
        d.N_p, d.N_m = 0.0, 1.0
        rec["comb_m"] = [float(d.f_analytical(x)) for x in POINTS]
        out["approx_bessel_product"].append(rec)
    return out


def main():
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            ps = load("plot_schwinger_configuration")
            pd = load("plot_distribution")
            fixture = {
                "_provenance": "tests/golden/make_schwinger_fixture.py importing /root/reference/tools/plot_schwinger_configuration.py "
                               "and plot_distribution.py (the reference author's Python); numbers only",
                "mod_2pi": mod_2pi_cases(ps),
                "schwinger": schwinger_cases(ps),
                "distributions": distribution_cases(pd),
            }
        finally:
            os.chdir(cwd)
    with open(os.path.join(HERE, "schwinger_ref_python.json"), "w") as f:
        json.dump(fixture, f)
    printed = {
        "_provenance": "transcribed verbatim from the docstring of /root/reference/tools/plot_distribution.py:16-37 "
                       "(= the doc comment of src/test_distribution.cc:433-449): output of the reference binary, 6 significant digits",
        "distribution": "BesselProductDistribution", "beta": 4, "x_p": 2.82743, "x_m": 0,
        "points": [[-3.14159, 0.0371872], [-3.09251, 0.0364034], [3.09251, 0.0385538], [3.14159, 0.0371872]],
        "samples_not_used": [2.05174, 1.82559, 0.806089],
    }
    with open(os.path.join(HERE, "reference_printed_outputs.json"), "w") as f:
        json.dump(printed, f, indent=1)
    print("wrote schwinger_ref_python.json (%d bytes) and reference_printed_outputs.json" %
          os.path.getsize(os.path.join(HERE, "schwinger_ref_python.json")))


if __name__ == "__main__":
    main()
