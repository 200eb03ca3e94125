"""ctypes front end of the CPU oracle (oracle/oracle.cc).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package ``mlmcpathintegral_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

HARMONIC, QUARTIC, ROTOR, GFF, SCHWINGER = range(5)
P_MOMENTUM, P_ACCEPT, P_GFF_NORMAL, P_VONMISES, P_INIT = 1, 2, 3, 4, 6


def build(force=False):
    src = os.path.join(_HERE, "oracle.cc")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64, _u32, _i, _d, _vp = C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_void_p

_SIGS = {
    "orc_philox4x32_10": (None, [_up, _up, _up]),
    "orc_dev_random": (None, [_u64, _u32, _u32, _u32, _u32, _u32, _dp]),
    "orc_mod_2pi": (_d, [_d]),
    "orc_vertex_cart2lin": (_u32, [_i, _i, _i, _i, _i]),
    "orc_vertex_lin2cart": (None, [_i, _i, _i, _u32, C.POINTER(_i), C.POINTER(_i)]),
    "orc_link_cart2lin": (_u32, [_i, _i, _i, _i, _i]),
    "orc_link_lin2cart": (None, [_i, _i, _u32, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "orc_neighbours2d": (None, [_i, _i, _i, _up]),
    "orc_neighbours1d": (None, [_u32, _up]),
    "orc_action_1d": (_vp, [_i, _u32, _d, _d, _d, _d, _d]),
    "orc_action_gff": (_vp, [_i, _i, _d]),
    "orc_action_schwinger": (_vp, [_i, _i, _d]),
    "orc_action_free": (None, [_vp]),
    "orc_action_size": (_u32, [_vp]),
    "orc_action_evaluate": (_d, [_vp, _dp]),
    "orc_action_force": (None, [_vp, _dp, _dp]),
    "orc_action_overrelaxation_update": (_i, [_vp, _dp, _u32]),
    "orc_action_heatbath_update": (_i, [_vp, _dp, _u32]),
    "orc_action_initialise_state": (None, [_vp, _dp]),
    "orc_action_wminimum": (_d, [_vp, _d, _d]),
    "orc_action_wcurvature": (_d, [_vp, _d, _d]),
    "orc_action_gff_mu2": (_d, [_vp]),
    "orc_action_staples": (None, [_vp, _dp, _u32, C.POINTER(_d), C.POINTER(_d)]),
    "orc_expcos_draws": (None, [_u64, _d, _d, _d, _u32, _dp]),
    "orc_expsin2_draws": (None, [_u64, _d, _u32, _dp]),
    "orc_dev_expcos_draw": (_d, [_u64, _u32, _u32, _u32, _d, _d, _d]),
    "orc_dev_expsin2_draw": (_d, [_u64, _u32, _u32, _u32, _d]),
    "orc_dev_vs_draw": (_d, [_u64, _u32, _u32, _u32, _d, _d, _d]),
    "orc_vs_tables": (None, [_d, _vp, _vp]),
    "orc_qoi_xsquared": (_d, [_dp, _u32]),
    "orc_qoi_susceptibility": (_d, [_dp, _u32, _d]),
    "orc_qoi_2d_susceptibility": (_d, [_dp, _i, _i]),
    "orc_qoi_avg_plaquette": (_d, [_dp, _i, _i]),
    "orc_qoi_2d_phi_squared": (_d, [_dp, _u32]),
    "orc_hmc_new": (_vp, [_vp, _u32, _d, _u32, _u32, _i, _u32, _u32]),
    "orc_hmc_free": (None, [_vp]),
    "orc_hmc_draw": (_i, [_vp, _dp]),
    "orc_hmc_set_state": (None, [_vp, _dp]),
    "orc_hmc_get_state": (None, [_vp, _dp]),
    "orc_hmc_dt": (_d, [_vp]),
    "orc_hmc_tuned": (_i, [_vp]),
    "orc_hmc_p_accept": (_d, [_vp]),
    "orc_hmc_reset_stats": (None, [_vp]),
    "orc_heatbath_new": (_vp, [_vp, _u32, _u32, _u32, _i]),
    "orc_heatbath_free": (None, [_vp]),
    "orc_heatbath_draw": (None, [_vp, _dp]),
    "orc_heatbath_set_state": (None, [_vp, _dp]),
    "orc_dev_sweep": (None, [_vp, _dp, _i, _u64, _u32, _u32]),
    "orc_dev_site_update": (None, [_vp, _dp, _u32, _i, _u64, _u32, _u32]),
    "orc_dev_hmc_trajectory": (_i, [_vp, _dp, _u32, _d, _u64, _u32, _u32, _dp, _dp]),
    "orc_dev_initialise": (None, [_vp, _dp, _u64, _u32]),
    "orc_dev_twolevel_draw": (_i, [_vp, _vp, _dp, _dp, _u64, _u32, _u32, _dp]),
    "orc_dev_lattice_twolevel_draw": (_i, [_vp, _vp, _dp, _dp, _u64, _u32, _u32, _dp]),
    "orc_dev_lattice_twolevel_draw_cfa": (_i, [_vp, _vp, _i, _dp, _dp, _u64, _u32, _u32, _dp]),
    "orc_gaussfill_pdf": (_d, [_d, _dp, _dp]),
    "orc_gaussfill_dev_draw": (None, [_d, _dp, _u64, _u32, _u32, _u32, _dp]),
    "orc_expcos_pdf": (_d, [_d, _d, _d, _d]),
    "orc_i0_scaled": (_d, [_d]),
    "orc_expsin2_pdf": (_d, [_d, _d]),
    "orc_bessel_product_pdf": (_d, [_d, _d, _d, _d]),
    "orc_approx_bessel_pdf": (_d, [_d, _d, _d, _d]),
    "orc_approx_bessel_params": (None, [_d, _d, _dp]),
    "orc_schwinger_plaquettes": (None, [_vp, _dp, _dp]),
    "orc_mlmc_ref_run": (None, [_i, _u32, _d, _d, _d, _d, _d, _u32, _u32, _d, _u32, _u32, _u32, _u32, _i, _u64, _dp, _dp]),
    "orc_single_level_ref_run": (None, [_i, _u32, _d, _d, _d, _d, _d, _u32, _u32, _d, _u32, _u32, _u32, _u64, _dp]),
    "orc_bessel_product_ref_draws": (None, [_u64, _d, _d, _d, _u32, _u32, _dp]),
    "orc_ho_cholesky": (_i, [_vp, _dp]),
    "orc_dev_exact_draw": (None, [_vp, _dp, _dp, _u64, _u32, _u32]),
    "orc_dev_gff_exact_draw": (None, [_vp, _dp, _u64, _u32, _u32]),
    "orc_schwinger_copy_from_fine": (None, [_i, _i, _i, _i, _dp, _dp]),
    "orc_schwinger_copy_from_coarse": (None, [_i, _i, _i, _i, _dp, _dp]),
    "orc_gff_transfer": (None, [_i, _i, _i, _i, _dp, _dp, _i]),
    "orc_stats_new": (_vp, [_u32]),
    "orc_stats_free": (None, [_vp]),
    "orc_stats_record": (None, [_vp, _dp, _u32]),
    "orc_stats_reset": (None, [_vp, _i]),
    "orc_stats_get": (None, [_vp, _dp]),
    "orc_gff_level_new": (_vp, [_u32, _u32, _i, _i, _d, _i, _d]),
    "orc_gff_level_free": (None, [_vp]),
    "orc_gff_level_size": (_u32, [_vp]),
    "orc_gff_level_n_coarse": (_u32, [_vp]),
    "orc_gff_level_mu2": (_d, [_vp]),
    "orc_gff_level_tables": (None, [_vp, _vp, _vp]),
    "orc_gff_level_matrix": (None, [_vp, _i, _dp]),
    "orc_gff_level_evaluate": (_d, [_vp, _dp]),
    "orc_gff_level_dev_draw": (None, [_vp, _dp, _u64, _u32, _u32]),
    "orc_gff_cfa_evaluate": (_d, [_vp, _dp]),
    "orc_gff_cfa_dev_fill": (_d, [_vp, _dp, _u64, _u32, _u32]),
    "orc_gff_copy": (None, [_vp, _dp, _dp, _i]),
    "orc_gff_dev_twolevel_draw": (_i, [_vp, _vp, _dp, _dp, _u64, _u32, _u32, _dp]),
    "orc_rank_seeds": (None, [_u32, _u32, _vp]),
    "orc_rank_engine_outputs": (None, [_u32, _u32, _u32, _u32, _vp]),
    "orc_ho_xsquared_analytical": (_d, [_u32, _d, _d, _d]),
    "orc_gff_phi_squared_analytical": (_d, [_d, _i, _i]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        for name, (res, args) in _SIGS.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = res, args
    return _lib


class Action:
    """Handle on an oracle action object (HO / quartic / rotor / GFF / Schwinger)."""

    def __init__(self, kind, **kw):
        L = lib()
        self.kind = kind
        if kind in (HARMONIC, QUARTIC, ROTOR):
            self.h = L.orc_action_1d(kind, kw["M"], kw["T_final"], kw.get("m0", 1.0), kw.get("mu2", 1.0),
                                     kw.get("lam", 0.0), kw.get("x0", 0.0))
        elif kind == GFF:
            self.h = L.orc_action_gff(kw["Mt"], kw["Mx"], kw["mass"])
        elif kind == SCHWINGER:
            self.h = L.orc_action_schwinger(kw["Mt"], kw["Mx"], kw["beta"])
        else:
            raise ValueError(kind)
        self.kw = kw
        self.size = L.orc_action_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_action_free(self.h)
            self.h = None

    def evaluate(self, x):
        return lib().orc_action_evaluate(self.h, np.ascontiguousarray(x, dtype=np.float64))

    def force(self, x):
        f = np.zeros(self.size)
        lib().orc_action_force(self.h, np.ascontiguousarray(x, dtype=np.float64), f)
        return f

    def overrelaxation_update(self, x, ell):
        return lib().orc_action_overrelaxation_update(self.h, x, ell)

    def heatbath_update(self, x, ell):
        return lib().orc_action_heatbath_update(self.h, x, ell)

    def initialise_state(self):
        x = np.zeros(self.size)
        lib().orc_action_initialise_state(self.h, x)
        return x

    def dev_sweep(self, x, heatbath, seed, chain, step):
        lib().orc_dev_sweep(self.h, x, int(heatbath), seed, chain, step)

    def dev_site_update(self, x, site, heatbath, seed, chain, step):
        lib().orc_dev_site_update(self.h, x, int(site), int(heatbath), seed, chain, step)

    def dev_hmc_trajectory(self, x, nt, dt, seed, chain, step):
        en = np.zeros(4)
        dH = np.zeros(1)
        acc = lib().orc_dev_hmc_trajectory(self.h, x, nt, dt, seed, chain, step, en, dH)
        return acc, en, dH[0]

    def dev_twolevel_draw(self, coarse, x_coarse, theta, seed, chain, step):
        """self = fine action; returns (accept, [dS_fine, dS_coarse, dS_trial]); theta updated in place."""
        terms = np.zeros(3)
        acc = lib().orc_dev_twolevel_draw(self.h, coarse.h, np.ascontiguousarray(x_coarse), theta, seed, chain, step, terms)
        return acc, terms

    def dev_lattice_twolevel_draw(self, coarse, phi_coarse, theta, seed, chain, step, cfa_kind=0):
        """self = fine Schwinger action; returns (accept, terms); theta updated in place.  cfa_kind 1 = the Gaussian
        conditioned fine action (lattices coarsened in both directions)."""
        terms = np.zeros(3)
        acc = lib().orc_dev_lattice_twolevel_draw_cfa(self.h, coarse.h, cfa_kind, np.ascontiguousarray(phi_coarse), theta, seed, chain, step, terms)
        if acc < 0:
            raise ValueError("invalid coarsening for fill-in")
        return acc, terms

    def dev_initialise(self, seed, chain):
        x = np.zeros(self.size)
        lib().orc_dev_initialise(self.h, x, seed, chain)
        return x


class Statistics:
    def __init__(self, k_max):
        self.h = lib().orc_stats_new(k_max)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_stats_free(self.h)
            self.h = None

    def record(self, q):
        q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
        lib().orc_stats_record(self.h, q, q.size)

    def reset(self, hard=False):
        lib().orc_stats_reset(self.h, int(hard))

    def get(self):
        out = np.zeros(6)
        lib().orc_stats_get(self.h, out)
        return dict(zip(("average", "variance", "variance_error", "tau_int", "error", "samples"), out))
