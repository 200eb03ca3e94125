"""ctypes binding of the C ABI declared in include/mlmcpi_hip.h (libmlmcpi_hip.so).

This is the only way Python (tests, bench.py, smoke) reaches the HIP kernels; the C++ host layer
under include/mlmcpi/ binds the same symbols.  There is no CPU fallback: if the shared library is
missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MLMCPI_LIB_VARIANT=<suffix>: load libmlmcpi_hip_<suffix>.so from the same directory instead -- an experiment build of the
# same sources with other -D switches (make -C mlmcpathintegral_amd/csrc variant VARIANT=<suffix> EXTRA=-D...), so that two
# builds can be timed on ONE GPU box (box-to-box spread is 7 %).  Tools only; tests and bench lines use the product build.
_VARIANT = os.environ.get("MLMCPI_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, f"libmlmcpi_hip_{_VARIANT}.so" if _VARIANT else "libmlmcpi_hip.so")

HARMONIC, QUARTIC, ROTOR, GFF, SCHWINGER = range(5)


class MlmcpiError(RuntimeError):
    pass


class PathAction(C.Structure):
    """mlmcpi_path_action"""
    _fields_ = [("kind", C.c_int32), ("M", C.c_uint32), ("T_final", C.c_double), ("m0", C.c_double),
                ("mu2", C.c_double), ("lam", C.c_double), ("x0", C.c_double)]


class LatticeAction(C.Structure):
    """mlmcpi_lattice_action"""
    _fields_ = [("kind", C.c_int32), ("Mt", C.c_uint32), ("Mx", C.c_uint32), ("beta", C.c_double),
                ("mass", C.c_double)]


_vp, _u32, _u64, _i, _d, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_size_t
_PA, _LA = C.POINTER(PathAction), C.POINTER(LatticeAction)

# name -> (restype, argtypes); must list EVERY symbol of include/mlmcpi_hip.h (tests check this)
SIGNATURES = {
    "mlmcpi_abi_version": (_i, []),
    "mlmcpi_last_error": (C.c_char_p, []),
    "mlmcpi_device_count": (_i, [C.POINTER(_i)]),
    "mlmcpi_set_device": (_i, [_i]),
    "mlmcpi_device_name": (_i, [C.c_char_p, _sz]),
    "mlmcpi_malloc": (_i, [C.POINTER(_vp), _sz]),
    "mlmcpi_free": (_i, [_vp]),
    "mlmcpi_memset": (_i, [_vp, _i, _sz, _vp]),
    "mlmcpi_copy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "mlmcpi_copy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "mlmcpi_copy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "mlmcpi_stream_synchronize": (_i, [_vp]),
    "mlmcpi_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "mlmcpi_vertex_cart2lin": (_u32, [_u32, _u32, _i, _i, _i]),
    "mlmcpi_vertex_lin2cart": (None, [_u32, _u32, _i, _u32, C.POINTER(_i), C.POINTER(_i)]),
    "mlmcpi_link_cart2lin": (_u32, [_u32, _u32, _i, _i, _i]),
    "mlmcpi_link_lin2cart": (None, [_u32, _u32, _u32, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mlmcpi_neighbours_1d": (_i, [_u32, _vp]),
    "mlmcpi_neighbours_2d": (_i, [_u32, _u32, _i, _vp]),
    "mlmcpi_path_evaluate": (_i, [_PA, _vp, _u32, _vp, _vp]),
    "mlmcpi_path_force": (_i, [_PA, _vp, _vp, _u32, _vp]),
    "mlmcpi_path_initialise": (_i, [_PA, _vp, _u32, _u64, _u32, _vp]),
    "mlmcpi_qoi_xsquared": (_i, [_vp, _u32, _u32, _vp, _vp]),
    "mlmcpi_qoi_susceptibility": (_i, [_vp, _u32, _d, _u32, _vp, _vp]),
    "mlmcpi_path_hmc_workspace_bytes": (_i, [_PA, _u32, _u32, C.POINTER(_sz)]),
    "mlmcpi_path_hmc_draw": (_i, [_PA, _vp, _u32, _u32, _d, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_path_hmc_run": (_i, [_PA, _vp, _u32, _u32, _d, _u32, _u32, _i, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_path_hmc_run_layout": (_i, [_PA, _u32, _u32, C.POINTER(C.c_int32)]),
    "mlmcpi_path_sweep_draw": (_i, [_PA, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _vp]),
    "mlmcpi_path_sweep_draw_from": (_i, [_PA, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _vp, _vp]),
    "mlmcpi_path_sweep_draw_qoi": (_i, [_PA, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_path_twolevel_workspace_bytes": (_i, [_PA, _u32, C.POINTER(_sz)]),
    "mlmcpi_path_twolevel_draw": (_i, [_PA, _PA, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_path_twolevel_draw_masked": (_i, [_PA, _PA, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "mlmcpi_path_copy_from_fine": (_i, [_vp, _vp, _u32, _u32, _vp]),
    "mlmcpi_path_copy_from_coarse": (_i, [_vp, _vp, _u32, _u32, _vp]),
    "mlmcpi_ho_cholesky_factor": (_i, [_PA, _vp]),
    "mlmcpi_path_exact_draw": (_i, [_PA, _vp, _vp, _u32, _u64, _u32, _u32, _vp]),
    "mlmcpi_lattice_copy_from_fine": (_i, [_LA, _u32, _u32, _vp, _vp, _u32, _vp]),
    "mlmcpi_lattice_copy_from_coarse": (_i, [_LA, _u32, _u32, _vp, _vp, _u32, _vp]),
    "mlmcpi_lattice_twolevel_workspace_bytes": (_i, [_LA, _LA, _u32, C.POINTER(_sz)]),
    "mlmcpi_lattice_twolevel_draw": (_i, [_LA, _LA, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_lattice_twolevel_draw_cfa": (_i, [_LA, _LA, C.c_int32, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_lattice_exact_workspace_bytes": (_i, [_LA, _u32, C.POINTER(_sz)]),
    "mlmcpi_lattice_exact_draw": (_i, [_LA, _vp, _u32, _u64, _u32, _u32, _vp, _vp]),
    "mlmcpi_lattice_state_size": (_i, [_LA, C.POINTER(_u32)]),
    "mlmcpi_lattice_evaluate": (_i, [_LA, _vp, _u32, _vp, _vp]),
    "mlmcpi_lattice_force": (_i, [_LA, _vp, _vp, _u32, _vp]),
    "mlmcpi_lattice_initialise": (_i, [_LA, _vp, _u32, _u64, _u32, _vp]),
    "mlmcpi_lattice_sweep_draw": (_i, [_LA, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _u32, _vp]),
    "mlmcpi_gff_level_create": (_i, [_u32, _u32, C.c_int32, C.c_int32, _d, C.c_int32, _d, _vp]),
    "mlmcpi_gff_level_destroy": (_i, [_vp]),
    "mlmcpi_gff_level_info": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mlmcpi_gff_level_tables": (_i, [_vp, _vp, _vp]),
    "mlmcpi_gff_level_matrix": (_i, [_vp, C.c_int32, _vp]),
    "mlmcpi_gff_level_evaluate": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "mlmcpi_gff_level_draw": (_i, [_vp, _vp, _u32, _u64, _u32, _u32, _vp]),
    "mlmcpi_gff_copy_from_fine": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "mlmcpi_gff_copy_from_coarse": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "mlmcpi_gff_cfa_fill": (_i, [_vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp]),
    "mlmcpi_gff_cfa_evaluate": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "mlmcpi_gff_twolevel_workspace_bytes": (_i, [_vp, _u32, _vp]),
    "mlmcpi_gff_twolevel_draw": (_i, [_vp, _vp, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_lattice_sweep_draw_qoi": (_i, [_LA, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _u32, C.c_int32, _vp, _vp, _vp]),
    "mlmcpi_lattice_sweep_draw_qoi_record": (_i, [_LA, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _u32, C.c_int32, _vp, _vp, _vp, _vp]),
    "mlmcpi_lattice_sweep_draw_from": (_i, [_LA, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _u32, _vp, _vp]),
    "mlmcpi_lattice_sweep_draw_pingpong": (_i, [_LA, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _u32, _u32,
                                                C.POINTER(C.c_int32), _vp]),
    "mlmcpi_qoi_phi_squared": (_i, [_vp, _u32, _u32, _vp, _vp]),
    "mlmcpi_qoi_avg_plaquette": (_i, [_vp, _u32, _u32, _u32, _vp, _vp]),
    "mlmcpi_qoi_2d_susceptibility": (_i, [_vp, _u32, _u32, _u32, _vp, _vp]),
    "mlmcpi_lattice_hmc_workspace_bytes": (_i, [_LA, _u32, C.POINTER(_sz)]),
    "mlmcpi_lattice_hmc_draw": (_i, [_LA, _vp, _u32, _u32, _d, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "mlmcpi_stats_accumulate": (_i, [_vp, _vp, _u32, _vp]),
    "mlmcpi_stats_window_record": (_i, [_vp, _vp, _u32, _u32, _vp]),
    "mlmcpi_test_philox": (_i, [_vp, _vp, _vp]),
    "mlmcpi_test_random": (_i, [_u64, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "mlmcpi_test_expcos": (_i, [_u64, _u32, _u32, _d, _vp, _vp, _u32, _vp, _vp]),
    "mlmcpi_test_expsin2": (_i, [_u64, _u32, _u32, _vp, _u32, _vp, _vp]),
    "mlmcpi_path_site_updates": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, C.c_int32, _u64, _u32, _u32, _vp]),
    "mlmcpi_lattice_site_updates": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, C.c_int32, _u64, _u32, _u32, _vp]),
    "mlmcpi_schwinger_chit_analytical": (_i, [_d, _u32, _vp]),
    "mlmcpi_schwinger_beta_coarse_nonperturbative": (_i, [_d, _u32, C.c_int32, _vp]),
    "mlmcpi_test_vs_draw": (_i, [_u64, _u32, _u32, _d, _vp, _vp, _u32, _vp, _vp]),
    "mlmcpi_vs_table": (_i, [_d, _vp, _vp]),
}

# functions whose int return value is a status code
_STATUS = {n for n, (r, _) in SIGNATURES.items() if r is _i and n != "mlmcpi_abi_version"}

_lib = None


def load():
    """Load libmlmcpi_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MlmcpiError(f"{LIB_PATH} not found: the HIP extension is not built "
                              "(run `make -C mlmcpathintegral_amd/csrc`); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if _VARIANT and not hasattr(lib, name):
                continue  # an experiment build of older sources (A/B tools only): entry points added since are absent
            f = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def call(name, *args):
    """Call an ABI function; turn a negative status into an exception."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _STATUS and rc != 0:
        raise MlmcpiError(f"{name} failed with status {rc}: {lib.mlmcpi_last_error().decode()}")
    return rc


def path_action(kind, M, T_final, m0=1.0, mu2=1.0, lam=0.0, x0=0.0):
    return PathAction(kind, M, T_final, m0, mu2, lam, x0)


def lattice_action(kind, Mt, Mx, beta=0.0, mass=0.0):
    return LatticeAction(kind, Mt, Mx, beta, mass)


def set_option(name, value=""):
    """tuning knob of the library (never changes results): see mlmcpi_set_option in include/mlmcpi_hip.h"""
    call("mlmcpi_set_option", name.encode(), (value or "").encode())
