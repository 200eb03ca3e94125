"""CPU: the C-ABI library loads, exports every symbol include/mlmcpi_hip.h declares, its host-side
integer index maps are bit-exact against the oracle and against the reference's own Lattice
classes (oracle/_ref), and it fails loudly -- never falls back -- when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mlmcpi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlmcpi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mlmcpathintegral_amd import abi
    lib = abi.load()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"libmlmcpi_hip.so lacks {n}"
    assert set(names) == set(abi.SIGNATURES), set(names) ^ set(abi.SIGNATURES)
    assert lib.mlmcpi_abi_version() == 1


def test_comm_library_exports_every_declared_symbol():
    """include/mlmcpi_comm.h (the RCCL statistics all-reduce): libmlmcpi_rccl.so loads without a GPU and without an RCCL
    runtime in the process (RCCL is opened on first use), and exports what the header declares."""
    from mlmcpathintegral_amd import comm
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mlmcpi_comm.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mlmcpi_comm_[a-z0-9_]+)\s*\(", text)))
    lib = comm.load()
    assert len(names) == 11 and set(names) == set(comm.SIGNATURES)
    for n in names:
        assert hasattr(lib, n), f"libmlmcpi_rccl.so lacks {n}"
    # bad arguments are rejected before anything touches RCCL or a GPU
    h = C.c_void_p()
    assert lib.mlmcpi_comm_init(3, 2, b"x" * 128, 0, C.byref(h)) == -1 and b"rank 3 of 2" in lib.mlmcpi_comm_last_error()


def test_rendezvous_file_of_a_dead_run_is_rejected(tmp_path):
    """a rendezvous file left behind by a run that died must not be accepted by the next run's ranks > 0
    (they would sit in ncclCommInitRank with a mismatched id for ever).  mlmcpi_comm_init_file accepts a record only
    while the process that wrote it is alive; an r02-format file (the bare 128-byte id) is not a record at all.  Both
    are decided before RCCL or a GPU is touched."""
    import struct
    import subprocess
    import sys
    from mlmcpathintegral_amd import comm
    lib = comm.load()
    h = C.c_void_p()
    # (a) a record whose writer is gone: a pid that has just exited, with a made-up start time
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    stale = tmp_path / "stale_id"
    stale.write_bytes(b"MLMCPI1\0" + struct.pack("<qQ", dead.pid, 12345) + b"\x01" * 128)
    rc = lib.mlmcpi_comm_init_file(1, 2, str(stale).encode(), 0, 0.3, C.byref(h))
    assert rc == -1 and b"stale: its writer is gone" in lib.mlmcpi_comm_last_error(), lib.mlmcpi_comm_last_error()
    # (b) same pid as a LIVE process but another start time (pid reuse): still stale
    reused = tmp_path / "reused_id"
    reused.write_bytes(b"MLMCPI1\0" + struct.pack("<qQ", os.getpid(), 1) + b"\x01" * 128)
    rc = lib.mlmcpi_comm_init_file(1, 2, str(reused).encode(), 0, 0.3, C.byref(h))
    assert rc == -1 and b"stale" in lib.mlmcpi_comm_last_error()
    # (c) the old format
    old = tmp_path / "old_id"
    old.write_bytes(b"\x02" * 128)
    rc = lib.mlmcpi_comm_init_file(1, 2, str(old).encode(), 0, 0.3, C.byref(h))
    assert rc == -1 and b"not a rendezvous record" in lib.mlmcpi_comm_last_error()
    # (d) nothing there
    rc = lib.mlmcpi_comm_init_file(1, 2, str(tmp_path / "none").encode(), 0, 0.2, C.byref(h))
    assert rc == -1 and b"no file" in lib.mlmcpi_comm_last_error()


def test_no_silent_cpu_fallback():
    """Without a GPU a compute entry point must return an error, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mlmcpathintegral_amd import abi
    n = C.c_int(-1)
    rc = abi.load().mlmcpi_device_count(C.byref(n))
    assert rc == -4 and n.value == 0
    with pytest.raises(abi.MlmcpiError):
        p = C.c_void_p()
        abi.call("mlmcpi_malloc", C.byref(p), 1024)


@pytest.mark.parametrize("Mt,Mx", [(4, 4), (6, 10), (16, 8), (2, 2)])
def test_index_maps_match_oracle(orc, Mt, Mx):
    from mlmcpathintegral_amd import abi
    lib, L = abi.load(), orc.lib()
    for rot in (0, 1):
        for i in range(-Mt, 2 * Mt):
            for j in range(-Mx, 2 * Mx):
                if rot and (i + j) % 2:
                    continue
                assert lib.mlmcpi_vertex_cart2lin(Mt, Mx, rot, i, j) == L.orc_vertex_cart2lin(Mt, Mx, rot, i, j)
        nv = Mt * Mx // 2 if rot else Mt * Mx
        for ell in range(nv):
            a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            lib.mlmcpi_vertex_lin2cart(Mt, Mx, rot, ell, C.byref(a), C.byref(b))
            L.orc_vertex_lin2cart(Mt, Mx, rot, ell, C.byref(c), C.byref(d))
            assert (a.value, b.value) == (c.value, d.value)
            assert lib.mlmcpi_vertex_cart2lin(Mt, Mx, rot, a.value, b.value) == ell
        mine = np.zeros(nv * 8, dtype=np.uint32)
        theirs = np.zeros(nv * 8, dtype=np.uint32)
        assert lib.mlmcpi_neighbours_2d(Mt, Mx, rot, mine.ctypes.data_as(C.c_void_p)) == 0
        L.orc_neighbours2d(Mt, Mx, rot, theirs)
        assert (mine == theirs).all()
    for i in range(-Mt, 2 * Mt):
        for j in range(-Mx, 2 * Mx):
            for mu in (0, 1):
                ell = lib.mlmcpi_link_cart2lin(Mt, Mx, i, j, mu)
                assert ell == L.orc_link_cart2lin(Mt, Mx, i, j, mu)
                a, b, c = C.c_int(), C.c_int(), C.c_int()
                lib.mlmcpi_link_lin2cart(Mt, Mx, ell, C.byref(a), C.byref(b), C.byref(c))
                assert (a.value, b.value, c.value) == (i % Mt, j % Mx, mu)
    M = Mt * Mx
    mine = np.zeros(2 * M, dtype=np.uint32)
    theirs = np.zeros(2 * M, dtype=np.uint32)
    lib.mlmcpi_neighbours_1d(M, mine.ctypes.data_as(C.c_void_p))
    L.orc_neighbours1d(M, theirs)
    assert (mine == theirs).all()


# ---- against the reference's own compiled Lattice / Statistics classes -----------------------------
REF = os.path.join(ROOT, "oracle", "_ref", "libref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    lib = C.CDLL(REF)
    lib.ref_lattice2d_new.restype = C.c_void_p
    lib.ref_lattice2d_new.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_int]
    lib.ref_lattice2d_coarse.restype = C.c_void_p
    lib.ref_lattice2d_coarse.argtypes = [C.c_void_p]
    for f in ("ref_lattice2d_free",):
        getattr(lib, f).argtypes = [C.c_void_p]
    for f in ("ref_lattice2d_Mt", "ref_lattice2d_Mx", "ref_lattice2d_rotated", "ref_lattice2d_nvertices"):
        getattr(lib, f).argtypes = [C.c_void_p]
        getattr(lib, f).restype = C.c_uint
    lib.ref_vertex_cart2lin.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.ref_vertex_cart2lin.restype = C.c_uint
    lib.ref_link_cart2lin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.ref_link_cart2lin.restype = C.c_uint
    lib.ref_lattice2d_neighbours.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_lattice1d_neighbours.argtypes = [C.c_uint, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
    lib.ref_stats_new.restype = C.c_void_p
    lib.ref_stats_new.argtypes = [C.c_uint]
    lib.ref_stats_free.argtypes = [C.c_void_p]
    lib.ref_stats_record.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lib.ref_stats_reset.argtypes = [C.c_void_p, C.c_int]
    lib.ref_stats_get.argtypes = [C.c_void_p, C.c_void_p]
    return lib


@pytest.mark.parametrize("ctype", [0, 1, 2, 3, 4])
def test_index_maps_match_compiled_reference(ref, ctype):
    """Whole coarsening hierarchy of a 16 x 8 lattice (incl. the rotated levels of CoarsenRotate):
    vertex/link maps and neighbour tables of the ABI == lattice/lattice2d.{hh,cc} compiled."""
    from mlmcpathintegral_amd import abi
    lib = abi.load()
    h = ref.ref_lattice2d_new(16, 8, ctype, 0)
    levels = 0
    while h:
        Mt, Mx, rot = ref.ref_lattice2d_Mt(h), ref.ref_lattice2d_Mx(h), ref.ref_lattice2d_rotated(h)
        nv = ref.ref_lattice2d_nvertices(h)
        theirs = np.zeros(nv * 8, dtype=np.uint32)
        ref.ref_lattice2d_neighbours(h, theirs.ctypes.data_as(C.c_void_p))
        mine = np.zeros(nv * 8, dtype=np.uint32)
        assert lib.mlmcpi_neighbours_2d(Mt, Mx, rot, mine.ctypes.data_as(C.c_void_p)) == 0
        assert (mine == theirs).all(), (ctype, levels)
        for i in range(-2, Mt + 2):
            for j in range(-2, Mx + 2):
                if rot and (i + j) % 2:
                    continue
                assert lib.mlmcpi_vertex_cart2lin(Mt, Mx, rot, i, j) == ref.ref_vertex_cart2lin(h, i, j)
                if not rot:
                    for mu in (0, 1):
                        assert lib.mlmcpi_link_cart2lin(Mt, Mx, i, j, mu) == ref.ref_link_cart2lin(h, i, j, mu)
        nxt = ref.ref_lattice2d_coarse(h)
        ref.ref_lattice2d_free(h)
        h = nxt
        levels += 1
    assert levels >= 2


def test_lattice1d_matches_compiled_reference(ref):
    from mlmcpathintegral_amd import abi
    for M in (2, 7, 128):
        theirs = np.zeros(2 * M, dtype=np.uint32)
        a = C.c_double()
        ref.ref_lattice1d_neighbours(M, 4.0, theirs.ctypes.data_as(C.c_void_p), C.byref(a))
        mine = np.zeros(2 * M, dtype=np.uint32)
        abi.load().mlmcpi_neighbours_1d(M, mine.ctypes.data_as(C.c_void_p))
        assert (mine == theirs).all() and a.value == 4.0 / M


def test_statistics_match_compiled_reference(ref, orc):
    """common/statistics.cc compiled vs the oracle's restatement, bit for bit, on an AR(1) series."""
    rng = np.random.default_rng(11)
    q = np.zeros(5000)
    for k in range(1, q.size):
        q[k] = 0.7 * q[k - 1] + rng.normal()
    q += 2.0
    r = ref.ref_stats_new(20)
    o = orc.Statistics(20)
    for part in (q[:1000], q[1000:]):
        ref.ref_stats_record(r, part.ctypes.data_as(C.c_void_p), part.size)
        o.record(part)
        if part is not q[1000:]:
            ref.ref_stats_reset(r, 0)  # montecarlosinglelevel.cc:27-37: soft reset after burn-in
            o.reset()
    out = np.zeros(6)
    ref.ref_stats_get(r, out.ctypes.data_as(C.c_void_p))
    got = o.get()
    assert list(out) == [got[k] for k in ("average", "variance", "variance_error", "tau_int", "error", "samples")]
    ref.ref_stats_free(r)


@pytest.mark.parametrize("M,T,m0,mu2", [(16, 4.0, 1.0, 1.0), (128, 4.0, 1.0, 1.0), (100, 10.0, 0.7, 2.0)])
def test_ho_cholesky_factor_host(orc, M, T, m0, mu2):
    """mlmcpi_ho_cholesky_factor runs on the host (no GPU): against the oracle's restatement of
    HarmonicOscillatorAction::build_covariance (Gauss-Jordan inverse + Cholesky), against the defining property
    L L^T Q = 1, and against the reference's closed form of <x^2> = (L L^T)_00 (harmonicoscillatoraction.cc:69-74)."""
    import ctypes as C
    from mlmcpathintegral_amd import abi
    act = abi.path_action(abi.HARMONIC, M, T, m0, mu2)
    LT = np.zeros((M, M))
    abi.call("mlmcpi_ho_cholesky_factor", C.byref(act), LT.ctypes.data_as(C.c_void_p))
    L = LT.T
    assert np.allclose(np.triu(L, 1), 0.0)
    A = orc.Action(orc.HARMONIC, M=M, T_final=T, m0=m0, mu2=mu2)
    Lo = np.zeros((M, M))
    assert orc.lib().orc_ho_cholesky(A.h, Lo.reshape(-1)) == 0
    assert np.max(np.abs(L - Lo)) < 1e-12
    a = T / M
    Q = np.zeros((M, M))
    for i in range(M):
        Q[i, i] = a * m0 * mu2 + 2 * m0 / a
        Q[i, (i + 1) % M] += -m0 / a
        Q[i, (i - 1) % M] += -m0 / a
    assert np.max(np.abs(L @ L.T @ Q - np.eye(M))) < 1e-10
    assert abs((L @ L.T)[0, 0] - orc.lib().orc_ho_xsquared_analytical(M, T, m0, mu2)) < 1e-12
    with pytest.raises(abi.MlmcpiError, match="only for the harmonic oscillator"):
        abi.call("mlmcpi_ho_cholesky_factor", C.byref(abi.path_action(abi.ROTOR, M, T, 0.25)), LT.ctypes.data_as(C.c_void_p))


# ---- analytic helpers of the quenched Schwinger model (host code of libmlmcpi_hip.so; no GPU) ------------------------------
def _phi_chit_scipy(beta, n_plaq, nmax=20):
    """common/auxilliary.cc:44-79 + 98-193 restated with scipy (quad for the reference's I'_n, I''_n; ive for I_n)"""
    from scipy import integrate, special
    In = np.array([special.ive(n, beta) for n in range(nmax)])
    dIn = np.array([integrate.quad(lambda p: -1 / (4 * np.pi ** 2) * p * np.exp(beta * (np.cos(p) - 1)) * np.sin(n * p), -np.pi, np.pi,
                                   epsabs=1e-15, epsrel=1e-12, limit=400)[0] for n in range(nmax)])
    ddIn = np.array([integrate.quad(lambda p: 1 / (8 * np.pi ** 3) * p * p * np.exp(beta * (np.cos(p) - 1)) * np.cos(n * p), -np.pi, np.pi,
                                    epsabs=1e-15, epsrel=1e-12, limit=400)[0] for n in range(nmax)])
    w = np.array([(1 + (n > 0)) * (In[n] / In[0]) ** n_plaq for n in range(nmax)])
    ok = w > 0
    return float(np.sum(beta * w[ok] / w.sum() * (ddIn[ok] / In[ok] - (n_plaq - 1) * dIn[ok] ** 2 / In[ok] ** 2)))


@pytest.mark.parametrize("beta,n_plaq,survey", [(1.0, 16, 0.6500978), (4.0, 256, 1.9338785), (1.0, 1024 * 1024, 42610.18),
                                                (8.0, 64, None), (0.3, 36, None), (40.0, 1024, None),
                                                # small P: the weights (I_n / I_0)^P of the high orders are not negligible
                                                # a priori, and I_19(0.5) = 1e-29 I_0 (a quadrature returns noise there)
                                                (0.5, 4, None), (2.0, 4, None), (0.5, 16, None), (1800.0, 4096, None)])
def test_schwinger_chit_analytical(beta, n_plaq, survey):
    """V chi_t = (P / beta) Phi_chi(beta, P): the product's own quadrature against the scipy restatement of the
    reference's formulas and against the values SURVEY 8(c) recorded (analytic side of qoi2dsusceptibility.cc:30-34)."""
    from mlmcpathintegral_amd import abi
    v = C.c_double()
    abi.call("mlmcpi_schwinger_chit_analytical", beta, n_plaq, C.byref(v))
    want = n_plaq / beta * _phi_chit_scipy(beta, n_plaq)
    assert abs(v.value - want) < 1e-9 * max(1.0, abs(want)), (v.value, want)
    if survey is not None:
        assert abs(v.value - survey) < 6e-8 * survey * 10, (v.value, survey)   # the survey printed 7 - 8 digits


@pytest.mark.parametrize("beta,n_plaq,rho", [(8.0, 256, 4), (8.0, 256, 2), (16.0, 1024, 4), (6.0, 64, 4), (30.0, 4096, 4)])
def test_schwinger_beta_coarse_nonperturbative(beta, n_plaq, rho):
    """quenchedschwingerrenormalisation.cc:7-64: x beta with chi_t(x beta, P / rho) = chi_t(beta, P), x in [0.01, 2]"""
    from mlmcpathintegral_amd import abi
    bc, a, b = C.c_double(), C.c_double(), C.c_double()
    abi.call("mlmcpi_schwinger_beta_coarse_nonperturbative", beta, n_plaq, rho, C.byref(bc))
    abi.call("mlmcpi_schwinger_chit_analytical", bc.value, n_plaq // rho, C.byref(a))
    abi.call("mlmcpi_schwinger_chit_analytical", beta, n_plaq, C.byref(b))
    assert 0.01 * beta <= bc.value <= 2.0 * beta
    assert abs(a.value - b.value) < 1e-9 * b.value, (bc.value, a.value, b.value)
    # to O(1 / beta) the matched coupling is the perturbative one (quenchedschwingerrenormalisation.hh:84-104)
    pert = (1.0 / rho) * (1.0 + (1.5 if rho == 4 else 0.5) / beta) * beta
    assert abs(bc.value - pert) < 2.5 / beta * pert, (bc.value, pert)


def test_schwinger_beta_coarse_keeps_the_reference_domain():
    """The bracket [0.01, 2] beta evaluates Phi_chit at 2 beta, which the reference refuses beyond 2000
    (auxilliary.cc:46-52): so does the port, instead of returning a root from the unstable region."""
    from mlmcpathintegral_amd import abi
    bc = C.c_double()
    abi.call("mlmcpi_schwinger_beta_coarse_nonperturbative", 1000.0, 4096, 4, C.byref(bc))
    with pytest.raises(abi.MlmcpiError, match="beta <= 1000"):
        abi.call("mlmcpi_schwinger_beta_coarse_nonperturbative", 1000.5, 4096, 4, C.byref(bc))


def test_tuning_options_are_checked_by_name_and_value():
    """mlmcpi_set_option (host code; the knobs never change results): every documented knob with every documented value is
    accepted, anything else is MLMCPI_ERR_INVALID with a message -- a typo in a benchmark's environment must not be a
    silent no-op at run time."""
    from mlmcpathintegral_amd import abi
    good = {"MLMCPI_SWEEP_TILE": ["64x32x256", "128x64x512", ""], "MLMCPI_OR_KERNEL": ["block", "patch", "lds", "perm", ""],
            "MLMCPI_OR_THREADS": ["256", "512", "1024", ""], "MLMCPI_OR_HEAT": ["fused", "split", "wide", "narrow", ""]}
    for name, values in good.items():
        for v in values:
            abi.set_option(name, v)
    for name, v in (("MLMCPI_OR_HEAT", "sideways"), ("MLMCPI_OR_KERNEL", "blocks"), ("MLMCPI_SWEEP_TILE", "63x32x256"),
                    ("MLMCPI_SWEEP_TILE", "64x32x100"), ("MLMCPI_OR_THREADS", "300"), ("MLMCPI_NO_SUCH_KNOB", "1")):
        with pytest.raises(abi.MlmcpiError) as e:
            abi.set_option(name, v)
        assert "unknown option or value" in str(e.value)
    for name in good:
        abi.set_option(name, "")
