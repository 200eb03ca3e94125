"""The closed form of K overrelaxation sweeps (tests/closed_form.py, the arithmetic of schwinger_perm_kernel) against K sweeps
of the oracle's device-order restatement of quenchedschwingeraction.cc:57-65 -- CPU only."""
import numpy as np
import pytest

from closed_form import angle_diff, rotor_overrelax_closed_form, schwinger_overrelax_closed_form


@pytest.mark.parametrize("Mt,Mx", [(2, 2), (4, 6), (16, 16), (64, 32), (130, 70)])
@pytest.mark.parametrize("K", [1, 2, 5, 10, 13])
def test_closed_form_equals_k_multicolour_sweeps(orc, Mt, Mx, K):
    A = orc.Action(orc.SCHWINGER, Mt=Mt, Mx=Mx, beta=1.0)
    rng = np.random.default_rng(1000 * Mt + K)
    x0 = rng.uniform(-np.pi, np.pi, A.size)
    want = x0.copy()
    for s in range(K):
        A.dev_sweep(want, False, 7, 0, s)   # overrelaxation draws nothing: seed, chain and step do not matter
    got = schwinger_overrelax_closed_form(x0, Mt, Mx, K)
    err = angle_diff(got, want).max()
    assert err <= 4e-15 * (2 * K + 2) * 8, f"{Mt} x {Mx}, K = {K}: {err:.3e}"


def test_sweeps_permute_the_plaquettes(orc):
    """One sweep moves the plaquette at an even row (column) index two rows (columns) down and the one at an odd index two
    up -- the statement the closed form rests on, checked on the oracle's own sweep."""
    Mt, Mx = 12, 8
    A = orc.Action(orc.SCHWINGER, Mt=Mt, Mx=Mx, beta=2.0)
    rng = np.random.default_rng(5)
    x = rng.uniform(-np.pi, np.pi, A.size)

    def plaq(v):
        t = v.reshape(Mx, Mt, 2)
        return t[:, :, 0] + np.roll(t[:, :, 1], -1, axis=1) - np.roll(t[:, :, 0], -1, axis=0) - t[:, :, 1]
    P0 = plaq(x)
    A.dev_sweep(x, False, 7, 0, 0)
    P1 = plaq(x)
    J, I = np.meshgrid(np.arange(Mx), np.arange(Mt), indexing="ij")
    src = P0[(J + 2 * (1 - 2 * (J & 1))) % Mx, (I + 2 * (1 - 2 * (I & 1))) % Mt]
    assert angle_diff(P1, src).max() <= 1e-13


@pytest.mark.parametrize("M", [2, 6, 64, 1000])
@pytest.mark.parametrize("K", [1, 2, 7, 10, 16])
def test_rotor_closed_form_equals_k_even_odd_sweeps(orc, M, K):
    """the 1-D counterpart (rotor_sweep_kernel): K even / odd overrelaxation sweeps of rotoraction.cc:40-56 permute the
    differences of the path"""
    A = orc.Action(orc.ROTOR, M=M, T_final=M / 8.0, m0=0.25)
    rng = np.random.default_rng(M + K)
    x0 = rng.uniform(-np.pi, np.pi, M)
    want = x0.copy()
    for s in range(K):
        A.dev_sweep(want, False, 7, 0, s)
    err = angle_diff(rotor_overrelax_closed_form(x0, K), want).max()
    assert err <= 4e-15 * (2 * K + 2) * 8, f"M = {M}, K = {K}: {err:.3e}"
