"""The two closed-form maps thread -> work of the fused Schwinger launch (round 5), restated in Python and checked for exact
cover: every task / cell of the region is taken by exactly one (wave, lane, slot), none outside it.  The kernels'
results do not depend on which lane computes what (GPU tests: bit-identical states over 134 hashed draws against the
linear hand-out), but a hole or a double in a map would be a silent wrong answer on the device, so the arithmetic of
lattice2d.hip `PermTasks::task` and of the `kMapped` phases of `schwinger_image_heat` is pinned here.
(Reference semantics of what is mapped: quenchedschwingeraction.cc:46-65; the map itself has no counterpart there.)"""
import itertools

import pytest


def perm_tasks(NT, RING, TH):
    """PermTasks<NT, RING, TH>::task for every (wave, lane, slot): yields (mu1, r, c)."""
    OW, HR = 64 + 2 * RING, TH // 2 + RING
    H2, NW = HR // 2, NT // 64
    NS = -(-HR // NW)
    XC = OW - 64
    L0, L1 = XC * H2, (XC // 2) * HR
    NL0, NL1 = -(-L0 // 64), -(-L1 // 64)
    WF = HR - NW * (NS - 1)
    assert NL0 + NL1 <= NW - WF, "the left-over wave-tasks fit the free last slots"
    for wave, lane, k in itertools.product(range(NW), range(64), range(NS)):
        m = wave + NW * k
        if m < H2:
            yield (False, 2 * m, 2 * lane if lane < 32 else 2 * (lane - 32) + 1)
        elif m < HR:
            yield (True, 2 * (m - H2) + (lane >> 5), 2 * (lane & 31))
        elif XC > 0 and k == NS - 1:
            j = wave - WF
            if 0 <= j < NL0:
                L = 64 * j + lane
                if L < L0:
                    yield (False, 2 * (L // XC), 64 + L % XC)
            elif NL0 <= j < NL0 + NL1:
                L = 64 * (j - NL0) + lane
                if L < L1:
                    yield (True, L // (XC // 2), 64 + 2 * (L % (XC // 2)))


@pytest.mark.parametrize("NT,RING,TH", [(512, 2, 64), (1024, 2, 64), (512, 0, 64), (512, 0, 32)])
def test_closed_form_tasks_cover_a_half_exactly_once(NT, RING, TH):
    OW, HR = 64 + 2 * RING, TH // 2 + RING
    want = {(False, r, c) for r in range(0, HR, 2) for c in range(OW)} | {(True, r, c) for r in range(HR) for c in range(0, OW, 2)}
    got = list(perm_tasks(NT, RING, TH))
    assert len(got) == len(set(got)) == len(want) == HR * OW and set(got) == want


def heat_cells(NT, mu, par):
    """Pass 0 and the left-over list of one colour phase of the fused launch (64 x 64 tile, two rings: image 68 wide)."""
    HB, TW, TH, bw = 2, 64, 64, 68
    NW = NT // 64
    NIT = 32 // NW
    main, left = [], []
    for wave, lane in itertools.product(range(NW), range(64)):
        for k in range(NIT):
            if mu == 0:
                r, c = HB + par + 2 * (wave + NW * k), HB - 1 + lane
            else:
                r, c = HB + 2 * (wave + NW * k) + (lane >> 5), HB + par + 2 * (lane & 31)
            main.append(r * bw + c)
    if mu == 0:
        r_first, ncol = HB + par, TW + 2
        nr = (HB + TH - r_first) // 2 + 1
        n_top = (nr - 32) * ncol
        for i in range(n_top + 64):
            ri, ci = (32, i) if i < n_top else ((i - n_top) >> 1, 64 + ((i - n_top) & 1))
            left.append((r_first + 2 * ri) * bw + HB - 1 + ci)
    else:
        c_first = HB + par
        nc = ((HB + TW - 1 if par else HB + TW) - c_first) // 2 + 1
        for i in range((nc - 32) * TH):
            left.append((HB + i) * bw + c_first + 64)
    return main, left


@pytest.mark.parametrize("NT", [512, 1024])
def test_heat_bath_cells_cover_every_colour_phase_exactly_once(NT):
    HB, TW, TH, bw = 2, 64, 64, 68
    for mu, par in itertools.product((0, 1), (0, 1)):
        if mu == 0:   # rows [HB, HB + TH] of one parity, columns [HB - 1, HB + TW]
            want = {r * bw + c for r in range(HB + par, HB + TH + 1, 2) for c in range(HB - 1, HB + TW + 1)}
        else:         # rows [HB, HB + TH), columns of one parity up to HB + TW (even) / HB + TW - 1 (odd)
            want = {r * bw + c for r in range(HB, HB + TH) for c in range(HB + par, (HB + TW - 1 if par else HB + TW) + 1, 2)}
        main, left = heat_cells(NT, mu, par)
        assert len(main) == 2048 and len(set(main + left)) == len(main) + len(left) == len(want)
        assert set(main + left) == want
        assert len(left) == {(0, 0): 130, (0, 1): 64, (1, 0): 64, (1, 1): 0}[(mu, par)]


def test_gff_heat_bath_pairs_cover_colour_zero_exactly_once():
    """gff_or_heat_kernel<5, 64> (512 threads): the colour-0 cells (image rows and columns 1 .. 66, c = r mod 2) by four rounds
    of 'a wave takes two rows x 32 cells' and a fifth of 130 threads; gffaction.cc:33-42 is what a cell computes."""
    HB, bw = 2, 68
    want = {(r, c) for r in range(1, 67) for c in range(1, 67) if (r + c) % 2 == 0}
    got = []
    for tid in range(512):
        wave, lane = tid // 64, tid % 64
        rr = lane >> 5
        r0, c0 = 1 + 2 * wave + rr, 1 + rr + 2 * (lane & 31)
        got += [(r0 + 16 * k, c0) for k in range(4)]
        if tid < 130:
            xr, xci = (65 + tid // 33, tid % 33) if tid < 66 else (1 + (tid - 66), 32)
            got.append((xr, HB - 1 + ((xr + HB - 1) & 1) + 2 * xci))
    assert len(got) == len(set(got)) == len(want) == 66 * 33 and set(got) == want
