// lattice2d.hip -- 2-D lattice kernels: Gaussian free field (vertex field, 5-point stencil) and
// quenched Schwinger model (U(1) link angles, plaquette action), batched over B chains in the
// reference's own SampleState layout (vertex l = Mt*j + i; link l = 2*Mt*j + 2*i + mu, i.e. one
// double2 {theta_0, theta_1} per site, which is exactly a 16-byte-per-lane coalesced load).
//
// Sweeps are overlapped-tile kernels: a workgroup stages its tile plus a halo of 2 sites per fused
// sweep in LDS, runs all colours of all fused sweeps there, and writes only the tile it owns to a
// second buffer.  Halo updates are recomputed by every workgroup that needs them; the counter-based
// RNG makes those recomputations bit-identical.  Per sweep HBM sees ~(1 + halo overhead) reads and
// one write of every entry -- instead of the 4-5 passes of one-kernel-per-colour -- and k fused
// sweeps divide that by k.
#include <algorithm>
#include <type_traits>
#include <mutex>

#include <hipfft/hipfft.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "internal.hpp"

namespace mlmcpi {

// Instrumentation build only (make EXTRA=-DMLMCPI_STAMPS; tools/exp_stamps.py): thread 0 of every workgroup of
// schwinger_or_heat_kernel leaves the 100 MHz wall clock at ten points, plus the XCD / CU it ran on.
#ifdef MLMCPI_STAMPS
__device__ unsigned long long g_stamps[16 * 65536];
#define MLMCPI_STAMP(k)                                                                                            \
  do {                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 65536)                                           \
      g_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();               \
  } while (0)
#define MLMCPI_STAMP_WHERE()                                                                                       \
  do {                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 65536)                                           \
      g_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 15] =                                                  \
          ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32) |         \
          __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));                                        \
  } while (0)
#else
#define MLMCPI_STAMP(k) do { } while (0)
#define MLMCPI_STAMP_WHERE() do { } while (0)
#endif

// linear iteration of a workgroup over an nr x nc region without per-element division
template <int NT, class F>
__device__ __forceinline__ void for_region(uint32_t nr, uint32_t nc, F f) {
  const uint32_t total = nr * nc;
  uint32_t idx = threadIdx.x;
  if (idx >= total) return;
  uint32_t ri = idx / nc, ci = idx - ri * nc;
  uint32_t dr = NT / nc, dc = NT - dr * nc;
  for (; idx < total; idx += NT) {
    f(ri, ci);
    ri += dr;
    ci += dc;
    if (ci >= nc) {
      ci -= nc;
      ++ri;
    }
  }
}

// Staging loop for global -> LDS: the same traversal as for_region, but UNR independent loads are
// issued back to back before any of them is consumed, so a thread has UNR HBM requests in flight
// instead of one (a rolled load -> wait -> ds_write loop is latency bound: ~10 dependent round
// trips per tile).
template <int NT, int UNR, class T, class Load, class Store>
__device__ __forceinline__ void stage_region(uint32_t nr, uint32_t nc, Load load, Store store) {
  const uint32_t total = nr * nc;
  uint32_t idx = threadIdx.x;
  uint32_t ri = idx / nc, ci = idx - ri * nc;
  const uint32_t dr = NT / nc, dc = NT - dr * nc;
  while (idx < total) {
    T v[UNR];
    uint32_t rr[UNR], cc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      rr[u] = ri;
      cc[u] = ci;
      if (idx + u * NT < total) v[u] = load(ri, ci);
      ri += dr;
      ci += dc;
      if (ci >= nc) {
        ci -= nc;
        ++ri;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (idx + u * NT < total) store(rr[u], cc[u], v[u]);
    idx += UNR * NT;
  }
}

// region given as nr x nc with a runtime nc (generic kernels)
template <int NT, int S, bool DIRECT = false, class Setup, class Commit>
__device__ __forceinline__ void heatbath_region(uint32_t nr, uint32_t nc, const RngKey &key, HbPool &pool, Setup setup,
                                                Commit commit) {
  if (DIRECT) {  // nc is a compile-time constant at the call site: the division is a multiply and a shift
    heatbath_cells<NT, S>(nr * nc, key, pool,
                          [&](uint32_t idx, double &tau, double &centre, uint32_t &site, uint32_t &o) {
                            const uint32_t ri = idx / nc;
                            setup(ri, idx - ri * nc, tau, centre, site, o);
                          },
                          commit);
    return;
  }
  // (row, column) of a thread's cells without a division per cell: one division for the first cell, then steps of NT
  // (heatbath_cells asks for a thread's cells in increasing order: idx = tid, tid + NT, tid + 2 NT, ...)
  uint32_t cur = threadIdx.x, ri = cur / nc, ci = cur - ri * nc;
  const uint32_t dr = NT / nc, dc = NT - dr * nc;
  heatbath_cells<NT, S>(nr * nc, key, pool,
                        [&](uint32_t idx, double &tau, double &centre, uint32_t &site, uint32_t &o) {
                          while (cur < idx) {
                            cur += NT;
                            ri += dr;
                            ci += dc;
                            if (ci >= nc) {
                              ci -= nc;
                              ++ri;
                            }
                          }
                          setup(ri, ci, tau, centre, site, o);
                        },
                        commit);
}

// the same for the step-envelope sampler (device_common.hpp): cells are addressed by their LDS offset
// r0 * bw + c0 + (row step) * bw * ri + (column step) * ci
template <int NT, int S, bool DIRECT, class E, class Setup, class KappaExact, class Commit>
__device__ __forceinline__ void heatbath_region_step(uint32_t nr, uint32_t nc, uint32_t origin, uint32_t row_stride,
                                                     uint32_t col_stride, const RngKey &key, VsPool<E> &pool, Setup setup,
                                                     KappaExact kappa_exact, Commit commit) {
  if (DIRECT) {  // nc is a compile-time constant at the call site
    heatbath_cells_step<NT, S, E>(nr * nc, key, pool,
                                  [&](uint32_t idx) {
                                    const uint32_t ri = idx / nc;
                                    return origin + ri * row_stride + (idx - ri * nc) * col_stride;
                                  },
                                  setup, kappa_exact, commit);
    return;
  }
  // runtime nc: cells are handed out by a queue (heatbath_cells_step), in no particular order per thread, so (row, column)
  // come from a division -- by a reciprocal computed once, exact for the indices that occur (idx < 2^16 <= 2^24 / nc)
  const float rcp = 1.0f / (float)nc;
  heatbath_cells_step<NT, S, E>(nr * nc, key, pool,
                                [&](uint32_t idx) {
                                  uint32_t ri = (uint32_t)(((float)idx + 0.5f) * rcp);
                                  int32_t ci = (int32_t)(idx - ri * nc);
                                  if (ci < 0) { --ri; ci += (int32_t)nc; }
                                  else if (ci >= (int32_t)nc) { ++ri; ci -= (int32_t)nc; }
                                  return origin + ri * row_stride + (uint32_t)ci * col_stride;
                                },
                                setup, kappa_exact, commit);
}

struct TileGeom {
  uint32_t TW, TH;      // owned tile extent (even)
  uint32_t tiles_x;     // tiles per row of tiles
};

// The state a launch writes is read next by another launch, from HBM either way (one chain's state is 16 MiB against 4 MiB of
// L2 per XCD): a non-temporal store keeps it from pushing the halos the resident workgroups share out of the L2
// (measured on the one-launch Schwinger draw: -3.5 %).
__device__ __forceinline__ void store_streaming(double2 *p, double x, double y) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const d2_t v = {x, y};
  __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(p));
}

__device__ __forceinline__ uint32_t wrap_add(uint32_t base, uint32_t off, uint32_t n) {
  uint32_t v = base + off;
  while (v >= n) v -= n;
  return v;
}

// ---- Schwinger sweeps ----------------------------------------------------------------------------
// Colour order per sweep: (mu=0, j even), (mu=0, j odd), (mu=1, i even), (mu=1, i odd); links of one
// colour do not appear in each other's staples (quenchedschwingeraction.cc:25-43).  Every link whose
// six staple links lie inside the buffer is updated; the region of exact values shrinks by at most
// two sites per side per sweep, so a halo of 2*nsweeps keeps the owned tile exact (tile origins are
// even, which makes buffer parity equal lattice parity).
// TWC x THC > 0: single-sweep launch on a lattice that the TWC x THC tiles divide and that is wider than a buffer
// (Mt >= TWC + 4, Mx >= THC + 4): tile and buffer extents are compile-time constants (index arithmetic folds, cell
// coordinates come from divisions by constants) and a buffer coordinate wraps around the lattice at most once.  Same
// updates in the same order as the generic instantiation (TWC = THC = 0): bit-identical results.
// STEP: the heat-bath phases draw from the step envelope (2 beta <= kVsKappaMax; device_common.hpp) instead of the
// wrapped-Cauchy one; pool_cap then counts entries of VsPool.
// LDS bytes in front of the tile image of a heat-bath launch of schwinger_sweep_kernel (kernel and host agree through this)
__host__ __device__ inline size_t sweep_pool_bytes(bool step, bool fixed, uint32_t cap) {
  const size_t b = step ? (fixed ? VsPool<uint16_t>::bytes(cap) : VsPool<uint32_t>::bytes(cap)) : HbPool::bytes(cap);
  return (b + 15) / 16 * 16;
}

template <bool HEAT, int NT, int TWC = 0, int THC = 0, bool STEP = false>
__global__ void __launch_bounds__(NT, HEAT ? (NT == 256 ? 4 : NT == 512 ? 2 : 1) : 1)
    schwinger_sweep_kernel(uint32_t Mt, uint32_t Mx, double beta, const double2 *__restrict__ in,
                           double2 *__restrict__ out, TileGeom tg, uint32_t nsweeps_arg, uint32_t kinds, RngKey key0,
                           uint32_t pool_cap, int qoi_op, double *__restrict__ qoi_partial, const uint32_t *__restrict__ vs_table) {
  extern __shared__ double lds[];
  __shared__ double qoi_red[NT / kWave];
  constexpr bool FIXED = TWC > 0;
  const uint32_t nsweeps = FIXED ? 1u : nsweeps_arg;
  const uint32_t H = 2 * nsweeps;
  const uint32_t tile = blockIdx.x, b = blockIdx.y;
  const uint32_t ty = tile / tg.tiles_x, tx = tile - ty * tg.tiles_x;
  const uint32_t i0 = tx * (FIXED ? TWC : tg.TW), j0 = ty * (FIXED ? THC : tg.TH);
  const uint32_t ow = FIXED ? TWC : min(tg.TW, Mt - i0), oh = FIXED ? THC : min(tg.TH, Mx - j0);
  const uint32_t bw = ow + 2 * H, bh = oh + 2 * H;
  // lattice coordinate of a buffer coordinate
  auto wrap = [&](uint32_t base, uint32_t off, uint32_t n) {
    if (FIXED) {
      const uint32_t v = base + off;
      return v >= n ? v - n : v;
    }
    return wrap_add(base, off, n);
  };
  // The sampler's tables and the list / pool of open cells of the heat-bath phases at the START of the LDS (table look-ups
  // are then instruction offsets; r04), the tile image behind them (launch_sweep_nt sizes the allocation the same way)
  using PoolEntry = typename std::conditional<FIXED, uint16_t, uint32_t>::type;   // 68 x 36 cells: 12 bits of offset
  double *pool_lds = lds;
  HbPool pool = HbPool::carve(pool_lds, HEAT && !STEP ? pool_cap : 0u);
  VsPool<PoolEntry> vpool = VsPool<PoolEntry>::carve(pool_lds, HEAT && STEP ? pool_cap : 0u, STEP ? vs_table : nullptr);
  double *th0 = lds + (HEAT ? sweep_pool_bytes(STEP, FIXED, pool_cap) / sizeof(double) : 0), *th1 = th0 + (size_t)bw * bh;
  const double beta2 = 2. * beta;
  const uint32_t sc = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt)) % Mt);  // lattice column of buffer column 0
  const uint32_t sr = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx)) % Mx);
  const double2 *src = in + (size_t)b * Mt * Mx;
  RngKey key = key0;
  key.chain += b;

  stage_region<NT, (NT >= 1024 ? 3 : 5), double2>(
      bh, bw, [&](uint32_t r, uint32_t c) { return src[(size_t)wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)]; },
      [&](uint32_t r, uint32_t c, double2 v) {
        th0[r * bw + c] = v.x;
        th1[r * bw + c] = v.y;
      });
  __syncthreads();

  for (uint32_t s = 0; s < nsweeps; ++s) {
    const bool heat = HEAT && (FIXED || ((kinds >> s) & 1u));
    RngKey skey = key;
    skey.step += s;
    // Update regions.  In general every link whose staple links lie inside the buffer is updated:
    //   mu = 0: rows [1, bh-2] of one parity, columns [0, bw-2];   mu = 1: columns [1, bw-2] of one parity, rows [0, bh-2].
    // The LAST sweep of a launch only has to be right on the owned tile (rows [H, H+oh), columns [H, H+ow)), so each
    // of its phases is cut down to what later phases still read:
    //   phase 3 (mu = 1, odd columns)   owned links;
    //   phase 2 (mu = 1, even columns)  + the column right of the tile (phase 3 reads theta_1(i+1, j));
    //   phases 0, 1 (mu = 0)            rows [H, H+oh], columns [H-1, H+ow] (phases 2, 3 read theta_0 at (i-1 .. i, j .. j+1)).
    // For a heat-bath launch (always a single sweep) that is 7 % fewer draws.
    const bool last = s + 1 == nsweeps;
    const uint32_t r_hi0 = last ? H + oh : bh - 2;                              // mu = 0 rows: upper end (inclusive)
    const uint32_t c_lo0 = last ? H - 1 : 0, c_hi0 = last ? H + ow : bw - 2;    // mu = 0 columns
    const uint32_t r_lo1 = last ? H : 0, r_hi1 = last ? H + oh - 1 : bh - 2;    // mu = 1 rows
    for (uint32_t par = 0; par < 2; ++par) {
      // even rows first; H is even, so H + par has the parity of this phase
      const uint32_t r_first = last ? H + par : (par ? 1 : 2);
      const uint32_t nr = r_first <= r_hi0 ? (r_hi0 - r_first) / 2 + 1 : 0;
      const uint32_t ncol = c_hi0 - c_lo0 + 1;
      if (heat && STEP) {
        heatbath_region_step<NT, 5, FIXED, PoolEntry>(
            nr, ncol, r_first * bw + c_lo0, 2 * bw, 1, skey, vpool,
            [&](uint32_t o, VsCell &cell) {
              const uint32_t r = o / bw, c = o - r * bw;
              vs_cell(beta2, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw], cell);
              cell.site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt));
            },
            [&](uint32_t o) {
              return vs_kappa_exact(beta2, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw]);
            },
            [&](uint32_t o, double v) { th0[o] = v; });
      } else if (heat) {
        heatbath_region<NT, 5, FIXED>(
            nr, ncol, skey, pool,
            [&](uint32_t ri, uint32_t ci, double &tau, double &centre, uint32_t &site, uint32_t &o) {
              const uint32_t r = r_first + 2 * ri, c = c_lo0 + ci;
              o = r * bw + c;
              const double tp = th0[o + bw] + th1[o] - th1[o + 1];  // staple angles, unwrapped (expcos_params)
              const double tm = th0[o - bw] + th1[o - bw + 1] - th1[o - bw];
              expcos_params(beta, tp, tm, tau, centre);
              site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt));
            },
            [&](uint32_t o, double v) { th0[o] = v; });
      } else
      for_region<NT>(nr, ncol, [&](uint32_t ri, uint32_t ci) {
        const uint32_t r = r_first + 2 * ri, o = r * bw + c_lo0 + ci;
        // overrelaxation: mod_2pi(theta+ + theta- - theta); the two staple angles need no wrap of their own here
        // (2 pi-periodicity of the final map), which drops two of the three mod_2pi per update
        const double tp = th0[o + bw] + th1[o] - th1[o + 1];
        const double tm = th0[o - bw] + th1[o - bw + 1] - th1[o - bw];
        th0[o] = mod_2pi_fast((tp + tm) - th0[o]);
      });
      __syncthreads();
    }
    for (uint32_t par = 0; par < 2; ++par) {
      const uint32_t c_first = last ? H + par : (par ? 1 : 2);
      // last sweep: even columns up to H + ow (one past the tile), odd columns up to H + ow - 1
      const uint32_t c_hi1 = last ? (par ? H + ow - 1 : H + ow) : bw - 2;
      const uint32_t nc = c_first <= c_hi1 ? (c_hi1 - c_first) / 2 + 1 : 0;
      const uint32_t nrow = r_hi1 - r_lo1 + 1;
      if (heat && STEP) {
        heatbath_region_step<NT, 5, FIXED, PoolEntry>(
            nrow, nc, r_lo1 * bw + c_first, bw, 2, skey, vpool,
            [&](uint32_t o, VsCell &cell) {
              const uint32_t r = o / bw, c = o - r * bw;
              vs_cell(beta2, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1], cell);
              cell.site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)) + 1;
            },
            [&](uint32_t o) {
              return vs_kappa_exact(beta2, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1]);
            },
            [&](uint32_t o, double v) { th1[o] = v; });
      } else if (heat) {
        heatbath_region<NT, 5, FIXED>(
            nrow, nc, skey, pool,
            [&](uint32_t ri, uint32_t ci, double &tau, double &centre, uint32_t &site, uint32_t &o) {
              const uint32_t r = r_lo1 + ri, c = c_first + 2 * ci;
              o = r * bw + c;
              const double tp = th0[o] + th1[o + 1] - th0[o + bw];
              const double tm = th0[o + bw - 1] + th1[o - 1] - th0[o - 1];
              expcos_params(beta, tp, tm, tau, centre);
              site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)) + 1;
            },
            [&](uint32_t o, double v) { th1[o] = v; });
      } else
      for_region<NT>(nrow, nc, [&](uint32_t ri, uint32_t ci) {
        const uint32_t r = r_lo1 + ri, c = c_first + 2 * ci, o = r * bw + c;
        const double tp = th0[o] + th1[o + 1] - th0[o + bw];
        const double tm = th0[o + bw - 1] + th1[o - 1] - th0[o - 1];
        th1[o] = mod_2pi_fast((tp + tm) - th1[o]);
      });
      __syncthreads();
    }
  }

  // Optional fused QoI of the final state (qoi/qft/qoiavgplaquette.cc:8-27, qoi2dsusceptibility.cc:8-27): the plaquette
  // (i, j) needs theta(i+1, j, 1) and theta(i, j+1, 0); for the owned tile those are the column right of it and the row
  // above it, which the last sweep's pruned regions bring to their final values (that is what they are for).  One
  // partial per tile; lattice_finish_kernel sums them in tile order.
  double acc[1] = {0.0};
  double2 *dst = out + (size_t)b * Mt * Mx;
  for_region<NT>(oh, ow, [&](uint32_t r, uint32_t c) {
    const uint32_t o = (r + H) * bw + (c + H);
    dst[(size_t)(j0 + r) * Mt + (i0 + c)] = make_double2(th0[o], th1[o]);
    if (qoi_op) {
      const double thp = th0[o] + th1[o + 1] - th0[o + bw] - th1[o];   // quenchedschwingeraction.cc:14-17
      acc[0] += qoi_op == 3 ? cos_reduced(thp) : mod_2pi(thp);         // 3 = L_PLAQ, 4 = L_CHARGE
    }
  });
  if (qoi_op) {
    block_sum<1>(acc, qoi_red);
    if (threadIdx.x == 0) qoi_partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
  }
}

// ---- Schwinger overrelaxation, specialised ---------------------------------------------------------------
// Same sweep (same colour order, same update regions, therefore bit-identical results) as
// schwinger_sweep_kernel<false, 256>, for launches of K overrelaxation sweeps on lattices that the
// TW x TH tiles divide.  Everything the generic kernel recomputes per update is hoisted:
//   * tile geometry is compile time, so every neighbour is an immediate offset of one LDS address;
//   * each thread's cells (linear index tid + 256 m) are the same in every sweep, so their LDS
//     offsets are computed once and kept in registers (M0 + M1 VGPRs);
//   * LDS rows are padded to an odd number of doubles and the mu = 1 phases run with lanes along
//     rows, so column-parity phases are bank-conflict free instead of stride-2.
// Per update that leaves the ~10 fp64 instructions of the update itself (6 adds, 1 mod_2pi: the staple angles are
// not wrapped separately, the final mod_2pi takes care of it).
template <int TW, int TH, int K, int NT>
__global__ void __launch_bounds__(NT)
    schwinger_or_kernel(uint32_t Mt, uint32_t Mx, const double2 *__restrict__ in, double2 *__restrict__ out,
                        uint32_t tiles_x) {
  constexpr int H = 2 * K, BW = TW + 2 * H, BH = TH + 2 * H, P = BW + 1;
  constexpr int NR0 = (BH - 2) / 2, NC0 = BW - 1, T0 = NR0 * NC0, M0 = (T0 + NT - 1) / NT;
  constexpr int NCI = (BW - 2) / 2, NR1 = BH - 1, T1 = NCI * NR1, M1 = (T1 + NT - 1) / NT;
  extern __shared__ double lds[];
  double *th0 = lds, *th1 = lds + BH * P;
  // LDS byte address of lds[0] for the ds_read_b64 issued through inline asm (the low word of a flat LDS address is the
  // LDS offset): 0 today, but static LDS added to this kernel would move it
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  const uint32_t sc = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt)) % Mt);
  const uint32_t sr = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx)) % Mx);
  const double2 *src = in + (size_t)b * Mt * Mx;

  stage_region<NT, 5, double2>(
      BH, BW, [&](uint32_t r, uint32_t c) { return src[(size_t)wrap_add(sr, r, Mx) * Mt + wrap_add(sc, c, Mt)]; },
      [&](uint32_t r, uint32_t c, double2 v) {
        th0[r * P + c] = v.x;
        th1[r * P + c] = v.y;
      });

  // cell offsets: mu = 0, even rows (odd rows: - P); mu = 1, even columns (odd columns: - 1)
  int o0[M0], o1[M1];
#pragma unroll
  for (int m = 0; m < M0; ++m) {
    const int idx = tid + NT * m, ri = idx / NC0, c = idx - ri * NC0;
    o0[m] = (2 + 2 * ri) * P + c;
  }
#pragma unroll
  for (int m = 0; m < M1; ++m) {
    const int idx = tid + NT * m, ci = idx / NR1, r = idx - ci * NR1;
    o1[m] = r * P + 2 + 2 * ci;
  }
  __syncthreads();

  for (int s = 0; s < K; ++s) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
#pragma unroll
      for (int m = 0; m < M0; ++m) {
        if ((m + 1) * NT <= T0 || (int)tid + NT * m < T0) {
          const int o = o0[m] - par * P;
          const uint32_t a0 = lds0 + (uint32_t)(o - P) * 8u, a1 = a0 + (uint32_t)(BH * P) * 8u;  // th0 / th1 at (r-1, c)
          double t0_up = lds_read_f64<2 * P * 8>(a0), t0_dn = lds_read_f64<0>(a0), t0_own = lds_read_f64<P * 8>(a0);
          double t1_c = lds_read_f64<P * 8>(a1), t1_r = lds_read_f64<P * 8 + 8>(a1);
          double t1_dc = lds_read_f64<0>(a1), t1_dr = lds_read_f64<8>(a1);
          lds_wait7(t0_up, t0_dn, t0_own, t1_c, t1_r, t1_dc, t1_dr);
          const double tp = t0_up + t1_c - t1_r;  // staple angles unwrapped: see the generic kernel
          const double tm = t0_dn + t1_dr - t1_dc;
          th0[o] = mod_2pi_fast((tp + tm) - t0_own);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
#pragma unroll
      for (int m = 0; m < M1; ++m) {
        if ((m + 1) * NT <= T1 || (int)tid + NT * m < T1) {
          const int o = o1[m] - par;
          const uint32_t a0 = lds0 + (uint32_t)(o - 1) * 8u, a1 = a0 + (uint32_t)(BH * P) * 8u;  // th0 / th1 at (r, c-1)
          double t0_c = lds_read_f64<8>(a0), t0_u = lds_read_f64<P * 8 + 8>(a0);
          double t0_lu = lds_read_f64<P * 8>(a0), t0_l = lds_read_f64<0>(a0);
          double t1_r = lds_read_f64<16>(a1), t1_l = lds_read_f64<0>(a1), t1_own = lds_read_f64<8>(a1);
          lds_wait7(t0_c, t0_u, t0_lu, t0_l, t1_r, t1_l, t1_own);
          const double tp = t0_c + t1_r - t0_u;
          const double tm = t0_lu + t1_l - t0_l;
          th1[o] = mod_2pi_fast((tp + tm) - t1_own);
        }
      }
      __syncthreads();
    }
  }

  double2 *dst = out + (size_t)b * Mt * Mx;
  for_region<NT>(TH, TW, [&](uint32_t r, uint32_t c) {
    const uint32_t o = (r + H) * P + (c + H);
    dst[(size_t)(j0 + r) * Mt + (i0 + c)] = make_double2(th0[o], th1[o]);
  });
}

// ---- Schwinger overrelaxation, register-tiled ---------------------------------------------------------------
// The LDS-resident kernel above reads 7 doubles from LDS and writes 1 per link update (64 B): at 128 B/clk per CU that
// is the bound it runs into (16 waves x 64 LDS instructions x 4 clk per sweep and tile ~ the measured 0.43 ms per
// 4-sweep launch), not HBM and not the VALU.  Here every thread keeps a 2 x 2 block of vertices -- 8 link angles -- in
// registers for all K sweeps; LDS only carries what crosses block boundaries: per sweep a thread reads 16 neighbour
// values (6, 3, 5, 2 in the four colour phases; values that cannot have changed in between are reused) and publishes
// its 8 updated links, 24 B per update instead of 64.  The LDS image is eight structure-of-arrays planes
// [mu][row parity][column parity] over blocks, so that consecutive lanes (consecutive blocks of a row) touch
// consecutive doubles in every access.  Same updates, same order and the same arithmetic as the kernels above (every
// link whose six staple links lie inside the buffer is updated), hence bit-identical results.
// One block per thread: (64 + 4K)/2 x (32 + 4K)/2 <= 1024 blocks for K <= 4.
template <int K>
__global__ void __launch_bounds__(1024)
    schwinger_or_patch_kernel(uint32_t Mt, uint32_t Mx, const double2 *__restrict__ in, double2 *__restrict__ out,
                              uint32_t tiles_x) {
  constexpr int TW = 64, TH = 32, H = 2 * K, BW = TW + 2 * H, BH = TH + 2 * H, NPX = BW / 2, NPY = BH / 2;
  constexpr int NP = NPX * NPY;  // blocks per tile = active threads
  static_assert(NP <= 1024, "one 2 x 2 block per thread");
  extern __shared__ double lds[];
  // plane(mu, c, a)[pj][pi]: link mu of vertex (2 pi + a, 2 pj + c)
  auto plane = [&](int mu, int c, int a) { return lds + ((mu * 2 + c) * 2 + a) * NP; };
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  const bool active = tid < NP;
  const int pj = active ? (int)tid / NPX : 0, pi = active ? (int)tid - pj * NPX : 0;
  const double2 *src = in + (size_t)b * Mt * Mx;
  // global coordinates of the block's lower-left vertex (even, so (gi, gi + 1) never straddles the wrap)
  const uint32_t gi = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt) + 2 * pi) % Mt);
  const uint32_t gj0 = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx) + 2 * pj) % Mx);
  const uint32_t gj1 = gj0 + 1 == Mx ? 0 : gj0 + 1;
  double t0[2][2] = {{0, 0}, {0, 0}}, t1[2][2] = {{0, 0}, {0, 0}};  // [c][a]
  if (active) {
    const double2 v00 = src[(size_t)gj0 * Mt + gi], v01 = src[(size_t)gj0 * Mt + gi + 1];
    const double2 v10 = src[(size_t)gj1 * Mt + gi], v11 = src[(size_t)gj1 * Mt + gi + 1];
    t0[0][0] = v00.x; t1[0][0] = v00.y; t0[0][1] = v01.x; t1[0][1] = v01.y;
    t0[1][0] = v10.x; t1[1][0] = v10.y; t0[1][1] = v11.x; t1[1][1] = v11.y;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        plane(0, c, a)[tid] = t0[c][a];
        plane(1, c, a)[tid] = t1[c][a];
      }
  }
  // Neighbour blocks, clamped into the buffer.  Links on the buffer edge have no complete stencil; the kernels above
  // leave them alone, here they are updated from the clamped (wrong) neighbours instead -- no predicates in the sweep
  // loop.  Either way what an edge link holds only ever reaches links that are already outside the exact region
  // (which shrinks by one block per sweep no matter what its surroundings hold), never the owned tile.
  const int me = active ? (int)tid : 0;  // idle threads of the last wave: every index is entry 0, nothing is written
  const int dn = pj > 0 ? me - NPX : me, up = pj + 1 < NPY ? me + NPX : me;
  const int lf = pi > 0 ? me - 1 : me, rt = pi + 1 < NPX ? me + 1 : me;
  const int rtdn = (pi + 1 < NPX ? 1 : 0) + (pj > 0 ? -NPX : 0) + me;
  const int lfup = (pi > 0 ? -1 : 0) + (pj + 1 < NPY ? NPX : 0) + me;
  __syncthreads();

  for (int s = 0; s < K; ++s) {
    // phase 0: mu = 0, even rows (c = 0)
    // (the idle threads of the last wave read entry 0 and compute on zeros: no predicate, no zero fill in the loop)
    const double D0 = plane(0, 1, 0)[dn], D1 = plane(0, 1, 1)[dn], E0 = plane(1, 1, 0)[dn], E1 = plane(1, 1, 1)[dn];
    double R0 = plane(1, 0, 0)[rt];
    const double RD = plane(1, 1, 0)[rtdn];
    {
      {  // a = 0
        const double tp = t0[1][0] + t1[0][0] - t1[0][1];
        const double tm = D0 + E1 - E0;
        t0[0][0] = mod_2pi_fast((tp + tm) - t0[0][0]);
        if (active) plane(0, 0, 0)[me] = t0[0][0];
      }
      {  // a = 1
        const double tp = t0[1][1] + t1[0][1] - R0;
        const double tm = D1 + RD - E1;
        t0[0][1] = mod_2pi_fast((tp + tm) - t0[0][1]);
        if (active) plane(0, 0, 1)[me] = t0[0][1];
      }
    }
    __syncthreads();
    // phase 1: mu = 0, odd rows (c = 1)
    const double U0 = plane(0, 0, 0)[up], U1 = plane(0, 0, 1)[up];
    double R1 = plane(1, 1, 0)[rt];
    {
      {
        const double tp = U0 + t1[1][0] - t1[1][1];
        const double tm = t0[0][0] + t1[0][1] - t1[0][0];
        t0[1][0] = mod_2pi_fast((tp + tm) - t0[1][0]);
        if (active) plane(0, 1, 0)[me] = t0[1][0];
      }
      {
        const double tp = U1 + t1[1][1] - R1;
        const double tm = t0[0][1] + R0 - t1[0][1];
        t0[1][1] = mod_2pi_fast((tp + tm) - t0[1][1]);
        if (active) plane(0, 1, 1)[me] = t0[1][1];
      }
    }
    __syncthreads();
    // phase 2: mu = 1, even columns (a = 0)
    const double L01 = plane(0, 0, 1)[lf], L11 = plane(0, 1, 1)[lf], M01 = plane(1, 0, 1)[lf], M11 = plane(1, 1, 1)[lf];
    const double LU = plane(0, 0, 1)[lfup];
    {
      {  // c = 0
        const double tp = t0[0][0] + t1[0][1] - t0[1][0];
        const double tm = L11 + M01 - L01;
        t1[0][0] = mod_2pi_fast((tp + tm) - t1[0][0]);
        if (active) plane(1, 0, 0)[me] = t1[0][0];
      }
      {  // c = 1
        const double tp = t0[1][0] + t1[1][1] - U0;
        const double tm = LU + M11 - L11;
        t1[1][0] = mod_2pi_fast((tp + tm) - t1[1][0]);
        if (active) plane(1, 1, 0)[me] = t1[1][0];
      }
    }
    __syncthreads();
    // phase 3: mu = 1, odd columns (a = 1); the right neighbour's mu = 1 links changed in phase 2
    R0 = plane(1, 0, 0)[rt];
    R1 = plane(1, 1, 0)[rt];
    {
      {  // c = 0
        const double tp = t0[0][1] + R0 - t0[1][1];
        const double tm = t0[1][0] + t1[0][0] - t0[0][0];
        t1[0][1] = mod_2pi_fast((tp + tm) - t1[0][1]);
        if (active) plane(1, 0, 1)[me] = t1[0][1];
      }
      {  // c = 1
        const double tp = t0[1][1] + R1 - U1;
        const double tm = U0 + t1[1][0] - t0[1][0];
        t1[1][1] = mod_2pi_fast((tp + tm) - t1[1][1]);
        if (active) plane(1, 1, 1)[me] = t1[1][1];
      }
    }
    __syncthreads();
  }

  // owned vertices: buffer columns [H, H + TW), rows [H, H + TH); H is even, so a block is owned as a whole
  if (active && pi >= H / 2 && pi < (H + TW) / 2 && pj >= H / 2 && pj < (H + TH) / 2) {
    double2 *dst = out + (size_t)b * Mt * Mx;
    const size_t o0 = (size_t)(j0 + 2 * pj - H) * Mt + (i0 + 2 * pi - H);
    dst[o0] = make_double2(t0[0][0], t1[0][0]);
    dst[o0 + 1] = make_double2(t0[0][1], t1[0][1]);
    dst[o0 + Mt] = make_double2(t0[1][0], t1[1][0]);
    dst[o0 + Mt + 1] = make_double2(t0[1][1], t1[1][1]);
  }
}

// Experiment (-DMLMCPI_SKEW=n): the two workgroups a CU holds start together and stay in step -- both in their load / store
// phases (HBM bound, issue slots idle), then both in their sweeps (issue bound, HBM idle).  Delaying the second workgroup
// of every CU by n x 3.6 us at the start of the launch puts them half a period apart.
#ifdef MLMCPI_SKEW
#define MLMCPI_SKEW_START()                                                                      \
  do {                                                                                           \
    const uint32_t lin_ = blockIdx.y * gridDim.x + blockIdx.x;                                   \
    if (lin_ >= kComputeUnits && lin_ < 2 * kComputeUnits)                                       \
      for (int i_ = 0; i_ < MLMCPI_SKEW; ++i_) __builtin_amdgcn_s_sleep(127);                    \
  } while (0)
#else
#define MLMCPI_SKEW_START() do { } while (0)
#endif

// ---- Schwinger overrelaxation, 4 x 4 register blocks on 64 x 64 tiles ------------------------------------------
// The 2 x 2 kernel above recomputes (64 + 4K)(32 + 4K) / (64 * 32) = 1.875 x the owned updates at K = 4 and moves 24 B
// through LDS per update.  Here a thread keeps a 4 x 4 block of vertices (32 link angles, 64 VGPRs) for all K sweeps and
// the tile is 64 x 64: redundancy (64 + 4K)^2 / 64^2 = 1.56 at K = 4 (1.72 at K = 5, which the 1024-thread limit of the
// 2 x 2 kernel could not reach), and LDS holds only the 20 values per block that a neighbouring block reads:
//   TOP0[a], TOP1[a]   both links of the top row        (read by the block above as its row -1)
//   BOT0[a]            mu = 0 links of the bottom row   (row PH of the block below)
//   LEFT1[c]           mu = 1 links of the left column  (column PW of the block to the left)
//   RIGHT0[c], RIGHT1[c]  both links of the right column (column -1 of the block to the right)
// with the four corner values that belong to two of these lists stored once.  Per sweep a thread reads 32 and writes
// 20 doubles for its 32 updates (13 B per update).  Planes are [value][block], so consecutive lanes touch consecutive
// doubles.  Same updates in the same colour order with the same arithmetic as every other overrelaxation kernel here:
// bit-identical results.  Block edges of the buffer read clamped neighbours, as in the 2 x 2 kernel: what they compute
// is wrong, and never reaches the owned tile (the exact region shrinks by 2 sites per sweep from a halo of 2K).
//
// Measured on MI355X (1024 x 1024, 32 chains; tools/exp_or_block.py, timestamps taken inside the kernel): a sweep costs
// 0.028 ms of the launch, which is the fp64 issue time of its 9 instructions per update, and the rest of the launch
// (0.21 ms at K = 1, against 0.17 ms for a plain copy of the state) is the load and store phase of the workgroups,
// which the two workgroups a CU holds overlap only partly with each other's sweeps.  Persistent workgroups and an
// XCD-aware tile order changed nothing; writing the tile back in whole 1 KiB rows per wave instruction instead of
// 16 B per lane at a 64 B stride took 0.02-0.035 ms off every launch (see the end of the kernel); doing the same for
// the loads did not pay.  The K >= 4 launches run at the package power limit (1.37 kW, sclk 2.17-2.25 GHz).
template <int K>
struct OrBlockGeom {
  static constexpr int TW = 64, TH = 64, PW = 4, PH = 4, H = 2 * K;
  static constexpr int BW = TW + 2 * H, BH = TH + 2 * H, NPX = BW / PW, NPY = BH / PH, NP = NPX * NPY;
  static constexpr int NT = (NP + 63) / 64 * 64;
  static constexpr int NPLANE = 3 * PW + 3 * PH - 4;
  static constexpr size_t lds_bytes = (size_t)NPLANE * NP * sizeof(double);
  // plane numbers (corner values stored once)
  static constexpr int top0(int a) { return a; }
  static constexpr int top1(int a) { return PW + a; }
  static constexpr int right0(int c) { return c == PH - 1 ? top0(PW - 1) : 3 * PW - 1 + (PH - 1) + c; }
  static constexpr int right1(int c) { return c == PH - 1 ? top1(PW - 1) : 3 * PW - 1 + 2 * (PH - 1) + c; }
  static constexpr int bot0(int a) { return a == PW - 1 ? right0(0) : 2 * PW + a; }
  static constexpr int left1(int c) { return c == PH - 1 ? top1(0) : 3 * PW - 1 + c; }
};

// The buffer of geometry G (tile + halo G::H) into 4 x 4 register blocks, then KS <= G::H / 2 overrelaxation sweeps on it.
// Ends behind the barrier of the last colour phase: the plane area of the LDS is dead from there on.
template <class G, int KS>
__device__ __forceinline__ void or_block_sweeps(double *lds, const double2 *__restrict__ src, uint32_t Mt, uint32_t Mx,
                                                uint32_t i0, uint32_t j0, double (&t0)[G::PH][G::PW], double (&t1)[G::PH][G::PW]) {
  constexpr int PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NPY = G::NPY, NP = G::NP;
  static_assert(2 * KS <= H, "a sweep costs two sites of halo");
  auto pl = [&](int p) { return lds + p * NP; };
  const uint32_t tid = threadIdx.x;
  if (tid >= (uint32_t)G::NT) {  // waves beyond the blocks (a caller with a wider workgroup): only the barriers
    for (int i = 0; i < 1 + 4 * KS; ++i) __syncthreads();
    return;
  }
  const bool active = tid < NP;
  const int pj = active ? (int)tid / NPX : 0, pi = active ? (int)tid - pj * NPX : 0;
  const int me = active ? (int)tid : 0;  // idle threads of the last wave: every index is entry 0, nothing is written
  // neighbour blocks, clamped into the buffer
  const int dn = pj > 0 ? me - NPX : me, up = pj + 1 < NPY ? me + NPX : me;
  const int lf = pi > 0 ? me - 1 : me, rt = pi + 1 < NPX ? me + 1 : me;
  const int rtdn = (pi + 1 < NPX ? 1 : 0) + (pj > 0 ? -NPX : 0) + me;
  const int lfup = (pi > 0 ? -1 : 0) + (pj + 1 < NPY ? NPX : 0) + me;
  // t0, t1: [c][a] = links of vertex (PW pi + a, PH pj + c)

  // the block's columns in the lattice: H is even, so (gi, gi + 1) never straddles the wrap, (gi + 1, gi + 2) may
  {
    uint32_t gi[PW / 2], gj[PH];
    gi[0] = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt) + PW * pi) % Mt);
    gj[0] = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx) + PH * pj) % Mx);
#pragma unroll
    for (int a = 1; a < PW / 2; ++a) gi[a] = gi[a - 1] + 2 == Mt ? 0 : gi[a - 1] + 2;
#pragma unroll
    for (int c = 1; c < PH; ++c) gj[c] = gj[c - 1] + 1 == Mx ? 0 : gj[c - 1] + 1;
#pragma unroll
    for (int c = 0; c < PH; ++c)
#pragma unroll
      for (int a = 0; a < PW; a += 2) {
        double2 v0 = make_double2(0, 0), v1 = v0;
        if (active) {
          v0 = src[(size_t)gj[c] * Mt + gi[a / 2]];
          v1 = src[(size_t)gj[c] * Mt + gi[a / 2] + 1];
        }
        t0[c][a] = v0.x; t1[c][a] = v0.y; t0[c][a + 1] = v1.x; t1[c][a + 1] = v1.y;
      }
  }
  MLMCPI_STAMP(1);  // (the loads are issued; the first publish waits for their values)
  // what a neighbour reads of link mu at (a, c): up to three lists, a corner value once
  auto publish = [&](int mu, int a, int c, double v) {
    const int p1 = c == PH - 1 ? (mu ? G::top1(a) : G::top0(a)) : -1;
    const int p2 = mu == 0 ? (c == 0 ? G::bot0(a) : -1) : (a == 0 ? G::left1(c) : -1);
    const int p3 = a == PW - 1 ? (mu ? G::right1(c) : G::right0(c)) : -1;
    if (!active) return;
    if (p1 >= 0) pl(p1)[me] = v;
    if (p2 >= 0 && p2 != p1) pl(p2)[me] = v;
    if (p3 >= 0 && p3 != p1 && p3 != p2) pl(p3)[me] = v;
  };
#pragma unroll
  for (int c = 0; c < PH; ++c)
#pragma unroll
    for (int a = 0; a < PW; ++a) {
      publish(0, a, c, t0[c][a]);
      publish(1, a, c, t1[c][a]);
    }
  __syncthreads();
  MLMCPI_STAMP(2);  // buffer in registers, rims published

  for (int s = 0; s < KS; ++s) {
    // row -1: t0(a, -1), t1(a, -1) for a = 0 .. PW (the last from the block below to the right);
    // column PW: t1(PW, c) for c = -1 .. PH - 1 at index c + 1.  None of these changes during phases 0 and 1.
    double dn0[PW], dn1[PW + 1], rt1[PH + 1];
#pragma unroll
    for (int a = 0; a < PW; ++a) {
      dn0[a] = pl(G::top0(a))[dn];
      dn1[a] = pl(G::top1(a))[dn];
    }
    dn1[PW] = pl(G::top1(0))[rtdn];
    rt1[0] = dn1[PW];
#pragma unroll
    for (int c = 0; c < PH; ++c) rt1[c + 1] = pl(G::left1(c))[rt];
    // phases 0, 1: mu = 0, even rows then odd rows
    //   tp = t0(i, j+1) + t1(i, j) - t1(i+1, j),  tm = t0(i, j-1) + t1(i+1, j-1) - t1(i, j-1)
    double up0[PW + 1];  // row PH: t0(a, PH) for a = -1 .. PW - 1 at index a + 1 (final after phase 0)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      if (par == 1) {
#pragma unroll
        for (int a = 0; a < PW; ++a) up0[a + 1] = pl(G::bot0(a))[up];
        up0[0] = pl(G::bot0(PW - 1))[lfup];
      }
#pragma unroll
      for (int c = par; c < PH; c += 2)
#pragma unroll
        for (int a = 0; a < PW; ++a) {
          const double t0_up = c + 1 < PH ? t0[c + 1 < PH ? c + 1 : 0][a] : up0[a + 1];
          const double t0_dn = c > 0 ? t0[c > 0 ? c - 1 : 0][a] : dn0[a];
          const double t1_c = t1[c][a];
          const double t1_r = a + 1 < PW ? t1[c][a + 1 < PW ? a + 1 : 0] : rt1[c + 1];
          const double t1_dr = c > 0 ? (a + 1 < PW ? t1[c > 0 ? c - 1 : 0][a + 1 < PW ? a + 1 : 0] : rt1[c]) : dn1[a + 1];
          const double t1_dc = c > 0 ? t1[c > 0 ? c - 1 : 0][a] : dn1[a];
          const double tp = t0_up + t1_c - t1_r;
          const double tm = t0_dn + t1_dr - t1_dc;
          t0[c][a] = mod_2pi_fast((tp + tm) - t0[c][a]);
          publish(0, a, c, t0[c][a]);
        }
      __syncthreads();
    }
    // column -1: t0(-1, c) for c = 0 .. PH (the last is up0[0]), t1(-1, c); final after phase 1
    double lf0[PH + 1], lf1[PH];
#pragma unroll
    for (int c = 0; c < PH; ++c) {
      lf0[c] = pl(G::right0(c))[lf];
      lf1[c] = pl(G::right1(c))[lf];
    }
    lf0[PH] = up0[0];
    // phases 2, 3: mu = 1, even columns then odd columns
    //   tp = t0(i, j) + t1(i+1, j) - t0(i, j+1),  tm = t0(i-1, j+1) + t1(i-1, j) - t0(i-1, j)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      if (par == 1) {  // the right neighbour's column 0 changed in phase 2
#pragma unroll
        for (int c = 0; c < PH; ++c) rt1[c + 1] = pl(G::left1(c))[rt];
      }
#pragma unroll
      for (int a = par; a < PW; a += 2)
#pragma unroll
        for (int c = 0; c < PH; ++c) {
          const double t0_c = t0[c][a];
          const double t1_r = a + 1 < PW ? t1[c][a + 1 < PW ? a + 1 : 0] : rt1[c + 1];
          const double t0_u = c + 1 < PH ? t0[c + 1 < PH ? c + 1 : 0][a] : up0[a + 1];
          const double t0_lu = a > 0 ? (c + 1 < PH ? t0[c + 1 < PH ? c + 1 : 0][a > 0 ? a - 1 : 0] : up0[a]) : lf0[c + 1];
          const double t1_l = a > 0 ? t1[c][a > 0 ? a - 1 : 0] : lf1[c];
          const double t0_l = a > 0 ? t0[c][a > 0 ? a - 1 : 0] : lf0[c];
          const double tp = t0_c + t1_r - t0_u;
          const double tm = t0_lu + t1_l - t0_l;
          t1[c][a] = mod_2pi_fast((tp + tm) - t1[c][a]);
          publish(1, a, c, t1[c][a]);
        }
      __syncthreads();
    }
  }
}

template <int K>
__global__ void __launch_bounds__(OrBlockGeom<K>::NT)
    schwinger_or_block_kernel(uint32_t Mt, uint32_t Mx, const double2 *__restrict__ in, double2 *__restrict__ out,
                              uint32_t tiles_x) {
  using G = OrBlockGeom<K>;
  constexpr int TW = G::TW, TH = G::TH, PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NP = G::NP;
  extern __shared__ double lds[];
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  double t0[PH][PW], t1[PH][PW];
  MLMCPI_SKEW_START();
  or_block_sweeps<G, K>(lds, in + (size_t)b * Mt * Mx, Mt, Mx, i0, j0, t0, t1);

  // Owned vertices: buffer columns [H, H + TW), rows [H, H + TH).  A thread holds PW consecutive vertices of a row
  // (64 B), so storing block-wise would make every wave instruction write 64 x 16 B at a 64 B stride.  Instead each wave
  // transposes through LDS (the plane area is dead after the last barrier; wave-private staging, no workgroup barrier):
  // per block row c the owners put their four vertices down, and the wave writes the 256 vertices back as 4 coalesced
  // instructions -- lane l takes vertex l & 3 of block 16 i + (l >> 2), i = 0 .. 3.
  static_assert(PW == 4, "the coalesced side moves 4 vertices per block row");
  const uint32_t wave0 = tid & ~63u, lane = tid & 63u;
  double2 *stage = reinterpret_cast<double2 *>(lds) + (wave0 / 64) * (64 * PW);
  double2 *dst = out + (size_t)b * Mt * Mx;
  int uq[4], ur[4];  // tile coordinates of the vertex this lane writes for i = 0 .. 3 (row c = 0); uq < 0: none
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bt = (int)wave0 + 16 * i + (int)(lane >> 2);
    const int bj = bt / NPX, bi = bt - bj * NPX;
    uq[i] = PW * bi + (int)(lane & 3) - H;
    ur[i] = PH * bj - H;
    if (bt >= NP || uq[i] >= TW) uq[i] = -1;
  }
#pragma unroll
  for (int c = 0; c < PH; ++c) {
#pragma unroll
    for (int a = 0; a < PW; ++a) stage[PW * lane + a] = make_double2(t0[c][a], t1[c][a]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double2 w = stage[64 * i + lane];
      const int r = ur[i] + c;
      // (non-temporal, r05: one sweep 0.231 -> 0.226 ms over three same-box pairs)
      if (uq[i] >= 0 && r >= 0 && r < TH) store_streaming(&dst[(size_t)(j0 + r) * Mt + (i0 + uq[i])], w.x, w.y);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The tail shared by schwinger_or_heat_kernel and schwinger_perm_heat_kernel: the heat-bath sweep on the LDS image of a
// 64 x 64 tile and its two rings (theta_0 plane th0, theta_1 plane th1, IW = IH = 68), the optional QoI, the write-out.
template <int NT, bool STEP>
__device__ __forceinline__ void schwinger_image_heat(double *th0, double *th1, VsPool<uint32_t> &vpool, HbPool &hpool, uint32_t Mt,
                                                     uint32_t Mx, double beta, double2 *__restrict__ out, uint32_t i0, uint32_t j0,
                                                     uint32_t b, uint32_t tile, RngKey key0, int qoi_op,
                                                     double *__restrict__ qoi_partial, double *qoi_red) {
  constexpr int TW = 64, TH = 64, HB = 2, IW = TW + 2 * HB;
  // the heat-bath sweep: the last-sweep regions of schwinger_sweep_kernel with H = HB, bw = IW, oh = TH, ow = TW
  constexpr uint32_t bw = IW;
  const uint32_t sc = i0 >= (uint32_t)HB ? i0 - HB : i0 + Mt - HB;  // lattice column of image column 0
  const uint32_t sr = j0 >= (uint32_t)HB ? j0 - HB : j0 + Mx - HB;
  auto wrap = [](uint32_t base, uint32_t off, uint32_t n) {
    const uint32_t v = base + off;
    return v >= n ? v - n : v;
  };
  RngKey skey = key0;
  skey.chain += b;
  const double beta2 = 2. * beta;
  // Step-envelope phases with whole waves per round (NT = 512, 1024): the cells of pass 0 by a closed-form map
  // (heatbath_cells_step_mapped); other workgroup sizes, the wrapped-Cauchy sampler and -DMLMCPI_HB_LINEAR (the form this
  // replaces, for same-box A/B) hand them out by linear index.
#ifdef MLMCPI_HB_LINEAR
  constexpr bool kMapped = false;
#else
  constexpr bool kMapped = STEP && 32 % (NT / kWave) == 0;
#endif
  constexpr uint32_t NW = NT / kWave, NIT = kMapped ? 32 / NW : 1;
  // the stencil reads of the mapped phases go out as single ds_read_b64 at immediate offsets from one address (the compiler
  // pairs neighbouring doubles into ds_read2_b64: 8 LDS cycles against 2 + 2, MI355X_MICROARCH.md); th1 lies IW IH doubles
  // behind th0 at every call site
  constexpr int kT1 = IW * (TH + 2 * HB) * 8;
  const uint32_t lds_th0 = (uint32_t)(uintptr_t)th0;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
  for (uint32_t par = 0; par < 2; ++par) {  // mu = 0: rows [HB, HB + TH] of one parity, columns [HB - 1, HB + TW]
    const uint32_t r_first = HB + par, nr = (HB + TH - r_first) / 2 + 1;
    constexpr uint32_t ncol = TW + 2;
    if (!STEP)
      heatbath_region<NT, 5, true>(
          nr, ncol, skey, hpool,
          [&](uint32_t ri, uint32_t ci, double &tau, double &centre, uint32_t &site, uint32_t &o) {
            const uint32_t r = r_first + 2 * ri, c = HB - 1 + ci;
            o = r * bw + c;
            expcos_params(beta, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw], tau, centre);
            site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt));
          },
          [&](uint32_t o, double v) { th0[o] = v; });
    else if constexpr (kMapped) {
      // pass 0: wave w takes rows ri = w + NW k (k < NIT: 32 of the 33 rows), lane l column ci = l (64 of the 66); the
      // row of the Philox site is a scalar, its column one register for the whole phase
      uint32_t o_next = (r_first + 2 * wave) * bw + (HB - 1) + lane;
      uint32_t srow = wrap(sr, r_first + 2 * wave, Mx);                  // (scalar)
      const uint32_t scol = wrap(sc, HB - 1 + lane, Mt), n_top = (nr - NIT * NW) * ncol;
      heatbath_cells_step_mapped<NT, NIT, uint32_t>(
          n_top + 2 * (NIT * NW), skey, vpool,
          [&](uint32_t &o, uint32_t &site) {
            o = o_next;
            site = (srow * Mt + scol) << 1;
            o_next += 2 * NW * bw;
            srow = wrap(srow, 2 * NW, Mx);
          },
          [&](uint32_t i) {   // left over: row ri = 32 of the even phase (66 cells), then columns 64, 65 of the rows before it
            const uint32_t j = i - n_top, ri = i < n_top ? (uint32_t)(NIT * NW) : j >> 1, ci = i < n_top ? i : 64 + (j & 1u);
            return (r_first + 2 * ri) * bw + (HB - 1) + ci;
          },
          [&](uint32_t o) {
            const uint32_t r = o / bw, c = o - r * bw;
            return 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt));
          },
          [&](uint32_t o, double (&v)[6]) {   // from the address of th0[o - bw]: offsets are unsigned
            const uint32_t a = lds_th0 + (o - bw) * 8u;
            v[0] = lds_read_f64<2 * bw * 8>(a);                 // th0[o + bw]
            v[1] = lds_read_f64<kT1 + bw * 8>(a);               // th1[o]
            v[2] = lds_read_f64<kT1 + bw * 8 + 8>(a);           // th1[o + 1]
            v[3] = lds_read_f64<0>(a);                          // th0[o - bw]
            v[4] = lds_read_f64<kT1 + 8>(a);                    // th1[o - bw + 1]
            v[5] = lds_read_f64<kT1>(a);                        // th1[o - bw]
          },
          [&](const double (&v)[6], VsCell &cell) { vs_cell(beta2, v[0] + v[1] - v[2], v[3] + v[4] - v[5], cell); },
          [&](uint32_t o) {
            return vs_kappa_exact(beta2, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw]);
          },
          [&](uint32_t o, double v) { th0[o] = v; });
    } else
    heatbath_region_step<NT, 5, true, uint32_t>(
        nr, ncol, r_first * bw + (HB - 1), 2 * bw, 1, skey, vpool,
        [&](uint32_t o, VsCell &cell) {
          const uint32_t r = o / bw, c = o - r * bw;
          vs_cell(beta2, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw], cell);
          cell.site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt));
        },
        [&](uint32_t o) {
          return vs_kappa_exact(beta2, th0[o + bw] + th1[o] - th1[o + 1], th0[o - bw] + th1[o - bw + 1] - th1[o - bw]);
        },
        [&](uint32_t o, double v) { th0[o] = v; });
    __syncthreads();
    MLMCPI_STAMP(5 + par);
  }
  for (uint32_t par = 0; par < 2; ++par) {  // mu = 1: rows [HB, HB + TH), even columns up to HB + TW, odd ones up to HB + TW - 1
    const uint32_t c_first = HB + par, c_hi1 = par ? HB + TW - 1 : HB + TW;
    const uint32_t nc = (c_hi1 - c_first) / 2 + 1;
    if (!STEP)
      heatbath_region<NT, 5, true>(
          TH, nc, skey, hpool,
          [&](uint32_t ri, uint32_t ci, double &tau, double &centre, uint32_t &site, uint32_t &o) {
            const uint32_t r = HB + ri, c = c_first + 2 * ci;
            o = r * bw + c;
            expcos_params(beta, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1], tau, centre);
            site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)) + 1;
          },
          [&](uint32_t o, double v) { th1[o] = v; });
    else if constexpr (kMapped) {
      // pass 0: a wave takes two rows and 32 columns per round -- lane l: row ri = 2 (w + NW k) + (l >> 5), column ci = l & 31;
      // the 33rd column of the even phase is left over
      const uint32_t r0 = HB + 2 * wave + (lane >> 5), c0 = c_first + 2 * (lane & 31u);
      uint32_t o_next = r0 * bw + c0;
      uint32_t rowmt = wrap(sr, r0, Mx) * Mt;
      const uint32_t scol = wrap(sc, c0, Mt), mxmt = Mx * Mt;
      heatbath_cells_step_mapped<NT, NIT, uint32_t>(
          (nc - 32) * (uint32_t)TH, skey, vpool,
          [&](uint32_t &o, uint32_t &site) {
            o = o_next;
            site = ((rowmt + scol) << 1) | 1u;
            o_next += 2 * NW * bw;
            rowmt += 2 * NW * Mt;
            rowmt = min(rowmt, rowmt - mxmt);   // (wraps once: the image is no taller than the lattice)
          },
          [&](uint32_t i) { return (HB + i) * bw + c_first + 64; },
          [&](uint32_t o) {
            const uint32_t r = o / bw, c = o - r * bw;
            return 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)) + 1;
          },
          [&](uint32_t o, double (&v)[6]) {   // from the address of th0[o - 1]
            const uint32_t a = lds_th0 + (o - 1) * 8u;
            v[0] = lds_read_f64<8>(a);                          // th0[o]
            v[1] = lds_read_f64<kT1 + 16>(a);                   // th1[o + 1]
            v[2] = lds_read_f64<bw * 8 + 8>(a);                 // th0[o + bw]
            v[3] = lds_read_f64<bw * 8>(a);                     // th0[o + bw - 1]
            v[4] = lds_read_f64<kT1>(a);                        // th1[o - 1]
            v[5] = lds_read_f64<0>(a);                          // th0[o - 1]
          },
          [&](const double (&v)[6], VsCell &cell) { vs_cell(beta2, v[0] + v[1] - v[2], v[3] + v[4] - v[5], cell); },
          [&](uint32_t o) {
            return vs_kappa_exact(beta2, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1]);
          },
          [&](uint32_t o, double v) { th1[o] = v; });
    } else
    heatbath_region_step<NT, 5, true, uint32_t>(
        TH, nc, HB * bw + c_first, bw, 2, skey, vpool,
        [&](uint32_t o, VsCell &cell) {
          const uint32_t r = o / bw, c = o - r * bw;
          vs_cell(beta2, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1], cell);
          cell.site = 2 * (wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)) + 1;
        },
        [&](uint32_t o) {
          return vs_kappa_exact(beta2, th0[o] + th1[o + 1] - th0[o + bw], th0[o + bw - 1] + th1[o - 1] - th0[o - 1]);
        },
        [&](uint32_t o, double v) { th1[o] = v; });
    __syncthreads();
    MLMCPI_STAMP(7 + par);
  }

  // write-out and the optional QoI, as in schwinger_sweep_kernel
  double acc[1] = {0.0};
  double2 *dst = out + (size_t)b * Mt * Mx;
  // (a tile on the upper / right edge of a lattice that 64 x 64 tiles do not divide reaches beyond it: those vertices are
  // periodic images of vertices another tile owns -- computed here like any halo, neither written nor counted)
  for_region<NT>(TH, TW, [&](uint32_t r, uint32_t c) {
    if (j0 + r >= Mx || i0 + c >= Mt) return;
    const uint32_t o = (r + HB) * bw + (c + HB);
    store_streaming(&dst[(size_t)(j0 + r) * Mt + (i0 + c)], th0[o], th1[o]);
    if (qoi_op) {
      const double thp = th0[o] + th1[o + 1] - th0[o + bw] - th1[o];
      acc[0] += qoi_op == 3 ? cos_reduced(thp) : mod_2pi(thp);
    }
  });
  if (qoi_op) {
    block_sum<1>(acc, qoi_red);
    if (threadIdx.x == 0) qoi_partial[(size_t)b * gridDim.x + tile] = acc[0];
  }
  MLMCPI_STAMP(9);
}

// ---- Schwinger: K overrelaxation sweeps and the heat-bath sweep behind them in one launch ------------------------------
// A draw ends "... K overrelaxation sweeps, heat bath, QoI".  As two launches the state makes two round trips through HBM,
// and the two kernels leave opposite halves of the CU idle: the overrelaxation launch is bound by its load / store phases
// and the latencies of its colour phases (vector issue 0.44), the heat bath by vector issue (0.88) with its memory
// traffic hidden.  Here the workgroup that has just swept a tile K times in registers (or_block_sweeps on the geometry
// with halo 2K + 2, i.e. OrBlockGeom<K + 1>) lays the tile and the two rings the heat bath reads down as an LDS image (the
// plane area is dead by then, and large enough), runs the heat-bath sweep of schwinger_sweep_kernel<true, ., 64, 32, true>
// on it -- same regions (the pruned last-sweep form), same cells, same Philox words, same arithmetic: bit-identical
// results -- sums the QoI and writes the tile out.  One round trip instead of two, and the two workgroups of a CU are in
// different phases most of the time: the loads of one run under the sampler arithmetic of the other.
// Step-envelope sampler only (2 beta <= kVsKappaMax); lattices of at least 128 x 128 that 64 x 64 tiles divide.
template <int K>
struct OrHeatGeom {
  using G = OrBlockGeom<K + 1>;
  static constexpr int NT = G::NT, HB = 2, IW = G::TW + 2 * HB, IH = G::TH + 2 * HB;
  static constexpr size_t image_bytes = (size_t)2 * IW * IH * sizeof(double);
  // in front of the image: the sampler's tables and the list of open cells -- a colour phase leaves about 5 % of its ~2200
  // cells on it at beta = 1 (110 entries on average; a list that overflows leaves cells to their own lanes, measured at
  // +20 % on the launch with 64 entries)
  static constexpr uint32_t pool_cap = 544, hb_pool_cap = 128;   // step-envelope list (r05: 256 -> 544 for concentrations up to 16: a phase leaves up to a fifth of its ~2200 cells there; as many entries as the register-block geometry's LDS bound below admits); wrapped-Cauchy pool (24 B per entry)
  static constexpr size_t pool_bytes_of(size_t a, size_t b) { return ((a > b ? a : b) + 15) / 16 * 16; }
  static constexpr size_t pool_bytes = pool_bytes_of(VsPool<uint32_t>::bytes(pool_cap), HbPool::bytes(hb_pool_cap));
  static constexpr size_t hb_bytes = image_bytes + pool_bytes;
  static_assert(hb_bytes <= OrBlockGeom<6>::lds_bytes, "two workgroups per CU");
  static constexpr size_t lds_bytes = G::lds_bytes > hb_bytes ? G::lds_bytes : hb_bytes;
};

// WIDE: 1024 threads per workgroup, for launches with at most one workgroup per CU (few chains): the register-block part
// runs on the first OrHeatGeom<K>::NT threads as before, the heat-bath part on all sixteen waves.
// STEP = false (r04): the heat-bath part draws from the wrapped-Cauchy envelope (heatbath_region, as
// schwinger_sweep_kernel<true, ., 64, 32, false> does): actions beyond 2 beta = kVsKappaMax get the fused launch too.
template <int K, bool WIDE = false, bool STEP = true>
__global__ void __launch_bounds__(WIDE ? 1024 : OrHeatGeom<K>::NT, 4)
    schwinger_or_heat_kernel(uint32_t Mt, uint32_t Mx, double beta, const double2 *__restrict__ in, double2 *__restrict__ out,
                             uint32_t tiles_x, RngKey key0, int qoi_op, double *__restrict__ qoi_partial,
                             const uint32_t *__restrict__ vs_table) {
  using OH = OrHeatGeom<K>;
  using G = typename OH::G;
  constexpr int NT = WIDE ? 1024 : OH::NT, TW = G::TW, TH = G::TH, PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NP = G::NP;
  constexpr int HB = OH::HB, IW = OH::IW, IH = OH::IH, O = H - HB;  // image (0, 0) = buffer (O, O)
  extern __shared__ double lds[];
  __shared__ double qoi_red[NT / kWave];
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  double t0[PH][PW], t1[PH][PW];
  MLMCPI_STAMP(0);
  MLMCPI_STAMP_WHERE();
  MLMCPI_SKEW_START();
  or_block_sweeps<G, K>(lds, in + (size_t)b * Mt * Mx, Mt, Mx, i0, j0, t0, t1);
  MLMCPI_STAMP(3);  // K sweeps done

  // The sampler's tables, round counters and retry pool at the START of the LDS (their addresses are then instruction
  // offsets: a table look-up costs no address arithmetic beyond its index), the image behind them: theta_0 and theta_1
  // planes of IH x IW vertices; every block that reaches into it puts its part down
  VsPool<uint32_t> vpool = VsPool<uint32_t>::carve(lds, OH::pool_cap, STEP ? vs_table : nullptr);
  HbPool hpool = HbPool::carve(lds, STEP ? 0u : OH::hb_pool_cap);
  double *th0 = lds + OH::pool_bytes / sizeof(double), *th1 = th0 + IW * IH;
  if (tid < NP) {
    const int pj = (int)tid / NPX, pi = (int)tid - pj * NPX;
#pragma unroll
    for (int c = 0; c < PH; ++c) {
      const int r = PH * pj + c - O;
      if (r < 0 || r >= IH) continue;
#pragma unroll
      for (int a = 0; a < PW; ++a) {
        const int q = PW * pi + a - O;
        if (q >= 0 && q < IW) {
          th0[r * IW + q] = t0[c][a];
          th1[r * IW + q] = t1[c][a];
        }
      }
    }
  }
  __syncthreads();
  MLMCPI_STAMP(4);  // image down

  schwinger_image_heat<NT, STEP>(th0, th1, vpool, hpool, Mt, Mx, beta, out, i0, j0, b, tile, key0, qoi_op, qoi_partial, qoi_red);
}

// ---- Schwinger overrelaxation in closed form: K sweeps are a fixed permutation of the plaquettes -----------------------
// With P(i, j) = theta_0(i, j) + theta_1(i+1, j) - theta_0(i, j+1) - theta_1(i, j) the two staple sums of the mu = 0 link at
// (i, j) are theta - P(i, j) and theta + P(i, j-1) (quenchedschwingeraction.cc:25-43), so its overrelaxation update
// (quenchedschwingeraction.cc:57-65) is
//     theta <- theta + P(i, j-1) - P(i, j)   (mod 2 pi),
// after which P(i, j) and P(i, j-1) have changed places; likewise theta_1(i, j) <- theta_1 + P(i, j) - P(i-1, j) swaps
// P(i, j) and P(i-1, j).  In the multicolour order of the sweeps here -- (mu = 0, j even), (mu = 0, j odd), (mu = 1, i even),
// (mu = 1, i odd) -- a colour phase therefore swaps whole rows (columns) of plaquettes pairwise, and one sweep moves the
// plaquette at an even row (column) index two rows (columns) down and the one at an odd index two up: after s sweeps
//     P_s(i, j) = P_0(i + 2 s e_i, j + 2 s e_j),   e_x = +1 for even x, -1 for odd x,
// whatever the field is.  Summing the increments of a link over K sweeps gives (s = 0 .. K - 1)
//     theta_0(i, j)  +=  sum_s  P_0(i + 2 s e_i, j - 1 - p_j - 2 s)  -  P_0(i + 2 s e_i, j + p_j + 2 s),        p_x = x mod 2,
//     theta_1(i, j)  +=  sum_s  P_0(i + p_i + 2 s, j + 2 (s + 1) e_j)  -  P_0(i - 1 - p_i - 2 s, j + 2 (s + 1) e_j),
// the same map as K sweeps of any other overrelaxation kernel here up to the rounding of 4 K additions (measured against
// them and against the oracle's sweeps: <= 3e-14 at K = 10).  A link costs 2 K LDS reads and 2 K additions instead of
// 9.3 K fp64 instructions, there is no halo recomputation (only the links that are wanted are computed), no colour
// phases and no barriers between sweeps; what remains is the halo of 2 K in the plaquettes a workgroup needs.
// Pairs: the two mu = 0 links of a column at rows (j, j + 1), j even, share their first stream (p_j cancels in it), and so
// do the two mu = 1 links of a row at columns (i, i + 1), i even, their second: a task is such a pair -- three streams of K
// plaquettes, two links.  With S, X, X' the sums over the shared stream and the two others (each in the order s = 0, 1, ...),
//     mu = 0:  theta_0(i, j) += S - X,  theta_0(i, j+1) += S - X';        mu = 1:  theta_1(i, j) += X - S,  theta_1(i+1, j) += X' - S.
// That order of operations is the definition: a result depends on the field and K only -- not on the tile, the batch, the
// workgroup size or the kernel (schwinger_perm_kernel == the first part of schwinger_perm_heat_kernel, bit for bit) -- but
// K sweeps in one launch and the same sweeps in two differ in the last bits.
//
// The plane: P_0 over `rows` x W vertices in LDS.  A wave takes 63 columns of a group of rows and walks up: lane l loads
// the double2 of column 63 cw + l, row by row -- the load of the next row is the theta_0(i, j+1) of this one, and
// theta_1(i+1, j) comes from lane l + 1 by DPP (lane 63 only serves lane 62): ONE coalesced 16-byte load per plaquette,
// U + 1 rows in flight per thread.  W + 1 <= Mt and rows + 1 <= Mx are not required: columns and rows wrap as often as
// needed (a 64 x 64 lattice is its own halo).
constexpr uint32_t kPermMaxK = 10;  // sweeps per launch
#ifndef MLMCPI_PERM_U
#define MLMCPI_PERM_U 0    // rows in flight per thread in the first plane build; 0 = all of a thread's rows at the deepest launch (19 at 512 threads; r05: one round trip to HBM instead of two, 10 + 9 rows: -4.2 % on the launch, same-box A/B)
#endif

// Workgroups are handed to the 8 XCDs round robin by their linear index; each XCD has its own L2.  Experiment
// (-DMLMCPI_XCD_MAP): XCD x takes the x-th eighth of the (chain, tile) list instead, so that the workgroups resident on an
// XCD at any time are neighbouring tiles of one or two chains, whose halos -- 65 % of what a workgroup of the 10-sweep
// launch loads -- could be L2 hits.
__device__ __forceinline__ void perm_tile_of_workgroup(uint32_t &tile, uint32_t &b) {
  tile = blockIdx.x;
  b = blockIdx.y;
#ifdef MLMCPI_XCD_MAP   // measured (r04, same-box A/B at 1024 x 1024 x 32 and x 128): no difference
  const uint32_t total = gridDim.x * gridDim.y;
  if (total % 8 == 0) {
    const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x, id = (lin % 8) * (total / 8) + lin / 8;
    b = id / gridDim.x;
    tile = id - b * gridDim.x;
  }
#endif
}
// The plane in LDS.  Every stream of the closed form walks a diagonal of the plaquettes of ONE parity class: column and
// row parity do not change along it, the column moves by 2 e_c and the row by +-2 per step s.  So the plane is kept as four
// quadrants by (column parity, row parity), each Rh = rows / 2 rows of Wh = WP / 2 values, with the ODD index mirrored:
//     column C -> u = C / 2 (C even),  Wh - 1 - (C - 1) / 2 (C odd);      row R -> v = R / 2,  Rh - 1 - (R - 1) / 2 likewise.
// A step of any stream of any task is then (u, v) -> (u + 1, v + 1): ONE byte stride, kStep, for all of them -- whatever
// the parities, mu = 0 or 1 -- and with the pitch WP a compile-time constant (the width of the deepest launch, kPermMaxK
// sweeps; a shallower one leaves columns unused) the K reads of a stream are K immediate offsets from one address.  (The
// row-major plane this replaces cost 6.6 integer instructions of address arithmetic per read: a third of the launch's
// vector instructions outside the heat bath.)  Same values, same order of additions: results are bit for bit those of the
// row-major form.
template <int WP>
struct PermPlane {
  static constexpr int Wh = WP / 2, kRow = Wh * 8, kStep = kRow + 8;   // bytes
  uint32_t Rh, QB;                                                      // rows per quadrant; bytes per quadrant
  __device__ __forceinline__ explicit PermPlane(uint32_t rows) : Rh(rows / 2), QB((rows / 2) * (uint32_t)kRow) {}
  // byte offset of plaquette (C, R) = col(C) + row(R)
  __device__ __forceinline__ uint32_t col(uint32_t C) const { return (C & 1u) ? QB + (uint32_t)(Wh - 1 - (int)(C >> 1)) * 8u : (C >> 1) * 8u; }
  __device__ __forceinline__ uint32_t row(uint32_t R) const { return (R & 1u) ? 2u * QB + (Rh - 1u - (R >> 1)) * (uint32_t)kRow : (R >> 1) * (uint32_t)kRow; }
};

// where a thread stands in a build: its theta column, its rows [r, rend) of the `rows`, whether it owns a plaquette column
struct PermBuildPos {
  uint32_t c, r, rend, row_off, gj;   // row_off = gj Mt: the lattice row of build row r, in vertices (< 2^32: check_lattice_dims)
  uint32_t cb;                        // PermPlane::col of the plaquette column
  const double2 *p;                   // src + the lattice column
  bool active, owns;
};
template <int NT, class PP>
__device__ __forceinline__ PermBuildPos perm_build_pos(const PP &P, const double2 *__restrict__ src, uint32_t Mt, uint32_t Mx, uint32_t gi0,
                                                       uint32_t gj0, uint32_t W, uint32_t rows) {
  PermBuildPos q;
  // (the wave index on the scalar side: rows, row offsets and the loop conditions of the build are then scalar too)
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
  const uint32_t nwc = W > 63 ? 2 : 1;                                // column waves per row group (W <= 108: at most 2)
  const uint32_t groups = (NT / kWave) >> (nwc - 1);                   // row groups
  const uint32_t g = wave >> (nwc - 1), cw = wave & (nwc - 1);
  static_assert(((NT / kWave) & (NT / kWave - 1)) == 0, "waves per workgroup: a power of two (shifts for divisions)");
  const uint32_t rpg = groups == 4 ? (rows + 3) >> 2 : groups == 8 ? (rows + 7) >> 3 : (rows + groups - 1) / groups;
  q.c = 63 * cw + lane;                                                // theta column; the plaquette column of lanes 0 .. 62
  q.r = g * rpg;
  q.rend = g < groups ? min(rows, q.r + rpg) : 0;
  q.active = q.r < q.rend && q.c <= W;   // (not: whole waves, or the lanes beyond theta column W, which nobody reads)
  q.owns = lane < 63 && q.c < W;
  q.cb = P.col(q.c);
  q.p = src + wrap_add(gi0, q.c, Mt);
  q.gj = wrap_add(gj0, q.r, Mx);
  q.row_off = q.gj * Mt;
  return q;
}
// rows (q.r, min(q.r + U, q.rend)] of the thread's column into nxt[0 .. U); first: row q.r itself into cur
template <int U>
__device__ __forceinline__ void perm_rows_load(PermBuildPos &q, uint32_t Mt, uint32_t Mx, bool first, double2 &cur, double2 (&nxt)[U]) {
  if (!q.active) return;
  if (first) cur = q.p[q.row_off];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    q.gj = q.gj + 1 == Mx ? 0 : q.gj + 1;
    q.row_off = q.gj == 0 ? 0 : q.row_off + Mt;
    if (q.r + u < q.rend) nxt[u] = q.p[q.row_off];
  }
}
// their plaquettes into the plane (build row 0 = plane row R0); cur <- the last row, for the next chunk
template <int U, class PP>
__device__ __forceinline__ void perm_rows_store(PermBuildPos &q, const PP &P, double *plane, uint32_t R0, double2 &cur, const double2 (&nxt)[U]) {
  if (!q.active) return;
  char *const pb = reinterpret_cast<char *>(plane) + q.cb;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (q.r + u < q.rend) {
      const double right = wave_rotate_down(cur.y);   // theta_1 of the next column
      if (q.owns) *reinterpret_cast<double *>(pb + P.row(R0 + q.r + u)) = ((cur.x + right) - nxt[u].x) - cur.y;   // (the row part is scalar)
      cur = nxt[u];
    }
  q.r += U;
}
template <int NT, int U, class PP>
__device__ __forceinline__ void perm_build_rows(const PP &P, double *plane, const double2 *__restrict__ src, uint32_t Mt, uint32_t Mx,
                                                uint32_t gi0, uint32_t gj0, uint32_t W, uint32_t rows) {
  PermBuildPos q = perm_build_pos<NT>(P, src, Mt, Mx, gi0, gj0, W, rows);
  double2 cur = make_double2(0., 0.);
  bool first = true;
  while (__builtin_amdgcn_readfirstlane(q.r) < __builtin_amdgcn_readfirstlane(q.rend)) {   // (uniform per wave)
    double2 nxt[U];
    perm_rows_load<U>(q, Mt, Mx, first, cur, nxt);
    perm_rows_store<U>(q, P, plane, 0u, cur, nxt);
    first = false;
  }
}

// K sweeps for the (64 + 2 RING) x (TH + 2 RING) vertices around a 64 x TH tile (RING = 0: the tile; RING = 2: what the heat
// bath behind the sweeps reads; TH = 32: lattices that 64 x 32 tiles divide and 64 x 64 ones do not), in two halves of
// HR = TH / 2 + RING rows.  NB = 1: one plane of 2 HR + 4 K rows serves both; NB = 2 (K
// sweeps reach 2 K rows up and down: beyond K = 6 the whole plane does not fit beside a second workgroup): a plane of
// HR + 4 K rows; for the second half its upper HR + 4 K - HR rows move down and HR new rows are built on top.
// Tasks of a half: (HR / 2) x OW column pairs (mu = 0), then HR x (OW / 2) row pairs (mu = 1); thread t takes t, t + NT, ...
template <int NT, int RING, int TH = 64>
struct PermGeom {
  static constexpr int OW = 64 + 2 * RING, HR = TH / 2 + RING, NTASK = HR * OW, NV = (NTASK + NT - 1) / NT;
  static constexpr int WP = OW + 4 * (int)kPermMaxK;   // the plane's pitch (PermPlane): the width of the deepest launch
  static_assert(HR % 2 == 0 && OW % 2 == 0, "parities of the output = parities of the lattice; the halves move by whole quadrant rows");
  static __host__ __device__ constexpr uint32_t width(uint32_t K) { return OW + 4 * K; }
  static __host__ __device__ constexpr uint32_t rows(uint32_t K, uint32_t NB) { return (NB == 2 ? HR : 2 * HR) + 4 * K; }
  static __host__ __device__ constexpr size_t plane_bytes(uint32_t K, uint32_t NB) { return (size_t)WP * rows(K, NB) * sizeof(double); }
};

// The tasks of a half and who takes them.  A task is a column pair (mu = 0: rows r, r + 1 of column c) or a row pair
// (mu = 1: columns c, c + 1 of row r), coordinates inside the half.  r05: whole WAVES take whole rows -- wave-task m of a
// half is, for m < HR / 2, the mu = 0 tasks of row pair m in columns 0 .. 63 (lane l: the even columns on lanes 0 .. 31,
// the odd ones on 32 .. 63: a 32-lane group of a gather read stays inside one quadrant of the plane, contiguous banks),
// and for m >= HR / 2 the mu = 1 tasks of rows 2 (m - HR / 2) + (l >> 5) in column pairs l & 31; wave w takes m = w, w + NW,
// ... (slot k: m = w + NW k).  The kind of a slot and the row of its tasks are then wave-uniform and the column of a lane
// is the same in every slot: what was ~65 vector instructions of index arithmetic per task in the three places that need
// coordinates (own angles, gather, image) is scalar work plus a few additions.  The columns beyond 64 of the 68-wide
// output of the fused launch (4 x HR / 2 mu = 0 tasks, 2 x HR mu = 1 tasks) are left-over wave-tasks of one kind each, in
// the last slot of waves that have no main task there.  Which lane computes a task does not enter its result.
// (-DMLMCPI_PERM_TASKS_LINEAR: thread t takes tasks t, t + NT, ... of the list "mu = 0 row pairs, then mu = 1 rows", r04.)
#ifndef MLMCPI_PERM_TASKS_LINEAR
template <int NT, int RING, int TH = 64>
struct PermTasks {
  using PG = PermGeom<NT, RING, TH>;
  static constexpr int OW = PG::OW, HR = PG::HR, H2 = HR / 2, NW = NT / kWave, NS = (HR + NW - 1) / NW;
  static constexpr int XC = OW - 64;                                   // columns beyond a wave's 64 (0 or 4)
  static constexpr int L0 = XC * H2, L1 = (XC / 2) * HR;               // left-over tasks, mu = 0 and mu = 1
  static constexpr int NL0 = (L0 + 63) / 64, NL1 = (L1 + 63) / 64;     // ... as wave-tasks
  static constexpr int WF = HR - NW * (NS - 1);                        // the first wave without a main task in slot NS - 1
  static_assert(PG::NV == NS, "slots per thread");
  static_assert(NL0 + NL1 <= NW - WF, "the left-over wave-tasks fit the free last slots");
  uint32_t wave, lane, c_mu0, c_mu1, r_lo;
  __device__ PermTasks() {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    lane = threadIdx.x % kWave;
    c_mu0 = lane < 32 ? 2 * lane : 2 * (lane - 32) + 1;
    c_mu1 = 2 * (lane & 31u);
    r_lo = lane >> 5;
  }
  // slot k of this thread: false when there is no task in it; mu1 is wave-uniform
  __device__ __forceinline__ bool task(int k, bool &mu1, uint32_t &r, uint32_t &c) const {
    const uint32_t m = wave + (uint32_t)(NW * k);
    mu1 = false; r = 0; c = 0;
    if (NW * k + NW - 1 < H2 || m < (uint32_t)H2) {            // (the first clause: known at compile time for the early slots)
      r = 2 * m;
      c = c_mu0;
      return true;
    }
    if (m < (uint32_t)HR) {
      mu1 = true;
      r = 2 * (m - H2) + r_lo;
      c = c_mu1;
      return true;
    }
    if (XC > 0 && k == NS - 1) {
      const uint32_t j = wave - (uint32_t)WF;
      if (j < (uint32_t)NL0) {                                // mu = 0, columns 64 ..: task L = row pair L / XC, column 64 + L % XC
        const uint32_t L = 64 * j + lane;
        r = 2 * (L / (XC ? XC : 1));
        c = 64 + L % (XC ? XC : 1);
        return L < (uint32_t)L0;
      }
      if (j < (uint32_t)(NL0 + NL1)) {                        // mu = 1, column pairs 32 ..: row L / (XC / 2), column 64 + 2 (L % (XC / 2))
        const uint32_t L = 64 * (j - NL0) + lane;
        mu1 = true;
        r = L / (XC > 1 ? XC / 2 : 1);
        c = 64 + 2 * (L % (XC > 1 ? XC / 2 : 1));
        return L < (uint32_t)L1;
      }
    }
    return false;
  }
  __device__ __forceinline__ bool valid(int k) const {
    bool mu1; uint32_t r, c;
    return task(k, mu1, r, c);
  }
  __device__ __forceinline__ bool is_mu1(int k) const {
    bool mu1; uint32_t r, c;
    task(k, mu1, r, c);
    return mu1;
  }
  __device__ __forceinline__ void coords(int k, uint32_t &r, uint32_t &c) const {
    bool mu1;
    task(k, mu1, r, c);
  }
};
#else
// Task k of a thread, t = threadIdx.x + k NT: a column pair (mu = 0: rows r, r + 1 of column c; t < NT0) or a row pair
// (mu = 1: columns c, c + 1 of row r), coordinates inside the half.  Two divisions per thread (PermTasks), then constants.
template <int NT, int RING, int TH = 64>
struct PermTasks {
  using PG = PermGeom<NT, RING, TH>;
  static constexpr int OW = PG::OW, HR = PG::HR, NT0 = (HR / 2) * OW, OW2 = OW / 2;
  uint32_t q0, c0, q1, c1;   // threadIdx.x = q0 OW + c0 = q1 OW2 + c1
  __device__ PermTasks() {
    q0 = threadIdx.x / OW;
    c0 = threadIdx.x - q0 * OW;
    q1 = threadIdx.x / OW2;
    c1 = threadIdx.x - q1 * OW2;
  }
  __device__ __forceinline__ bool valid(int k) const { return threadIdx.x + k * NT < (uint32_t)PG::NTASK; }
  __device__ __forceinline__ bool is_mu1(int k) const { return threadIdx.x + k * NT >= (uint32_t)NT0; }
  __device__ __forceinline__ void coords(int k, uint32_t &r, uint32_t &c) const {
    if (!is_mu1(k)) {   // t = (q0 + dq) OW + c0 + dc
      const uint32_t dq = (k * NT) / OW, dc = (k * NT) % OW;
      uint32_t q = q0 + dq, cc = c0 + dc;
      if (cc >= (uint32_t)OW) { cc -= OW; ++q; }
      r = 2 * q;
      // the first OW / 2 tasks of a row pair take the even columns, the rest the odd ones: a 32-lane group of a gather read
      // stays inside one quadrant of the plane, contiguous banks (r05, same-box A/B: 0.7703 against 0.7747 ms with c = cc)
      c = cc < (uint32_t)(OW / 2) ? 2 * cc : 2 * (cc - OW / 2) + 1;
    } else {            // t - NT0 = (q1 + dq) OW2 + c1 + dc, dq possibly negative
      const int off = k * NT - NT0, dq = off >= 0 ? off / OW2 : -((-off + OW2 - 1) / OW2), dc = off - dq * OW2;   // 0 <= dc < OW2
      uint32_t q = q1 + (uint32_t)dq, cc = c1 + (uint32_t)dc;
      if (cc >= (uint32_t)OW2) { cc -= OW2; ++q; }
      r = q;
      c = 2 * cc;
    }
  }
};
#endif

// Five steps of the three streams of a task: fifteen 8-byte LDS reads at immediate offsets from three addresses, through
// inline asm (lds_read_f64: the compiler would pair the reads of a stream into ds_read2_b64, half the rate --
// MI355X_MICROARCH.md, LDS table), one wait naming all fifteen, then the additions in the order s = 0, 1, ...
template <int STEP>
__device__ __forceinline__ void perm_gather5(uint32_t pa, uint32_t px, uint32_t px2, double &S, double &X, double &X2) {
  double a0 = lds_read_f64<0>(pa), x0 = lds_read_f64<0>(px), y0 = lds_read_f64<0>(px2);
  double a1 = lds_read_f64<STEP>(pa), x1 = lds_read_f64<STEP>(px), y1 = lds_read_f64<STEP>(px2);
  double a2 = lds_read_f64<2 * STEP>(pa), x2 = lds_read_f64<2 * STEP>(px), y2 = lds_read_f64<2 * STEP>(px2);
  double a3 = lds_read_f64<3 * STEP>(pa), x3 = lds_read_f64<3 * STEP>(px), y3 = lds_read_f64<3 * STEP>(px2);
  double a4 = lds_read_f64<4 * STEP>(pa), x4 = lds_read_f64<4 * STEP>(px), y4 = lds_read_f64<4 * STEP>(px2);
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a0), "+v"(x0), "+v"(y0), "+v"(a1), "+v"(x1), "+v"(y1), "+v"(a2), "+v"(x2), "+v"(y2), "+v"(a3), "+v"(x3), "+v"(y3),
                 "+v"(a4), "+v"(x4), "+v"(y4)
               :
               : "memory");
  S += a0; X += x0; X2 += y0;
  S += a1; X += x1; X2 += y1;
  S += a2; X += x2; X2 += y2;
  S += a3; X += x3; X2 += y3;
  S += a4; X += x4; X2 += y4;
}
__device__ __forceinline__ void perm_gather1(uint32_t pa, uint32_t px, uint32_t px2, double &S, double &X, double &X2) {
  double a0 = lds_read_f64<0>(pa), x0 = lds_read_f64<0>(px), y0 = lds_read_f64<0>(px2);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(x0), "+v"(y0) : : "memory");
  S += a0; X += x0; X2 += y0;
}

// res[h][k] = the new angles of the two links of task k of half h
template <int NT, int RING, int TH = 64>
__device__ __forceinline__ void perm_sweeps(double *plane, const double2 *__restrict__ src, uint32_t Mt, uint32_t Mx, uint32_t i0,
                                            uint32_t j0, uint32_t K, uint32_t NB, double2 (&res)[2][PermGeom<NT, RING, TH>::NV]) {
  using PG = PermGeom<NT, RING, TH>;
  using PP = PermPlane<PG::WP>;
  constexpr int HR = PG::HR, NV = PG::NV;
  // all of a thread's rows at the deepest launch: (HR + 4 kPermMaxK) rows over (NT / 64) / 2 row groups (W > 63: two column waves)
  constexpr int kGroups = NT / kWave / 2;
  constexpr int U = MLMCPI_PERM_U ? MLMCPI_PERM_U : (HR + 4 * (int)kPermMaxK + kGroups - 1) / kGroups, UB = (HR + kGroups - 1) / kGroups;   // rows in flight per thread: first build (nothing else is live yet), new rows of the second
  const uint32_t W = PG::width(K), rows = PG::rows(K, NB), H = RING + 2 * K;
  const PP P(rows);
  // lattice coordinates of plane (0, 0) of the first build, and of output vertex (0, 0)
  // (x - h) mod n for x < n: a comparison where h <= n -- the rule; the two modulo operations of the general form are ~80
  // instructions each in front of the first load of the workgroup
  auto back = [](uint32_t x, uint32_t h, uint32_t n) { return h <= n ? (x >= h ? x - h : x + n - h) : (x + n - h % n) % n; };
  const uint32_t gi0 = back(i0, H, Mt), gj0 = back(j0, H, Mx);
  const uint32_t oi0 = back(i0, RING, Mt), oj0 = back(j0, RING, Mx);
  const PermTasks<NT, RING, TH> tasks;
  // the links of a half as they are now (HR, RING, 2 K and the tile origins are even: output parity = plane parity = lattice parity)
  // (32-bit byte offsets from the chain's base pointer -- the host admits lattices of less than 2^28 vertices to these
  // kernels --, wraps by the unsigned-minimum trick: v >= n ? v - n : v = min(v, v - n); twice more for extents below
  // the output window's, a uniform branch.  The 64-bit pointer form this replaces cost 30 vector instructions per task.)
  const char *const src_b = reinterpret_cast<const char *>(src);
  const bool small_lattice = Mx < (uint32_t)(2 * HR + 2) || Mt < (uint32_t)(PG::OW + 2);   // (uniform: two copies of the loop)
  auto load_theta_of = [&](int h, double2 (&th)[NV], auto small) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!tasks.valid(k)) continue;
      uint32_t r, c;
      tasks.coords(k, r, c);
      uint32_t gr = oj0 + r + h * HR, gc = oi0 + c;
      gr = min(gr, gr - Mx);
      gc = min(gc, gc - Mt);
      if ((bool)small) {
        gr = min(gr, gr - Mx); gr = min(gr, gr - Mx);
        gc = min(gc, gc - Mt); gc = min(gc, gc - Mt);
      }
      const uint32_t o = (gr * Mt + gc) * 16u;
      if (!tasks.is_mu1(k)) {
        const uint32_t o2 = gr + 1 == Mx ? o - gr * Mt * 16u : o + Mt * 16u;
        th[k] = make_double2(*reinterpret_cast<const double *>(src_b + o), *reinterpret_cast<const double *>(src_b + o2));
      } else {
        const uint32_t o2 = gc + 1 == Mt ? o - gc * 16u : o + 16u;
        th[k] = make_double2(*reinterpret_cast<const double *>(src_b + o + 8u), *reinterpret_cast<const double *>(src_b + o2 + 8u));
      }
    }
  };
  auto load_theta = [&](int h, double2 (&th)[NV]) {
    if (small_lattice) load_theta_of(h, th, std::true_type{}); else load_theta_of(h, th, std::false_type{});
  };
  // what K sweeps add to them
  auto gather = [&](int h, double2 (&d)[NV]) {
    const uint32_t row_off = 2 * K + (NB == 2 ? 0 : h * HR);  // plane row of output row 0 of this half (even)
    const uint32_t lds0 = (uint32_t)(uintptr_t)plane;   // the LDS byte address of the plane (see schwinger_or_kernel)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!tasks.valid(k)) continue;
      uint32_t r, c;
      tasks.coords(k, r, c);
      // the shared stream, the first of the two others and the second, at s = 0 (plane coordinates C = c + 2 K, R = r + row_off):
      //   mu = 0 (R even): A_s = P(C + 2 s e_C, R - 1 - 2 s), B_s = P(C + 2 s e_C, R + 2 s), B'_s two rows above B_s;
      //   mu = 1 (C even): D_s = P(C - 1 - 2 s, J_s), J_s = R + 2 (s + 1) e_R, C_s = P(C + 2 s, J_s), C'_s two columns on;
      // in quadrant coordinates every one of them advances by (1, 1) per s
      const uint32_t C = c + 2 * K, R = r + row_off, Rq = R >> 1;
      uint32_t a, x, x2;
      if (!tasks.is_mu1(k)) {
        const uint32_t cb = P.col(C);
        x = cb + Rq * (uint32_t)PP::kRow;                       // row R, even: v = R / 2
        x2 = x + (uint32_t)PP::kRow;
        a = cb + 2u * P.QB + (P.Rh - Rq) * (uint32_t)PP::kRow;  // row R - 1, odd: v = Rh - 1 - (R / 2 - 1)
      } else {
        // J_0 = R + 2 (v = R / 2 + 1) for even R, R - 2 (v = Rh - 1 - ((R - 1) / 2 - 1)) for odd R
        const uint32_t rb = (R & 1u) ? 2u * P.QB + (P.Rh - Rq) * (uint32_t)PP::kRow : (Rq + 1u) * (uint32_t)PP::kRow;
        x = rb + (C >> 1) * 8u;                                  // column C, even: u = C / 2
        x2 = x + 8u;
        a = rb + P.QB + ((uint32_t)PP::Wh - (C >> 1)) * 8u;      // column C - 1, odd: u = Wh - 1 - (C / 2 - 1)
      }
      uint32_t pa = lds0 + a, px = lds0 + x, px2 = lds0 + x2;
      double S = 0.0, X = 0.0, X2 = 0.0;
      uint32_t s = 0;
      for (; s + 5 <= K; s += 5) {
        perm_gather5<PP::kStep>(pa, px, px2, S, X, X2);
        pa += 5 * PP::kStep;
        px += 5 * PP::kStep;
        px2 += 5 * PP::kStep;
      }
      for (; s < K; ++s) {
        perm_gather1(pa, px, px2, S, X, X2);
        pa += PP::kStep;
        px += PP::kStep;
        px2 += PP::kStep;
      }
      d[k] = tasks.is_mu1(k) ? make_double2(X - S, X2 - S) : make_double2(S - X, S - X2);
    }
  };
  auto finish = [&](const double2 (&th)[NV], double2 (&d)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (tasks.valid(k)) d[k] = make_double2(mod_2pi_fast(th[k].x + d[k].x), mod_2pi_fast(th[k].y + d[k].y));
  };

  perm_build_rows<NT, U>(P, plane, src, Mt, Mx, gi0, gj0, W, rows);
  double2 th[NV];
  if (NB == 1) {
    // one plane: first half (its angles are in flight across the barrier and the first reads of the plane), second half
    load_theta(0, th);
    __syncthreads();
    MLMCPI_STAMP(1);  // plane built
    gather(0, res[0]);
    finish(th, res[0]);
    MLMCPI_STAMP(2);  // first half gathered
    load_theta(1, th);
    gather(1, res[1]);
    finish(th, res[1]);
    return;
  }
  // NB = 2: the HR new rows of the second plane are loaded while the first half is gathered (at most UB rows per thread:
  // HR / 4 row groups); the angles of the first half only after it, so that the gather has the registers
  PermBuildPos qb = perm_build_pos<NT>(P, src, Mt, Mx, gi0, wrap_add(gj0, rows, Mx), W, HR);
  double2 curb = make_double2(0., 0.), vb[UB];
  __syncthreads();
  MLMCPI_STAMP(1);  // plane built
  perm_rows_load<UB>(qb, Mt, Mx, true, curb, vb);
  gather(0, res[0]);
  MLMCPI_STAMP(2);  // first half gathered
#ifdef MLMCPI_THETA_EARLY
  // experiment (r05, same-box A/B: 0.781 against 0.778 ms without): the angles of the first half in flight while the plane's
  // rows move -- the longer live range costs a spill inside the second gather, which then waits for every load in flight
  load_theta(0, th);
#endif
  // rows [HR, rows) of the plane become rows [0, rows - HR), the HR new rows go on top.  HR is even: a row keeps its parity
  // and moves by HR / 2 quadrant rows -- down in the quadrants of the even rows, up (mirrored) in those of the odd rows
  constexpr int NC = (4 * 2 * (int)kPermMaxK * PP::Wh + NT - 1) / NT;   // 4 quadrants x (rows - HR) / 2 = 2 K quadrant rows
  const uint32_t nkeep = (rows - HR) / 2 * (uint32_t)PP::Wh, shift = (uint32_t)(HR / 2) * (uint32_t)PP::kRow;
  {
    double keep[NC];
    char *const pbw = reinterpret_cast<char *>(plane);
    // (a thread reads what it moves as soon as its own gather is done -- reads beside the reads of the gathers still running
    // -- and ONE barrier separates every read of the old plane, the gathers' and these, from the writes; the barrier that
    // used to stand in front of these reads as well was worth 0.1 % of the launch, same-box A/B)
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const uint32_t idx = threadIdx.x + q * NT, e = idx >> 2, qd = idx & 3u;
      if (e < nkeep) keep[q] = *reinterpret_cast<const double *>(pbw + qd * P.QB + e * 8u + ((qd & 2u) ? 0u : shift));
    }
    __syncthreads();  // the first half has read its plane, and so have the threads that move its rows
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const uint32_t idx = threadIdx.x + q * NT, e = idx >> 2, qd = idx & 3u;
      if (e < nkeep) *reinterpret_cast<double *>(pbw + qd * P.QB + e * 8u + ((qd & 2u) ? shift : 0u)) = keep[q];
    }
  }
#ifdef MLMCPI_THETA_MID
  load_theta(0, th);
#endif
  perm_rows_store<UB>(qb, P, plane, rows - HR, curb, vb);
#if !defined(MLMCPI_THETA_EARLY) && !defined(MLMCPI_THETA_MID)
  load_theta(0, th);
#endif
  double2 th1[NV];
  load_theta(1, th1);
  __syncthreads();
  MLMCPI_STAMP(10);  // second plane built
  finish(th, res[0]);   // (behind the gather instead: 48 bytes of spills)
  gather(1, res[1]);
  finish(th1, res[1]);
}

// the angles of perm_sweeps into the planes th0, th1 of an OW x 2 HR image (the caller puts barriers around it)
template <int NT, int RING, int TH = 64>
__device__ __forceinline__ void perm_store_image(double *th0, double *th1, const double2 (&res)[2][PermGeom<NT, RING, TH>::NV]) {
  using PG = PermGeom<NT, RING, TH>;
  constexpr int OW = PG::OW, HR = PG::HR, NV = PG::NV;
  const PermTasks<NT, RING, TH> tasks;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!tasks.valid(k)) continue;
      uint32_t r, c;
      tasks.coords(k, r, c);
      const uint32_t o = (r + h * HR) * OW + c;
      if (!tasks.is_mu1(k)) {
        th0[o] = res[h][k].x;
        th0[o + OW] = res[h][k].y;
      } else {
        th1[o] = res[h][k].x;
        th1[o + 1] = res[h][k].y;
      }
    }
}

// K <= kPermMaxK overrelaxation sweeps of a 64 x TH tile per workgroup; lds = perm_lds_bytes<TH>(K, NB)
constexpr size_t kPermPlaneMax = 80 * 1024;  // two workgroups per CU
template <int TH>
__host__ __device__ constexpr size_t perm_lds_bytes(uint32_t K, uint32_t NB) {   // the plane; then the tile's image in its place
  return PermGeom<512, 0, TH>::plane_bytes(K, NB) > 2 * 64 * TH * sizeof(double) ? PermGeom<512, 0, TH>::plane_bytes(K, NB)
                                                                                   : 2 * 64 * TH * sizeof(double);
}
template <int TH>
__global__ void __launch_bounds__(512, 4)
    schwinger_perm_kernel(uint32_t Mt, uint32_t Mx, const double2 *__restrict__ in, double2 *__restrict__ out, uint32_t tiles_x,
                          uint32_t K, uint32_t NB) {
  constexpr int NT = 512;
  using PG = PermGeom<NT, 0, TH>;
  extern __shared__ double lds[];
  uint32_t tile, b;
  perm_tile_of_workgroup(tile, b);
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * 64, j0 = ty * TH;
  double2 res[2][PG::NV];
  perm_sweeps<NT, 0, TH>(lds, in + (size_t)b * Mt * Mx, Mt, Mx, i0, j0, K, NB, res);
  double *th0 = lds, *th1 = lds + 64 * TH;
  __syncthreads();  // the plane is dead: the image takes its place
  perm_store_image<NT, 0, TH>(th0, th1, res);
  __syncthreads();
  double2 *dst = out + (size_t)b * Mt * Mx;
#pragma unroll
  for (int k = 0; k < 64 * TH / NT; ++k) {  // a wave writes a row of the tile
    const uint32_t v = threadIdx.x + k * NT, r = v / 64, c = v % 64;
    if (j0 + r < Mx && i0 + c < Mt)   // (an edge tile of a lattice the tiles do not divide: see schwinger_image_heat)
      store_streaming(&dst[(size_t)(j0 + r) * Mt + (i0 + c)], th0[v], th1[v]);
  }
}

// K overrelaxation sweeps in closed form, then the heat-bath sweep, the QoI and the write-out of schwinger_or_heat_kernel
// (the same tail, schwinger_image_heat): the whole draw of the reference's sampler (10 + 1 sweeps) is ONE launch with one
// round trip of the state through HBM.  LDS: tables + list | the plane, then the image in the same place.
template <int NT, bool STEP>
struct PermHeatGeom {
  using PG = PermGeom<NT, 2>;
  using OH = OrHeatGeom<1>;  // (pool and image sizes do not depend on K)
  static constexpr size_t lds_bytes(uint32_t K, uint32_t NB) {
    return OH::pool_bytes + (PG::plane_bytes(K, NB) > OH::image_bytes ? PG::plane_bytes(K, NB) : OH::image_bytes);
  }
};

template <int NT, bool STEP>
__global__ void __launch_bounds__(NT, 4)
    schwinger_perm_heat_kernel(uint32_t Mt, uint32_t Mx, double beta, const double2 *__restrict__ in, double2 *__restrict__ out,
                               uint32_t tiles_x, uint32_t K, uint32_t NB, RngKey key0, int qoi_op, double *__restrict__ qoi_partial,
                               const uint32_t *__restrict__ vs_table) {
  using PH = PermHeatGeom<NT, STEP>;
  using PG = typename PH::PG;
  using OH = typename PH::OH;
  constexpr int IW = OH::IW, IH = OH::IH;
  static_assert(PG::OW == IW && 2 * PG::HR == IH, "the closed-form stage fills the heat bath's image");
  extern __shared__ double lds[];
  __shared__ double qoi_red[NT / kWave];
  uint32_t tile, b;
  perm_tile_of_workgroup(tile, b);
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * 64, j0 = ty * 64;
  MLMCPI_STAMP(0);
  MLMCPI_STAMP_WHERE();
#ifdef MLMCPI_PRIO_A
  __builtin_amdgcn_s_setprio(MLMCPI_PRIO_A);
#endif
  // the sampler's table: one word per thread, fetched now, put down when the sweeps are done (nothing waits for it here;
  // staged at the start it cost a round trip in front of the plane's loads: 1 % of the launch)
  uint32_t tabw = 0;
  if (STEP && threadIdx.x < kVsTableBytes / 4) tabw = vs_table[threadIdx.x];
  HbPool hpool = HbPool::carve(lds, STEP ? 0u : OH::hb_pool_cap);
  double *th0 = lds + OH::pool_bytes / sizeof(double), *th1 = th0 + IW * IH;
  double2 res[2][PG::NV];
  perm_sweeps<NT, 2>(th0, in + (size_t)b * Mt * Mx, Mt, Mx, i0, j0, K, NB, res);
  MLMCPI_STAMP(3);  // K sweeps done
  VsPool<uint32_t> vpool = VsPool<uint32_t>::carve(lds, OH::pool_cap, nullptr);
  if (STEP) {
    if (threadIdx.x < kVsTableBytes / 4) reinterpret_cast<uint32_t *>(lds)[threadIdx.x] = tabw;
    if (threadIdx.x < 2) vpool.count[threadIdx.x] = 0;
    vpool.tab = VsTable::at(lds, vs_table);
  }
  __syncthreads();  // the plane is dead: the image takes its place
  perm_store_image<NT, 2>(th0, th1, res);
  __syncthreads();
  MLMCPI_STAMP(4);  // image down
#ifdef MLMCPI_PRIO_B
  __builtin_amdgcn_s_setprio(MLMCPI_PRIO_B);
#endif
  schwinger_image_heat<NT, STEP>(th0, th1, vpool, hpool, Mt, Mx, beta, out, i0, j0, b, tile, key0, qoi_op, qoi_partial, qoi_red);
}

// ---- GFF sweeps --------------------------------------------------------------------------------------
// Red/black order: (i+j) even, then odd.  gffaction.cc:33-42 (heat bath), :68-77 (overrelaxation);
// Delta is summed in the order of the reference's neighbour table (+i, -i, +j, -j).
// TWC x THC > 0: single-sweep launch on a lattice that the TWC x THC tiles divide and that is wider than a buffer: tile and
// buffer extents are compile-time constants, as in schwinger_sweep_kernel (same updates, bit-identical results).
template <bool HEAT, int NT, int TWC = 0, int THC = 0>
__global__ void __launch_bounds__(NT)
    gff_sweep_kernel(uint32_t Mt, uint32_t Mx, double mu2, const double *__restrict__ in, double *__restrict__ out,
                     TileGeom tg, uint32_t nsweeps_arg, uint32_t kinds, RngKey key0, int qoi_op = 0,
                     double *__restrict__ qoi_partial = nullptr) {
  extern __shared__ double lds[];
  __shared__ double qoi_red[NT / 64];
  constexpr bool FIXED = TWC > 0;
  const uint32_t nsweeps = FIXED ? 1u : nsweeps_arg;
  const uint32_t H = 2 * nsweeps;
  const uint32_t tile = blockIdx.x, b = blockIdx.y;
  const uint32_t ty = tile / tg.tiles_x, tx = tile - ty * tg.tiles_x;
  const uint32_t i0 = tx * (FIXED ? TWC : tg.TW), j0 = ty * (FIXED ? THC : tg.TH);
  const uint32_t ow = FIXED ? TWC : min(tg.TW, Mt - i0), oh = FIXED ? THC : min(tg.TH, Mx - j0);
  auto wrap = [&](uint32_t base, uint32_t off, uint32_t n) {  // lattice coordinate of a buffer coordinate
    if (FIXED) {
      const uint32_t v = base + off;
      return v >= n ? v - n : v;
    }
    return wrap_add(base, off, n);
  };
  const uint32_t bw = ow + 2 * H, bh = oh + 2 * H;
  double *phi = lds;
  double *nrm = lds + (size_t)bw * bh;  // HEAT only: the second normal of each Box-Muller pair, by cell
  const uint32_t sc = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt)) % Mt);
  const uint32_t sr = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx)) % Mx);
  const double *src = in + (size_t)b * Mt * Mx;
  RngKey key = key0;
  key.chain += b;
  const double inv_kappa = 1. / (4. + mu2), two_over_kappa = 2. / (4. + mu2), sigma = 1. / sqrt(4. + mu2);
  const PhiloxVKeys vk = philox_vkeys(key.k0, key.k1);

  stage_region<NT, (NT >= 1024 ? 3 : 5), double>(
      bh, bw, [&](uint32_t r, uint32_t c) { return src[(size_t)wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt)]; },
      [&](uint32_t r, uint32_t c, double v) { phi[r * bw + c] = v; });
  __syncthreads();

  for (uint32_t s = 0; s < nsweeps; ++s) {
    const bool heat = HEAT && (FIXED || ((kinds >> s) & 1u));
    RngKey skey = key;
    skey.step += s;
    for (uint32_t colour = 0; colour < 2; ++colour) {
      // One Philox call + one Box-Muller per vertex PAIR (l >> 1): the two vertices of a pair are horizontal
      // neighbours (c, c ^ 1), hence of opposite colour.  The colour-0 phase draws the pair and parks the
      // partner's normal in LDS; the colour-1 phase picks it up.  Only colour-1 cells whose partner sits in an
      // outermost buffer column (never updated, so nothing was parked) draw the pair themselves: exactly one
      // cell per row (column 1 or bw - 2).  They get a pass of their own, so that the waves of the main
      // colour-1 pass never execute the Philox + Box-Muller code (one boundary lane would drag its whole wave
      // through it).
      auto stencil = [&](uint32_t o) {
        double Delta = 0.0;
        Delta += phi[o + 1];
        Delta += phi[o - 1];
        Delta += phi[o + bw];
        Delta += phi[o - bw];
        return Delta;
      };
      auto draw_pair = [&](uint32_t r, uint32_t c, bool park) {  // returns this cell's normal
        const uint32_t ell = wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt);
        double n0, n1;
        rng_normals(skey, vk, ell >> 1, P_GFF_NORMAL, 0, n0, n1);
        if (park) nrm[r * bw + (c ^ 1u)] = (ell & 1u) ? n0 : n1;
        return (ell & 1u) ? n1 : n0;
      };
      // Update region of this phase: rows [r_lo, r_lo + nrow), columns [c_lo, c_lo + 2 nhalf), cells of the phase's
      // colour.  In general everything but the outermost ring of the buffer; the LAST sweep of a launch is cut down to
      // what is still read afterwards: colour 1 (last phase) the owned tile, colour 0 the tile plus one ring.
      const bool last = s + 1 == nsweeps;
      const uint32_t grow = colour == 0 ? 1u : 0u;  // rings around the owned tile in the last sweep
      const uint32_t r_lo = last ? H - grow : 1, nrow = last ? oh + 2 * grow : bh - 2;
      const uint32_t c_lo = last ? H - grow : 1, nhalf = last ? (ow + 2 * grow) / 2 : (bw - 2) / 2;
      auto column = [&](uint32_t r, uint32_t ci) { return c_lo + ((r + c_lo + colour) & 1u) + 2 * ci; };
      if (!heat) {
        for_region<NT>(nrow, nhalf, [&](uint32_t ri, uint32_t ci) {
          const uint32_t r = r_lo + ri;
          const uint32_t o = r * bw + column(r, ci);
          phi[o] = fma(two_over_kappa, stencil(o), -phi[o]);  // 2 Delta / kappa - phi without the fp64 division
        });
      } else if (colour == 0) {
        for_region<NT>(nrow, nhalf, [&](uint32_t ri, uint32_t ci) {
          const uint32_t r = r_lo + ri;
          const uint32_t c = column(r, ci);
          const uint32_t o = r * bw + c;
          phi[o] = fma(stencil(o), inv_kappa, sigma * draw_pair(r, c, true));
        });
      } else if (last) {
        // every colour-1 cell of the tile has its pair partner (c ^ 1, same row) inside the colour-0 region above
        for_region<NT>(nrow, nhalf, [&](uint32_t ri, uint32_t ci) {
          const uint32_t r = r_lo + ri;
          const uint32_t o = r * bw + column(r, ci);
          phi[o] = fma(stencil(o), inv_kappa, sigma * nrm[o]);
        });
      } else {
        for_region<NT>(nrow, nhalf, [&](uint32_t ri, uint32_t ci) {
          const uint32_t r = r_lo + ri;
          const uint32_t c = column(r, ci);
          if (c == 1 || c == bw - 2) return;  // boundary partners: next pass
          const uint32_t o = r * bw + c;
          phi[o] = fma(stencil(o), inv_kappa, sigma * nrm[o]);
        });
        for_region<NT>(bh - 2, 1, [&](uint32_t ri, uint32_t) {
          const uint32_t r = 1 + ri;
          const uint32_t c = (r & 1u) ? bw - 2 : 1;  // the colour-1 cell of this row next to an outermost column
          const uint32_t o = r * bw + c;
          phi[o] = fma(stencil(o), inv_kappa, sigma * draw_pair(r, c, false));
        });
      }
      __syncthreads();
    }
  }

  // Optional fused QoI of the final state (qoi/qft/qoi2dphisquared.cc:8-15): phi^2 summed over the owned sites while the
  // tile is in LDS, one partial per tile; lattice_finish_kernel sums them in tile order.
  double acc[1] = {0.0};
  double *dst = out + (size_t)b * Mt * Mx;
  for_region<NT>(oh, ow, [&](uint32_t r, uint32_t c) {
    const double v = phi[(r + H) * bw + (c + H)];
    dst[(size_t)(j0 + r) * Mt + (i0 + c)] = v;
    if (qoi_op) acc[0] += v * v;
  });
  if (qoi_op) {
    block_sum<1>(acc, qoi_red);
    if (threadIdx.x == 0) qoi_partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
  }
}

// ---- GFF overrelaxation, specialised --------------------------------------------------------------------
// Compile-time tile geometry, per-thread cell offsets computed once, immediate neighbour offsets:
// bit-identical to gff_sweep_kernel<false, 256> (same colour order and update regions).  The update is
// 4 adds and one fused multiply-add, so anything per-update besides the LDS traffic matters.
template <int TW, int TH, int K, int NT>
__global__ void __launch_bounds__(NT)
    gff_or_kernel(uint32_t Mt, uint32_t Mx, double mu2, const double *__restrict__ in, double *__restrict__ out,
                  uint32_t tiles_x) {
  constexpr int H = 2 * K, BW = TW + 2 * H, BH = TH + 2 * H, P = BW + 1;
  constexpr int NRR = BH - 2, NCC = (BW - 2) / 2, T = NRR * NCC, M = (T + NT - 1) / NT;
  extern __shared__ double lds[];
  double *phi = lds;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;  // see schwinger_or_kernel
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  const uint32_t sc = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt)) % Mt);
  const uint32_t sr = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx)) % Mx);
  const double *src = in + (size_t)b * Mt * Mx;
  const double two_over_kappa = 2. / (4. + mu2);

  stage_region<NT, 5, double>(
      BH, BW, [&](uint32_t r, uint32_t c) { return src[(size_t)wrap_add(sr, r, Mx) * Mt + wrap_add(sc, c, Mt)]; },
      [&](uint32_t r, uint32_t c, double v) { phi[r * P + c] = v; });

  // cell (r, c) of colour 0: r = 1 + ri, c = 1 + ((r + 1) & 1) + 2 ci; colour 1: the other cell of the pair
  int o0[M], o1[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int idx = tid + NT * m, ri = idx / NCC, ci = idx - ri * NCC, r = 1 + ri;
    o0[m] = r * P + 1 + ((r + 1) & 1) + 2 * ci;
    o1[m] = r * P + 1 + (r & 1) + 2 * ci;
  }
  __syncthreads();

  for (int s = 0; s < K; ++s) {
#pragma unroll
    for (int colour = 0; colour < 2; ++colour) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if ((m + 1) * NT <= T || (int)tid + NT * m < T) {
          const int o = colour ? o1[m] : o0[m];
          const uint32_t a = lds0 + (uint32_t)(o - P) * 8u;  // LDS byte address of the cell below
          double dn = lds_read_f64<0>(a), lf = lds_read_f64<(P - 1) * 8>(a), own = lds_read_f64<P * 8>(a);
          double rt = lds_read_f64<(P + 1) * 8>(a), up = lds_read_f64<2 * P * 8>(a), pad0 = 0.0, pad1 = 0.0;
          lds_wait7(dn, lf, own, rt, up, pad0, pad1);
          double Delta = 0.0;
          Delta += rt;
          Delta += lf;
          Delta += up;
          Delta += dn;
          phi[o] = fma(two_over_kappa, Delta, -own);  // 2 Delta / kappa - phi, without the fp64 division
        }
      }
      __syncthreads();
    }
  }

  double *dst = out + (size_t)b * Mt * Mx;
  for_region<NT>(TH, TW, [&](uint32_t r, uint32_t c) { dst[(size_t)(j0 + r) * Mt + (i0 + c)] = phi[(r + H) * P + (c + H)]; });
}

// ---- GFF overrelaxation, register-tiled ---------------------------------------------------------------------------
// Same idea as schwinger_or_patch_kernel: a thread keeps a 2 x 2 block of sites in registers for all K sweeps and LDS
// (four structure-of-arrays planes [row parity][column parity] over blocks) carries only the values that cross block
// boundaries: 4 reads + 2 writes per colour phase for 2 updates (24 B per update; gff_or_kernel moves 48 B).  Same
// updates, colour order and summation order (+i, -i, +j, -j): bit-identical to the kernels above.
template <int K>
__global__ void __launch_bounds__(1024)
    gff_or_patch_kernel(uint32_t Mt, uint32_t Mx, double mu2, const double *__restrict__ in, double *__restrict__ out,
                        uint32_t tiles_x) {
  constexpr int TW = 64, TH = 32, H = 2 * K, BW = TW + 2 * H, BH = TH + 2 * H, NPX = BW / 2, NPY = BH / 2;
  constexpr int NP = NPX * NPY;
  static_assert(NP <= 1024, "one 2 x 2 block per thread");
  extern __shared__ double lds[];
  auto plane = [&](int c, int a) { return lds + (c * 2 + a) * NP; };  // site (2 pi + a, 2 pj + c)
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  const bool active = tid < NP;
  const int pj = active ? (int)tid / NPX : 0, pi = active ? (int)tid - pj * NPX : 0;
  const double *src = in + (size_t)b * Mt * Mx;
  const double two_over_kappa = 2. / (4. + mu2);
  const uint32_t gi = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt) + 2 * pi) % Mt);
  const uint32_t gj0 = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx) + 2 * pj) % Mx);
  const uint32_t gj1 = gj0 + 1 == Mx ? 0 : gj0 + 1;
  double p[2][2] = {{0, 0}, {0, 0}};  // [c][a]
  if (active) {
    const double2 lo = *(const double2 *)(src + (size_t)gj0 * Mt + gi), hi = *(const double2 *)(src + (size_t)gj1 * Mt + gi);
    p[0][0] = lo.x; p[0][1] = lo.y; p[1][0] = hi.x; p[1][1] = hi.y;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a) plane(c, a)[tid] = p[c][a];
  }
  const int me = active ? (int)tid : 0;  // idle threads of the last wave: every index is entry 0, nothing is written
  const int dn = pj > 0 ? me - NPX : me, up = pj + 1 < NPY ? me + NPX : me;
  const int lf = pi > 0 ? me - 1 : me, rt = pi + 1 < NPX ? me + 1 : me;
  // sites on the buffer edge are updated from clamped neighbours instead of being left alone: see
  // schwinger_or_patch_kernel (their values never reach the owned tile)
  __syncthreads();

  for (int s = 0; s < K; ++s) {
    // colour 0: sites (a, c) = (0, 0) and (1, 1)
    // (no predicate and no zero fill in the loop: idle threads read entry 0 and compute on zeros)
    double e_lf = plane(0, 1)[lf], e_dn = plane(1, 0)[dn], e_rt = plane(1, 0)[rt], e_up = plane(0, 1)[up];
    {
      double Delta = 0.0;
      Delta += p[0][1]; Delta += e_lf; Delta += p[1][0]; Delta += e_dn;
      p[0][0] = fma(two_over_kappa, Delta, -p[0][0]);
      if (active) plane(0, 0)[me] = p[0][0];
    }
    {
      double Delta = 0.0;
      Delta += e_rt; Delta += p[1][0]; Delta += e_up; Delta += p[0][1];
      p[1][1] = fma(two_over_kappa, Delta, -p[1][1]);
      if (active) plane(1, 1)[me] = p[1][1];
    }
    __syncthreads();
    // colour 1: sites (1, 0) and (0, 1)
    e_rt = plane(0, 0)[rt]; e_dn = plane(1, 1)[dn]; e_lf = plane(1, 1)[lf]; e_up = plane(0, 0)[up];
    {
      double Delta = 0.0;
      Delta += e_rt; Delta += p[0][0]; Delta += p[1][1]; Delta += e_dn;
      p[0][1] = fma(two_over_kappa, Delta, -p[0][1]);
      if (active) plane(0, 1)[me] = p[0][1];
    }
    {
      double Delta = 0.0;
      Delta += p[1][1]; Delta += e_lf; Delta += e_up; Delta += p[0][0];
      p[1][0] = fma(two_over_kappa, Delta, -p[1][0]);
      if (active) plane(1, 0)[me] = p[1][0];
    }
    __syncthreads();
  }

  if (active && pi >= H / 2 && pi < (H + TW) / 2 && pj >= H / 2 && pj < (H + TH) / 2) {
    double *dst = out + (size_t)b * Mt * Mx;
    const size_t o0 = (size_t)(j0 + 2 * pj - H) * Mt + (i0 + 2 * pi - H);
    *(double2 *)(dst + o0) = make_double2(p[0][0], p[0][1]);
    *(double2 *)(dst + o0 + Mt) = make_double2(p[1][0], p[1][1]);
  }
}

// ---- GFF overrelaxation, 4 x 4 register blocks on 64 x 64 tiles ------------------------------------------------
// The construction of schwinger_or_block_kernel for the scalar field: a thread keeps 16 sites for all K sweeps, LDS
// carries the 12 sites on the rim of each block (TOP[a], BOT[a], LEFT[c], RIGHT[c], corners once), a colour phase reads
// the 8 neighbour values across the block's edges that belong to the other colour.  Redundancy (64 + 4K)^2 / 64^2
// (1.72 at K = 5) instead of 1.875 at K = 4 on 64 x 32 tiles, 1.75 LDS accesses per update instead of 3, three
// workgroups per CU.  Same sums in the same order as gff_or_patch_kernel: bit-identical.
// T: tile extent.  64 is the default; 32 x 32 tiles (r04) serve the lattices 64 x 64 tiles do not divide or that are too
// small for the fused launch (96 x 96: 339 -> see DESIGN 7, fast_path_cliff) -- the halo recomputation is 2.6 x at K = 5
// instead of 1.7 x, but the launches are bound by their passes over the state, not by the sweeps.
template <int K, int T = 64>
struct GffBlockGeom {
  static constexpr int TW = T, TH = T, PW = 4, PH = 4, H = 2 * K;
  static constexpr int BW = TW + 2 * H, BH = TH + 2 * H, NPX = BW / PW, NPY = BH / PH, NP = NPX * NPY;
  static constexpr int NT = (NP + 63) / 64 * 64;
  static constexpr int NPLANE = 2 * PW + 2 * (PH - 2);
  static constexpr size_t lds_bytes = (size_t)NPLANE * NP * sizeof(double);
  static constexpr int top(int a) { return a; }
  static constexpr int bot(int a) { return PW + a; }
  static constexpr int left(int c) { return c == 0 ? bot(0) : c == PH - 1 ? top(0) : 2 * PW + (c - 1); }
  static constexpr int right(int c) { return c == 0 ? bot(PW - 1) : c == PH - 1 ? top(PW - 1) : 2 * PW + (PH - 2) + (c - 1); }
};

// The buffer of geometry G (tile + halo G::H) into 4 x 4 register blocks, then KS <= G::H / 2 overrelaxation sweeps; ends
// behind the barrier of the last colour phase (the plane area is dead from there on).
template <class G, int KS>
__device__ __forceinline__ void gff_block_sweeps(double *lds, const double *__restrict__ src, uint32_t Mt, uint32_t Mx, double mu2,
                                                 uint32_t i0, uint32_t j0, double (&p)[G::PH][G::PW]) {
  constexpr int PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NPY = G::NPY, NP = G::NP;
  static_assert(2 * KS <= H, "a sweep costs two sites of halo");
  auto pl = [&](int p) { return lds + p * NP; };
  const uint32_t tid = threadIdx.x;
  const bool active = tid < NP;
  const int pj = active ? (int)tid / NPX : 0, pi = active ? (int)tid - pj * NPX : 0;
  const int me = active ? (int)tid : 0;  // idle threads of the last wave: every index is entry 0, nothing is written
  const int dn = pj > 0 ? me - NPX : me, up = pj + 1 < NPY ? me + NPX : me;
  const int lf = pi > 0 ? me - 1 : me, rt = pi + 1 < NPX ? me + 1 : me;
  const double two_over_kappa = 2. / (4. + mu2);
  // p: [c][a] = site (PW pi + a, PH pj + c)
  {
    uint32_t gi[PW / 2], gj[PH];  // H is even: a pair of sites (gi, gi + 1) never straddles the wrap
    gi[0] = (uint32_t)(((uint64_t)i0 + Mt - (H % Mt) + PW * pi) % Mt);
    gj[0] = (uint32_t)(((uint64_t)j0 + Mx - (H % Mx) + PH * pj) % Mx);
#pragma unroll
    for (int a = 1; a < PW / 2; ++a) gi[a] = gi[a - 1] + 2 == Mt ? 0 : gi[a - 1] + 2;
#pragma unroll
    for (int c = 1; c < PH; ++c) gj[c] = gj[c - 1] + 1 == Mx ? 0 : gj[c - 1] + 1;
#pragma unroll
    for (int c = 0; c < PH; ++c)
#pragma unroll
      for (int a = 0; a < PW; a += 2) {
        const double2 v = active ? *(const double2 *)(src + (size_t)gj[c] * Mt + gi[a / 2]) : make_double2(0, 0);
        p[c][a] = v.x;
        p[c][a + 1] = v.y;
      }
  }
  auto publish = [&](int a, int c, double v) {  // rim sites: a corner belongs to a row and a column, stored once
    const int p1 = c == PH - 1 ? G::top(a) : c == 0 ? G::bot(a) : -1;
    const int p2 = a == 0 ? G::left(c) : a == PW - 1 ? G::right(c) : -1;
    if (!active) return;
    if (p1 >= 0) pl(p1)[me] = v;
    if (p2 >= 0 && p2 != p1) pl(p2)[me] = v;
  };
#pragma unroll
  for (int c = 0; c < PH; ++c)
#pragma unroll
    for (int a = 0; a < PW; ++a) publish(a, c, p[c][a]);
  __syncthreads();

  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int col = 0; col < 2; ++col) {
      // neighbour values across the block's edges (they have the other colour: unchanged during this phase)
      double e_lf[PH], e_rt[PH], e_dn[PW], e_up[PW];
#pragma unroll
      for (int c = 0; c < PH; ++c) {
        if (((0 + c) & 1) == col) e_lf[c] = pl(G::right(c))[lf];
        if (((PW - 1 + c) & 1) == col) e_rt[c] = pl(G::left(c))[rt];
      }
#pragma unroll
      for (int a = 0; a < PW; ++a) {
        if (((a + 0) & 1) == col) e_dn[a] = pl(G::top(a))[dn];
        if (((a + PH - 1) & 1) == col) e_up[a] = pl(G::bot(a))[up];
      }
#pragma unroll
      for (int c = 0; c < PH; ++c)
#pragma unroll
        for (int a = 0; a < PW; ++a) {
          if (((a + c) & 1) != col) continue;
          // gffaction.cc:68-77, Delta summed in the order of the reference's neighbour table (+i, -i, +j, -j)
          double Delta = 0.0;
          Delta += a + 1 < PW ? p[c][a + 1 < PW ? a + 1 : 0] : e_rt[c];
          Delta += a > 0 ? p[c][a > 0 ? a - 1 : 0] : e_lf[c];
          Delta += c + 1 < PH ? p[c + 1 < PH ? c + 1 : 0][a] : e_up[a];
          Delta += c > 0 ? p[c > 0 ? c - 1 : 0][a] : e_dn[a];
          p[c][a] = fma(two_over_kappa, Delta, -p[c][a]);
          publish(a, c, p[c][a]);
        }
      __syncthreads();
    }
  }
}

template <int K, int T = 64>
__global__ void __launch_bounds__((GffBlockGeom<K, T>::NT))
    gff_or_block_kernel(uint32_t Mt, uint32_t Mx, double mu2, const double *__restrict__ in, double *__restrict__ out,
                        uint32_t tiles_x) {
  using G = GffBlockGeom<K, T>;
  constexpr int TW = G::TW, TH = G::TH, PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NP = G::NP;
  static_assert(PW == 4 && PH == 4, "the write-back moves 4 sites per block row");
  extern __shared__ double lds[];
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  double p[PH][PW];
  gff_block_sweeps<G, K>(lds, in + (size_t)b * Mt * Mx, Mt, Mx, mu2, i0, j0, p);

  // Owned sites: buffer columns [H, H + TW), rows [H, H + TH), written back through a per-wave transposition in LDS (the
  // planes are dead after the last barrier) so that a wave instruction covers whole rows: per block row the owners put
  // their 4 sites down as two double2, and lane l writes the pair (l & 1) of block 32 i + (l >> 1), i = 0, 1.
  const uint32_t wave0 = tid & ~63u, lane = tid & 63u;
  double2 *stage = reinterpret_cast<double2 *>(lds) + (wave0 / 64) * 128;
  double *dst = out + (size_t)b * Mt * Mx;
  int uq[2], ur[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int bt = (int)wave0 + 32 * i + (int)(lane >> 1);
    const int bj = bt / NPX, bi = bt - bj * NPX;
    uq[i] = PW * bi + 2 * (int)(lane & 1) - H;  // even, and TW is even: the pair is owned as a whole or not at all
    ur[i] = PH * bj - H;
    if (bt >= NP || uq[i] >= TW) uq[i] = -1;
  }
#pragma unroll
  for (int c = 0; c < PH; ++c) {
    stage[2 * lane] = make_double2(p[c][0], p[c][1]);
    stage[2 * lane + 1] = make_double2(p[c][2], p[c][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double2 w = stage[64 * i + lane];
      const int r = ur[i] + c;
      // (r05: an edge tile of a lattice the tiles do not divide reaches beyond it -- periodic images of sites other tiles
      // own, computed like any halo, not written; Mt is even, so a pair lies inside or outside as a whole)
      if (uq[i] >= 0 && r >= 0 && r < TH && j0 + (uint32_t)r < Mx && i0 + (uint32_t)uq[i] < Mt)
        *(double2 *)(dst + (size_t)(j0 + r) * Mt + (i0 + uq[i])) = w;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- GFF: K overrelaxation sweeps and the heat-bath sweep behind them in one launch ----------------------------------
// The construction of schwinger_or_heat_kernel for the scalar field: gff_block_sweeps on the geometry with halo 2K + 2,
// then the field on the tile and two rings as an LDS image (with the plane of parked Box-Muller partners behind it), the
// heat-bath sweep of gff_sweep_kernel<true, 256, 64, 32> in its pruned last-sweep form -- same cells, same Philox
// words, same arithmetic: bit-identical -- the phi^2 sum and the write-out.  The second normal of a Box-Muller pair is
// not parked in LDS here: the thread that draws the pair for a colour-0 cell (r, c) also updates the colour-1 cell
// (r, c ^ 1) and keeps its normal in a register (the two phases walk the same compile-time index space), so the image is
// the field alone and three workgroups fit a CU.
template <int K, int T = 64>
struct GffHeatGeom {
  using G = GffBlockGeom<K + 1, T>;
  static constexpr int NT = G::NT, HB = 2, IW = G::TW + 2 * HB, IH = G::TH + 2 * HB;
  static constexpr size_t image_bytes = (size_t)IW * IH * sizeof(double);
  static constexpr size_t lds_bytes = G::lds_bytes > image_bytes ? G::lds_bytes : image_bytes;
};

template <int K, int T = 64>
__global__ void __launch_bounds__((GffHeatGeom<K, T>::NT), 4)
    gff_or_heat_kernel(uint32_t Mt, uint32_t Mx, double mu2, const double *__restrict__ in, double *__restrict__ out,
                       uint32_t tiles_x, RngKey key0, int qoi_op, double *__restrict__ qoi_partial) {
  using OH = GffHeatGeom<K, T>;
  using G = typename OH::G;
  constexpr int NT = OH::NT, TW = G::TW, TH = G::TH, PW = G::PW, PH = G::PH, H = G::H, NPX = G::NPX, NP = G::NP;
  constexpr int HB = OH::HB, IW = OH::IW, IH = OH::IH, O = H - HB;  // image (0, 0) = buffer (O, O)
  extern __shared__ double lds[];
  __shared__ double qoi_red[NT / kWave];
  const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint32_t i0 = tx * TW, j0 = ty * TH;
  double p[PH][PW];
  gff_block_sweeps<G, K>(lds, in + (size_t)b * Mt * Mx, Mt, Mx, mu2, i0, j0, p);

  double *phi = lds;
  if (tid < NP) {
    const int pj = (int)tid / NPX, pi = (int)tid - pj * NPX;
#pragma unroll
    for (int c = 0; c < PH; ++c) {
      const int r = PH * pj + c - O;
      if (r < 0 || r >= IH) continue;
#pragma unroll
      for (int a = 0; a < PW; ++a) {
        const int q = PW * pi + a - O;
        if (q >= 0 && q < IW) phi[r * IW + q] = p[c][a];
      }
    }
  }
  __syncthreads();

  // the heat-bath sweep: the last-sweep regions of gff_sweep_kernel with H = HB, bw = IW, oh = TH, ow = TW
  constexpr uint32_t bw = IW;
  const uint32_t sc = i0 >= (uint32_t)HB ? i0 - HB : i0 + Mt - HB;  // lattice column of image column 0 (even)
  const uint32_t sr = j0 >= (uint32_t)HB ? j0 - HB : j0 + Mx - HB;
  auto wrap = [](uint32_t base, uint32_t off, uint32_t n) {
    const uint32_t v = base + off;
    return v >= n ? v - n : v;
  };
  RngKey skey = key0;
  skey.chain += b;
  const double inv_kappa = 1. / (4. + mu2), sigma = 1. / sqrt(4. + mu2);
  // the four neighbours as single ds_read_b64 at immediate offsets from the address of phi[o - bw] (the compiler pairs
  // phi[o - 1], phi[o + 1] into a ds_read2_b64: 8 LDS cycles against 2 + 2, MI355X_MICROARCH.md); summed in the order of
  // the reference's neighbour table (+i, -i, +j, -j)
  const uint32_t lds_phi = (uint32_t)(uintptr_t)phi;
  auto stencil_load = [&](uint32_t o, double (&v)[4]) {
    const uint32_t a = lds_phi + (o - bw) * 8u;
    v[0] = lds_read_f64<bw * 8 + 8>(a);
    v[1] = lds_read_f64<bw * 8 - 8>(a);
    v[2] = lds_read_f64<2 * bw * 8>(a);
    v[3] = lds_read_f64<0>(a);
  };
  auto stencil_sum = [&](double (&v)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
    double Delta = 0.0;
    Delta += v[0];
    Delta += v[1];
    Delta += v[2];
    Delta += v[3];
    return Delta;
  };
  // colour 0: the tile plus one ring, (TH + 2) x (TW + 2) / 2 cells; cell idx = tid + k NT of the thread, k < CELLS
  constexpr uint32_t nrow = TH + 2, nhalf = (TW + 2) / 2, total = nrow * nhalf, CELLS = (total + NT - 1) / NT;
  double partner[CELLS];  // the normal of (r, c ^ 1), the colour-1 cell of the same Box-Muller pair
  const PhiloxVKeys vk = philox_vkeys(skey.k0, skey.k1);
  // one colour-0 cell (image row r, column c, offset o, lattice site ell) and the parked normal of its pair
  auto cell0 = [&](uint32_t o, uint32_t ell, double &parked) {
    double n0, n1, nb[4];
    stencil_load(o, nb);   // in flight under the Philox call and the Box-Muller transform
    rng_normals(skey, vk, ell >> 1, P_GFF_NORMAL, 0, n0, n1);
    parked = (ell & 1u) ? n0 : n1;
    phi[o] = fma(stencil_sum(nb), inv_kappa, sigma * ((ell & 1u) ? n1 : n0));
  };
  auto cell1 = [&](uint32_t o, double parked) {
    double nb[4];
    stencil_load(o, nb);
    phi[o] = fma(stencil_sum(nb), inv_kappa, sigma * parked);
  };
  if constexpr (NT == 512 && T == 64) {
    // r05: cells by a closed-form map instead of by linear index (a division by 33, the parity of the row, two wraps and a
    // multiplication per cell and colour: ~29 of the ~240 vector instructions of a pair): a wave takes two rows x 32 cells
    // per round -- lane l: row 1 + 2 (w + 8 k) + (l >> 5), column 1 + (l >> 5) + 2 (l & 31), the same in every round --, four
    // rounds cover rows 1 .. 64; rows 65, 66 and the 33rd cell of every row (130 cells) are a fifth round of 130 threads,
    // as many as the linear hand-out leaves for its last.  Which lane draws a pair does not enter the result.
    static_assert(CELLS == 5 && nhalf == 33 && nrow == 66, "64 x 64 tile, 512 threads");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave, rr = lane >> 5;
    const uint32_t r0 = 1 + 2 * wave + rr, c0 = 1 + rr + 2 * (lane & 31u);
    const uint32_t colw = wrap(sc, c0, Mt), mxmt = Mx * Mt;
    uint32_t rowmt = wrap(sr, r0, Mx) * Mt, o = r0 * bw + c0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      cell0(o, rowmt + colw, partner[k]);
      o += 16 * bw;
      rowmt += 16 * Mt;
      rowmt = min(rowmt, rowmt - mxmt);   // (one wrap: the image is no taller than the lattice)
    }
    // the fifth round: t < 66: rows 65, 66, cell t % 33; 66 <= t < 130: row 1 + (t - 66), the 33rd cell
    const bool extra = tid < 130;
    const uint32_t xr = tid < 66 ? 65 + tid / 33 : 1 + (tid - 66), xci = tid < 66 ? tid % 33 : 32;
    const uint32_t xc = HB - 1 + ((xr + HB - 1) & 1u) + 2 * xci, xo = xr * bw + xc;
    partner[4] = 0.0;
    if (extra) cell0(xo, wrap(sr, xr, Mx) * Mt + wrap(sc, xc, Mt), partner[4]);
    __syncthreads();
    // colour 1: the cell (r, c ^ 1) of every colour-0 cell, where that lies inside the tile
    const uint32_t c1 = c0 ^ 1u;
    const bool col_in = c1 >= (uint32_t)HB && c1 < (uint32_t)(HB + TW);
    o = r0 * bw + c1;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t r = r0 + 16 * k;
      if (col_in && r >= (uint32_t)HB && r < (uint32_t)(HB + TH)) cell1(o, partner[k]);
      o += 16 * bw;
    }
    const uint32_t xc1 = xc ^ 1u;
    if (extra && xr >= (uint32_t)HB && xr < (uint32_t)(HB + TH) && xc1 >= (uint32_t)HB && xc1 < (uint32_t)(HB + TW))
      cell1(xr * bw + xc1, partner[4]);
    __syncthreads();
  } else {
#pragma unroll
  for (uint32_t k = 0; k < CELLS; ++k) {
    const uint32_t idx = tid + k * NT;
    partner[k] = 0.0;
    if (idx >= total) continue;
    const uint32_t ri = idx / nhalf, r = HB - 1 + ri;
    const uint32_t c = HB - 1 + ((r + HB - 1) & 1u) + 2 * (idx - ri * nhalf);
    cell0(r * bw + c, wrap(sr, r, Mx) * Mt + wrap(sc, c, Mt), partner[k]);
  }
  __syncthreads();
  // colour 1: the tile; the cell (r, c ^ 1) of every colour-0 cell, where that lies inside the tile
#pragma unroll
  for (uint32_t k = 0; k < CELLS; ++k) {
    const uint32_t idx = tid + k * NT;
    if (idx >= total) continue;
    const uint32_t ri = idx / nhalf, r = HB - 1 + ri;
    const uint32_t c = (HB - 1 + ((r + HB - 1) & 1u) + 2 * (idx - ri * nhalf)) ^ 1u;
    if (r < (uint32_t)HB || r >= (uint32_t)(HB + TH) || c < (uint32_t)HB || c >= (uint32_t)(HB + TW)) continue;
    cell1(r * bw + c, partner[k]);
  }
  __syncthreads();
  }

  double acc[1] = {0.0};
  double *dst = out + (size_t)b * Mt * Mx;
  for_region<NT>(TH, TW, [&](uint32_t r, uint32_t c) {
    if (j0 + r >= Mx || i0 + c >= Mt) return;   // (the part of an edge tile beyond the lattice: see gff_or_block_kernel)
    const double v = phi[(r + HB) * bw + (c + HB)];
    dst[(size_t)(j0 + r) * Mt + (i0 + c)] = v;
    if (qoi_op) acc[0] += v * v;
  });
  if (qoi_op) {
    block_sum<1>(acc, qoi_red);
    if (threadIdx.x == 0) qoi_partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
  }
}

// ---- streaming kernels: evaluate, force, QoI ----------------------------------------------------------
enum LatOp { L_GFF_ENERGY = 0, L_PHI2 = 1, L_SCHW_ENERGY = 2, L_PLAQ = 3, L_CHARGE = 4 };

__device__ __forceinline__ double plaquette_angle(const double2 *t, uint32_t Mt, uint32_t Mx, uint32_t i, uint32_t j) {
  const uint32_t ip = (i + 1 == Mt) ? 0 : i + 1, jp = (j + 1 == Mx) ? 0 : j + 1;
  // theta(i,j,0) + theta(i+1,j,1) - theta(i,j+1,0) - theta(i,j,1)   (quenchedschwingeraction.cc:14-17)
  const double2 here = t[(size_t)j * Mt + i];
  return here.x + t[(size_t)j * Mt + ip].y - t[(size_t)jp * Mt + i].x - here.y;
}

// grid (nrows_blocks, B): each workgroup strides over lattice rows j
template <int OP>
__global__ void __launch_bounds__(256) lattice_reduce_kernel(uint32_t Mt, uint32_t Mx, double mu2,
                                                             const double *__restrict__ state,
                                                             double *__restrict__ partial) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.y;
  double acc[1] = {0.0};
  if (OP == L_GFF_ENERGY || OP == L_PHI2) {
    const double *phi = state + (size_t)b * Mt * Mx;
    const double kappa = 4. + mu2;
    for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x) {
      const uint32_t jm = j == 0 ? Mx - 1 : j - 1, jp = j + 1 == Mx ? 0 : j + 1;
      for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
        const double v = phi[(size_t)j * Mt + i];
        if (OP == L_PHI2) {
          acc[0] += v * v;
        } else {  // gffaction.cc:15-23
          const uint32_t im = i == 0 ? Mt - 1 : i - 1, ip = i + 1 == Mt ? 0 : i + 1;
          double loc = kappa * v;
          loc -= phi[(size_t)j * Mt + ip];
          loc -= phi[(size_t)j * Mt + im];
          loc -= phi[(size_t)jp * Mt + i];
          loc -= phi[(size_t)jm * Mt + i];
          acc[0] += v * loc;
        }
      }
    }
  } else {
    const double2 *t = (const double2 *)state + (size_t)b * Mt * Mx;
    for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x)
      for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
        const double th = plaquette_angle(t, Mt, Mx, i, j);
        if (OP == L_SCHW_ENERGY) acc[0] += 1. - cos_reduced(th);
        if (OP == L_PLAQ) acc[0] += cos_reduced(th);
        if (OP == L_CHARGE) acc[0] += mod_2pi(th);
      }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
}

// Plaquette reductions (Schwinger energy, average plaquette, topological charge), one pass with ONE load per site.
// Grid (bands, B): a workgroup walks a band of consecutive rows bottom-up; a thread owns the columns tid + 256 c and
// keeps the current row of its columns in registers, so theta(i, j+1, 0) of this row is the `here` of the next one;
// theta(i+1, j, 1) comes from the neighbouring lane (the last lane of a wave loads it).  The generic kernel above issues
// three 16-byte loads per plaquette and runs at ~2.9 TB/s; this one is bound by the 16 B per site it has to read.
template <int OP, int NC>
__global__ void __launch_bounds__(256) schwinger_reduce_band_kernel(uint32_t Mt, uint32_t Mx, uint32_t rows_per_band,
                                                                    const double2 *__restrict__ state,
                                                                    double *__restrict__ partial) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.y, j0 = blockIdx.x * rows_per_band;
  const uint32_t j1 = min(j0 + rows_per_band, Mx);
  const double2 *t = state + (size_t)b * Mt * Mx;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  double2 cur[NC], nxt[NC];
  uint32_t col[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    col[c] = threadIdx.x + 256u * c;
    cur[c] = col[c] < Mt ? t[(size_t)j0 * Mt + col[c]] : make_double2(0., 0.);
  }
  double acc[1] = {0.0};
  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t jp = j + 1 == Mx ? 0 : j + 1;
    double edge[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      nxt[c] = col[c] < Mt ? t[(size_t)jp * Mt + col[c]] : make_double2(0., 0.);
      // the right neighbour of a wave's last lane (or of the last column) lives in another wave / at column 0
      const uint32_t ip = col[c] + 1 == Mt ? 0 : col[c] + 1;
      edge[c] = (col[c] < Mt && (lane == kWave - 1 || col[c] + 1 == Mt)) ? t[(size_t)j * Mt + ip].y : 0.0;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double from_lane = __shfl_down(cur[c].y, 1, kWave);
      const double right = (lane == kWave - 1 || col[c] + 1 == Mt) ? edge[c] : from_lane;
      if (col[c] < Mt) {
        // theta(i,j,0) + theta(i+1,j,1) - theta(i,j+1,0) - theta(i,j,1)   (quenchedschwingeraction.cc:14-17)
        const double th = cur[c].x + right - nxt[c].x - cur[c].y;
        if (OP == L_SCHW_ENERGY) acc[0] += 1. - cos_reduced(th);
        if (OP == L_PLAQ) acc[0] += cos_reduced(th);
        if (OP == L_CHARGE) acc[0] += mod_2pi(th);
      }
      cur[c] = nxt[c];
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
}

__global__ void __launch_bounds__(256) lattice_finish_kernel(const double *__restrict__ partial, uint32_t nsplit, uint32_t B,
                                                              int op, double scale, double *__restrict__ out,
                                                              double *__restrict__ acc = nullptr) {
  // one wave per chain: lane l sums partials l, l + 64, ... in order, then a fixed shuffle tree -- the result depends on
  // nsplit only, never on the launch.  acc != NULL: stats->record_sample of the value as well (stats_accumulate_kernel's
  // recurrence), for callers that would launch that next.
  const uint32_t b = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x % 64;
  if (b >= B) return;
  double s = 0.0;
  for (uint32_t k = lane; k < nsplit; k += 64) s += partial[(size_t)b * nsplit + k];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    const double v = (op == L_CHARGE) ? (1. / (4. * kPi * kPi)) * s * s : scale * s;  // qoi2dsusceptibility.cc:26
    out[b] = v;
    if (acc) {
      double *a = acc + 5 * (size_t)b;
      a[0] += 1.0;
      a[1] += v;
      a[2] += v * v;
      a[3] += v * v * v;
      a[4] += v * v * v * v;
    }
  }
}

// gffaction.cc:80-94
__global__ void __launch_bounds__(256) gff_force_kernel(uint32_t Mt, uint32_t Mx, double mu2,
                                                        const double *__restrict__ phi_all, double *__restrict__ f_all) {
  const uint32_t b = blockIdx.y;
  const double *phi = phi_all + (size_t)b * Mt * Mx;
  double *f = f_all + (size_t)b * Mt * Mx;
  const double kappa = 4. + mu2;
  for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x) {
    const uint32_t jm = j == 0 ? Mx - 1 : j - 1, jp = j + 1 == Mx ? 0 : j + 1;
    for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
      const uint32_t im = i == 0 ? Mt - 1 : i - 1, ip = i + 1 == Mt ? 0 : i + 1;
      double m = kappa * phi[(size_t)j * Mt + i];
      m -= phi[(size_t)j * Mt + ip];
      m -= phi[(size_t)j * Mt + im];
      m -= phi[(size_t)jp * Mt + i];
      m -= phi[(size_t)jm * Mt + i];
      f[(size_t)j * Mt + i] = m;
    }
  }
}

// Gather form of quenchedschwingeraction.cc:68-89: the reference scatters +-beta sin(theta_P) of
// plaquette (i,j) onto its four links; link (i,j,0) therefore receives F(i,j) - F(i,j-1) and link
// (i,j,1) receives F(i-1,j) - F(i,j) (each a two-term sum, so the value is order independent).
//
// One sine per plaquette.  (Until r03 every thread computed the three plaquettes its two links touch -- three sines per
// site, 0.37 ms for 1024^2 x 32 = 0.36 of the HBM roofline for a kernel that reads and writes the state once.)  A WAVE
// walks up a band of rows with 64 consecutive columns: lane l holds column (62 tile + l - 1) mod Mt, takes theta_1 of the
// column to its right from lane l + 1 and F of the column to its left from lane l - 1 (DPP rotations: no LDS, no barrier),
// and keeps F of the row below in a register.  Lanes 1 .. 62 emit; lane 0 only supplies F, lane 63 only theta_1: tiles step
// by 62 columns, 3 % redundant loads and sines, and nothing at the edge of a wave is special.  Rows are loaded two ahead
// of their use.  Same arithmetic per plaquette as before (same sum order): bit-identical forces.
#ifndef MLMCPI_FORCE_ROWS
#define MLMCPI_FORCE_ROWS 128
#endif
constexpr uint32_t kForceCols = 62, kForceRows = MLMCPI_FORCE_ROWS;
__host__ __device__ inline uint32_t force_waves(uint32_t Mt, uint32_t Mx) {
  return ((Mt + kForceCols - 1) / kForceCols) * ((Mx + kForceRows - 1) / kForceRows);
}
// emit(j, i, F_0, F_1): force on the two links of vertex (i, j)
template <class Emit>
__device__ __forceinline__ void schwinger_force_band(const double2 *__restrict__ t, uint32_t Mt, uint32_t Mx, double coupling,
                                                     uint32_t wave_id, Emit emit) {
  const uint32_t tiles = (Mt + kForceCols - 1) / kForceCols;
  const uint32_t band = wave_id / tiles, tile = wave_id - band * tiles, lane = threadIdx.x & (kWave - 1);
  const uint32_t col = tile * kForceCols + lane;                       // column + 1, not wrapped
  const uint32_t i = (uint32_t)(((uint64_t)col + Mt - 1) % Mt);
  const bool owner = lane >= 1 && lane <= kForceCols && col <= Mt;     // col - 1 < Mt: not a column of the next lap
  const uint32_t jb = band * kForceRows, je = min(jb + kForceRows, Mx);
  auto up_of = [&](uint32_t j) { return j + 1 == Mx ? 0u : j + 1; };
  auto F_of = [&](const double2 &a, const double2 &above) {
    // theta(i,j,0) + theta(i+1,j,1) - theta(i,j+1,0) - theta(i,j,1)   (quenchedschwingeraction.cc:14-17)
    return coupling * sin_reduced(a.x + wave_rotate_down(a.y) - above.x - a.y);
  };
  // Rows are loaded four ahead of their use, at the TOP of an iteration (vmcnt counts stores too and retires in order, so
  // waiting for a row implies waiting for every store issued before its load).  Measured: 0.234 ms with two rows of
  // lookahead as with four, bands of 32 rows; 0.228 ms with bands of 128 (fewer band edges); EXPERIMENTS 1.6.
  const uint32_t jm = jb == 0 ? Mx - 1 : jb - 1;
  uint32_t jn = jb;
  auto next_row = [&]() {   // (up to four rows past the band at its end: valid rows, not used -- guarding the load cost 5 %)
    jn = up_of(jn);
    return t[(size_t)jn * Mt + i];
  };
  const double2 below = t[(size_t)jm * Mt + i];
  double2 here = t[(size_t)jb * Mt + i], above = next_row(), ahead1 = next_row(), ahead2 = next_row();
  double F_below = F_of(below, here);
  for (uint32_t j = jb; j < je; ++j) {
    const double2 ahead3 = next_row();
    const double F = F_of(here, above);
    const double F_left = wave_rotate_up(F);
    if (owner) emit(j, i, F - F_below, F_left - F);
    F_below = F;
    here = above;
    above = ahead1;
    ahead1 = ahead2;
    ahead2 = ahead3;
  }
}

// grid (ceil(force_waves / 4), B)
__global__ void __launch_bounds__(256) schwinger_force_kernel(uint32_t Mt, uint32_t Mx, double beta,
                                                              const double2 *__restrict__ t_all,
                                                              double2 *__restrict__ f_all) {
  const uint32_t b = blockIdx.y, wave_id = blockIdx.x * 4 + threadIdx.x / kWave;
  if (wave_id >= force_waves(Mt, Mx)) return;   // (a whole wave)
  double2 *f = f_all + (size_t)b * Mt * Mx;
  schwinger_force_band(t_all + (size_t)b * Mt * Mx, Mt, Mx, beta, wave_id,
                       // (non-temporal stores, r05: 0.227 -> 0.221 ms over three same-box pairs, 0.59 -> 0.61 of 8 TB/s by the floor bytes)
                       [&](uint32_t j, uint32_t i, double f0, double f1) { store_streaming(&f[(size_t)j * Mt + i], f0, f1); });
}

__global__ void __launch_bounds__(256) lattice_init_kernel(int kind, uint32_t n, RngKey key0, double *__restrict__ x) {
  const uint32_t b = blockIdx.y;
  RngKey key = key0;
  key.chain += b;
  double *xb = x + (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) {
    if (kind == MLMCPI_SCHWINGER) {
      double u, v;
      rng_uniforms(key, l, P_INIT, 0, u, v);
      xb[l] = -kPi + 2.0 * kPi * u;
    } else {
      xb[l] = rng_normal0(key, l, P_INIT, 0);
    }
  }
}

// Statistics::record_sample with its autocorrelation window (common/statistics.cc:4-27), one chain per thread: per chain
// [n, a1 = running average, S_0 .. S_{W-1} = running averages of Q_j Q_{j-k}, head, ring of the last W values].  The same
// recurrences as the reference: a1 <- ((n - 1) a1 + Q) / n; S_k <- ((N_k - 1) S_k + Q Q_{-k}) / N_k, N_k = n - k, over the k
// the window holds.  tau_int = max(1, 1 + 2 sum_{k >= 1} (1 - k / n) (S_k - a1^2) / (S_0 - a1^2)) is left to the caller
// (:38-61): the multilevel driver reads it between draws (montecarlomultilevel.cc:170-190).
__global__ void stats_window_record_kernel(double *__restrict__ state, const double *__restrict__ q, uint32_t B, uint32_t W) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double *st = state + (size_t)b * (2 * W + 3);
  double *S = st + 2, *ring = st + 3 + W;
  const double Q = q[b];
  const double n = st[0] + 1.0;
  uint32_t head = (uint32_t)st[2 + W];   // slot of the most recent value
  head = head + 1 == W ? 0 : head + 1;
  ring[head] = Q;
  st[2 + W] = (double)head;
  st[0] = n;
  st[1] = ((n - 1.0) * st[1] + Q) / n;
  const uint32_t filled = n < (double)W ? (uint32_t)n : W;
  uint32_t slot = head;
  for (uint32_t k = 0; k < filled; ++k) {
    const double Nk = n - (double)k;
    S[k] = ((Nk - 1.0) * S[k] + Q * ring[slot]) / Nk;
    slot = slot == 0 ? W - 1 : slot - 1;
  }
}

// packed per-chain sums for the cross-rank reduction: [n, sum q, sum q^2, sum q^3, sum q^4]
__global__ void stats_accumulate_kernel(double *__restrict__ acc, const double *__restrict__ q, uint32_t B) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double v = q[b];
  double *a = acc + 5 * (size_t)b;
  a[0] += 1.0;
  a[1] += v;
  a[2] += v * v;
  a[3] += v * v * v;
  a[4] += v * v * v * v;
}

// ---- site-at-a-time updates: Action::heatbath_update / overrelaxation_update(state, l), action/action.hh:73-96 -----------
// gffaction.cc:33-42,68-77; quenchedschwingeraction.cc:25-65.  One thread per chain walks the site list in order (the
// reference's own sequential semantics: every update sees the ones before it), straight on the state in global memory.
// Same arithmetic and the same random numbers -- Philox (site, chain, step) -- as the sweep kernels, so the sites of a
// colour class visited in any order with the sweep's step reproduce that colour phase of the sweep.
template <bool SCHW>
__global__ void __launch_bounds__(64)
    lattice_site_update_kernel(uint32_t Mt, uint32_t Mx, double coupling, double *__restrict__ state, uint32_t B,
                               const uint32_t *__restrict__ sites, uint32_t n, uint32_t single, int heat, RngKey key0,
                               const uint32_t *__restrict__ vs_table) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  RngKey key = key0;
  key.chain += b;
  if (SCHW) {
    double *th = state + (size_t)b * 2 * Mt * Mx;
    auto link = [&](uint32_t i, uint32_t j, uint32_t mu) -> double & { return th[2 * (Mt * j + i) + mu]; };
    const bool step = 2. * coupling <= kVsKappaMax;
    const VsTable tab = VsTable::in_global(vs_table);
    for (uint32_t q = 0; q < n; ++q) {
      const uint32_t l = sites ? sites[q] : single;
      const uint32_t mu = l & 1u, v = l >> 1, j = v / Mt, i = v - j * Mt;
      const uint32_t ip = i + 1 == Mt ? 0 : i + 1, im = i == 0 ? Mt - 1 : i - 1, jp = j + 1 == Mx ? 0 : j + 1, jm = j == 0 ? Mx - 1 : j - 1;
      double tp, tm;  // staple sums, unwrapped (quenchedschwingeraction.cc:25-43; same sums as schwinger_sweep_kernel)
      if (mu == 0) {
        tp = link(i, jp, 0) + link(i, j, 1) - link(ip, j, 1);
        tm = link(i, jm, 0) + link(ip, jm, 1) - link(i, jm, 1);
      } else {
        tp = link(i, j, 0) + link(ip, j, 1) - link(i, jp, 0);
        tm = link(im, jp, 0) + link(im, j, 1) - link(im, j, 0);
      }
      double &x = th[l];
      if (!heat) {
        x = mod_2pi_fast((tp + tm) - x);
      } else if (step) {
        x = vs_draw(key, l, 2. * coupling, tp, tm, tab);
      } else {
        double tau, centre;
        expcos_params(coupling, tp, tm, tau, centre);
        x = mod_2pi_fast(vonmises_draw(key, l, tau) + centre);
      }
    }
  } else {
    double *phi = state + (size_t)b * Mt * Mx;
    const double inv_kappa = 1. / (4. + coupling), two_over_kappa = 2. / (4. + coupling), sigma = 1. / sqrt(4. + coupling);
    for (uint32_t q = 0; q < n; ++q) {
      const uint32_t l = sites ? sites[q] : single;
      const uint32_t j = l / Mt, i = l - j * Mt;
      const uint32_t ip = i + 1 == Mt ? 0 : i + 1, im = i == 0 ? Mt - 1 : i - 1, jp = j + 1 == Mx ? 0 : j + 1, jm = j == 0 ? Mx - 1 : j - 1;
      double Delta = 0.0;  // the order of the reference's neighbour table: +i, -i, +j, -j
      Delta += phi[Mt * j + ip];
      Delta += phi[Mt * j + im];
      Delta += phi[Mt * jp + i];
      Delta += phi[Mt * jm + i];
      if (!heat) {
        phi[l] = fma(two_over_kappa, Delta, -phi[l]);
      } else {
        double n0, n1;
        rng_normals(key, l >> 1, P_GFF_NORMAL, 0, n0, n1);
        phi[l] = fma(Delta, inv_kappa, sigma * ((l & 1u) ? n1 : n0));
      }
    }
  }
}

// ---- host dispatch ----------------------------------------------------------------------------------------
static int check_lattice_dims(const mlmcpi_lattice_action *act) {
  if (!act) return fail(MLMCPI_ERR_INVALID, "action is NULL");
  if (act->kind != MLMCPI_GFF && act->kind != MLMCPI_SCHWINGER)
    return fail(MLMCPI_ERR_INVALID, "kind %d is not a 2-D lattice action", act->kind);
  if (act->Mt < 2 || act->Mx < 2) return fail(MLMCPI_ERR_INVALID, "lattice %u x %u too small", act->Mt, act->Mx);
  if ((uint64_t)act->Mt * act->Mx > (1ull << 30)) return fail(MLMCPI_ERR_INVALID, "lattice too large for 32-bit site indices");
  return MLMCPI_OK;
}

static int check_lattice(const mlmcpi_lattice_action *act) {
  if (!act) return fail(MLMCPI_ERR_INVALID, "action is NULL");
  if (act->kind != MLMCPI_GFF && act->kind != MLMCPI_SCHWINGER)
    return fail(MLMCPI_ERR_INVALID, "kind %d is not a 2-D lattice action", act->kind);
  if (act->Mt < 2 || act->Mx < 2) return fail(MLMCPI_ERR_INVALID, "lattice %u x %u too small", act->Mt, act->Mx);
  if ((uint64_t)act->Mt * act->Mx > (1ull << 30)) return fail(MLMCPI_ERR_INVALID, "lattice too large for 32-bit site indices");
  // gffaction.hh:169-173: the GFF action requires a square lattice
  if (act->kind == MLMCPI_GFF && act->Mt != act->Mx)
    return fail(MLMCPI_ERR_INVALID, "Lattice has to be squared for GFF action");
  return MLMCPI_OK;
}

static double gff_mu2(const mlmcpi_lattice_action &A) {  // gffaction.hh:174-181 (unrotated lattice)
  const double a_lat = 1. / A.Mt;
  return a_lat * a_lat * A.mass * A.mass;
}

static uint32_t row_blocks(uint32_t Mx, uint32_t B) {
  uint32_t want = (2048 + B - 1) / B;
  return want < Mx ? (want ? want : 1) : Mx;
}

template <int OP>
static int launch_lattice_reduce(uint32_t Mt, uint32_t Mx, double mu2, const double *d_state, uint32_t B, double scale,
                                 double *d_out, hipStream_t st) {
  uint32_t nsplit = row_blocks(Mx, B);
  constexpr bool plaquettes = OP == L_SCHW_ENERGY || OP == L_PLAQ || OP == L_CHARGE;
  // plaquette reductions on lattices up to 2048 columns: bands of consecutive rows, one load per site
  uint32_t rows_per_band = 0;
  if (plaquettes && Mt <= 2048 && Mt >= 64) {
    rows_per_band = (Mx + nsplit - 1) / nsplit;
    if (rows_per_band < 8) rows_per_band = Mx < 8 ? Mx : 8;  // the first row of a band is loaded twice: keep bands tall
    nsplit = (Mx + rows_per_band - 1) / rows_per_band;
  }
  void *ws = nullptr;
  if (int rc = scratch((size_t)B * nsplit * sizeof(double), &ws, st)) return rc;
  if constexpr (plaquettes) if (rows_per_band) {
    const double2 *t = (const double2 *)d_state;
    const int nc = (int)((Mt + 255) / 256);
#define MLMCPI_BAND(NC) hipLaunchKernelGGL((schwinger_reduce_band_kernel<OP, NC>), dim3(nsplit, B), dim3(256), 0, st, Mt, Mx, rows_per_band, t, (double *)ws)
    switch (nc) {
      case 1: MLMCPI_BAND(1); break;
      case 2: MLMCPI_BAND(2); break;
      case 3: MLMCPI_BAND(3); break;
      case 4: MLMCPI_BAND(4); break;
      case 5: MLMCPI_BAND(5); break;
      case 6: MLMCPI_BAND(6); break;
      case 7: MLMCPI_BAND(7); break;
      default: MLMCPI_BAND(8);
    }
#undef MLMCPI_BAND
    MLMCPI_LAUNCH_CHECK("schwinger_reduce_band_kernel");
  }
  if (!rows_per_band) {
    hipLaunchKernelGGL((lattice_reduce_kernel<OP>), dim3(nsplit, B), dim3(256), 0, st, Mt, Mx, mu2, d_state, (double *)ws);
    MLMCPI_LAUNCH_CHECK("lattice_reduce_kernel");
  }
  hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)ws, nsplit, B, OP,
                     scale, d_out);
  MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
  return MLMCPI_OK;
}

struct SweepGeom {
  TileGeom tg;
  uint32_t tiles_y, NT;
  size_t lds_bytes;
  bool overridden;  // MLMCPI_SWEEP_TILE given: use the generic kernels with that geometry
};

// Tile shape and workgroup size for a launch of `nsweeps` fused sweeps.  Default: 64 x 32 owned sites,
// 256 threads (4 workgroups per CU at one sweep).  MLMCPI_SWEEP_TILE=TWxTHxNT overrides (tuning knob;
// results never depend on it).
static SweepGeom choose_geometry(const Tuning &tune, uint32_t Mt, uint32_t Mx, uint32_t nsweeps, uint32_t bytes_per_cell) {
  uint32_t TW = 64, TH = 32, NT = 256;
  bool overridden = false;
  if (tune.tile_w) {
    TW = tune.tile_w; TH = tune.tile_h; NT = tune.tile_nt;
    overridden = true;
  }
  SweepGeom g;
  g.overridden = overridden;
  g.tg.TW = Mt < TW ? Mt : TW;
  g.tg.TH = Mx < TH ? Mx : TH;
  g.tg.tiles_x = (Mt + g.tg.TW - 1) / g.tg.TW;
  g.tiles_y = (Mx + g.tg.TH - 1) / g.tg.TH;
  g.NT = NT;
  g.lds_bytes = (size_t)(g.tg.TW + 4 * nsweeps) * (g.tg.TH + 4 * nsweeps) * bytes_per_cell;
  return g;
}

template <bool SCHW, bool HEAT, int NT>
static int launch_sweep_nt(const SweepGeom &g, dim3 grid, hipStream_t st, uint32_t Mt, uint32_t Mx, double coupling,
                            const double *src, double *dst, uint32_t n, uint32_t kinds, RngKey key, int qoi_op = 0,
                            double *qoi_partial = nullptr) {
  if (SCHW) {
    // heat-bath launches: room for the tables and the list of open cells in front of the tile image, as many entries as still keep the workgroup's
    // LDS footprint within a quarter of the CU's 160 KiB (4 workgroups per CU), at least one wave's worth
    const bool fixed = HEAT && NT == 256 && n == 1 && !g.overridden && g.tg.TW == 64 && g.tg.TH == 32 && Mt % 64 == 0 &&
                       Mx % 32 == 0 && Mt >= 128 && Mx >= 64;  // compile-time tile geometry (bit-identical results)
    const bool step = HEAT && 2. * coupling <= kVsKappaMax;      // which sampler: a property of the action, not a knob
    uint32_t cap = 0;
    size_t lds = g.lds_bytes;
    const uint32_t *vs_table = nullptr;
    if (step)
      if (int rc = vs_table_device(2. * coupling, &vs_table)) return rc;
    if (HEAT) {
      const size_t quarter = 40 * 1024 - 64;  // (the kernel's static LDS: the QoI reduction scratch)
      const size_t entry = step ? (fixed ? sizeof(uint16_t) : sizeof(uint32_t)) : 24, fixed_part = (step ? kVsTableBytes + 16 : 8) + 16;
      cap = 64;
      if (lds + fixed_part + entry * cap <= quarter) cap = (uint32_t)((quarter - lds - fixed_part) / entry) & ~7u;
      if (cap > (step ? 256u : 1024u)) cap = step ? 256u : 1024u;
      lds = g.lds_bytes + sweep_pool_bytes(step, fixed && step, cap);
      if (lds > 160 * 1024 - 256) { cap = 0; lds = g.lds_bytes + sweep_pool_bytes(step, fixed && step, 0); }
    }
    if (fixed && step)
      hipLaunchKernelGGL((schwinger_sweep_kernel<HEAT, NT, 64, 32, HEAT>), grid, dim3(NT), lds, st, Mt, Mx, coupling,
                         (const double2 *)src, (double2 *)dst, g.tg, n, kinds, key, cap, qoi_op, qoi_partial, vs_table);
    else if (fixed)
      hipLaunchKernelGGL((schwinger_sweep_kernel<HEAT, NT, 64, 32>), grid, dim3(NT), lds, st, Mt, Mx, coupling,
                         (const double2 *)src, (double2 *)dst, g.tg, n, kinds, key, cap, qoi_op, qoi_partial, vs_table);
    else if (step)
      hipLaunchKernelGGL((schwinger_sweep_kernel<HEAT, NT, 0, 0, HEAT>), grid, dim3(NT), lds, st, Mt, Mx, coupling,
                         (const double2 *)src, (double2 *)dst, g.tg, n, kinds, key, cap, qoi_op, qoi_partial, vs_table);
    else
      hipLaunchKernelGGL((schwinger_sweep_kernel<HEAT, NT>), grid, dim3(NT), lds, st, Mt, Mx, coupling,
                         (const double2 *)src, (double2 *)dst, g.tg, n, kinds, key, cap, qoi_op, qoi_partial, vs_table);
  }
  else
  if (HEAT && NT == 256 && n == 1 && (kinds & 1u) && !g.overridden && g.tg.TW == 64 && g.tg.TH == 32 && Mt % 64 == 0 &&
      Mx % 32 == 0 && Mt >= 128 && Mx >= 64)  // single heat-bath sweep: compile-time geometry (bit-identical results)
    hipLaunchKernelGGL((gff_sweep_kernel<HEAT, NT, 64, 32>), grid, dim3(NT), g.lds_bytes, st, Mt, Mx, coupling, src, dst, g.tg,
                       n, kinds, key, qoi_op, qoi_partial);
  else
    hipLaunchKernelGGL((gff_sweep_kernel<HEAT, NT>), grid, dim3(NT), g.lds_bytes, st, Mt, Mx, coupling, src, dst, g.tg, n,
                       kinds, key, qoi_op, qoi_partial);
  return MLMCPI_OK;
}

template <bool SCHW, bool HEAT>
static int launch_sweep(const SweepGeom &g, dim3 grid, hipStream_t st, uint32_t Mt, uint32_t Mx, double coupling,
                        const double *src, double *dst, uint32_t n, uint32_t kinds, RngKey key, int qoi_op = 0,
                        double *qoi_partial = nullptr) {
  int rc;
  switch (g.NT) {
    case 1024: rc = launch_sweep_nt<SCHW, HEAT, 1024>(g, grid, st, Mt, Mx, coupling, src, dst, n, kinds, key, qoi_op, qoi_partial); break;
    case 512: rc = launch_sweep_nt<SCHW, HEAT, 512>(g, grid, st, Mt, Mx, coupling, src, dst, n, kinds, key, qoi_op, qoi_partial); break;
    default: rc = launch_sweep_nt<SCHW, HEAT, 256>(g, grid, st, Mt, Mx, coupling, src, dst, n, kinds, key, qoi_op, qoi_partial);
  }
  if (rc) return rc;
  MLMCPI_LAUNCH_CHECK("lattice sweep kernel");
  return MLMCPI_OK;
}

template <bool HEAT, int NT>
static int allow_full_lds() {
  // tiles with deep halos may use the whole 160 KiB of LDS
  // (the kernels also hold NT / 64 doubles of static LDS for the fused QoI reduction)
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_sweep_kernel<HEAT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  if (HEAT)
    MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_sweep_kernel<HEAT, NT, 0, 0, HEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)gff_sweep_kernel<HEAT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  return MLMCPI_OK;
}

// hipFuncSetAttribute applies to the current device: once per device, under a lock (one process may drive several GPUs)
static std::mutex g_lds_attr_mutex;
static bool g_lds_attr_set[64] = {false};
static int init_sweep_kernels() {
  int dev = 0;
  MLMCPI_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(MLMCPI_ERR_INVALID, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_lds_attr_mutex);
  if (g_lds_attr_set[dev]) return MLMCPI_OK;
  if (int rc = allow_full_lds<false, 256>()) return rc;
  if (int rc = allow_full_lds<true, 256>()) return rc;
  if (int rc = allow_full_lds<false, 512>()) return rc;
  if (int rc = allow_full_lds<true, 512>()) return rc;
  if (int rc = allow_full_lds<false, 1024>()) return rc;
  if (int rc = allow_full_lds<true, 1024>()) return rc;
  // specialised overrelaxation kernels whose LDS image exceeds the 64 KiB default (K = 5, 6)
#define MLMCPI_OR_ATTR(KK, NN) MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_kernel<64, 32, KK, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  MLMCPI_OR_ATTR(5, 256); MLMCPI_OR_ATTR(5, 512); MLMCPI_OR_ATTR(5, 1024);
  MLMCPI_OR_ATTR(6, 256); MLMCPI_OR_ATTR(6, 512); MLMCPI_OR_ATTR(6, 1024);
#undef MLMCPI_OR_ATTR
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_block_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrBlockGeom<5>::lds_bytes));
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_block_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrBlockGeom<6>::lds_bytes));
#define MLMCPI_OR_HEAT_ATTR(KK) \
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_heat_kernel<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrHeatGeom<KK>::lds_bytes)); \
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_heat_kernel<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrHeatGeom<KK>::lds_bytes)); \
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_heat_kernel<KK, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrHeatGeom<KK>::lds_bytes)); \
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)gff_or_heat_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GffHeatGeom<KK>::lds_bytes))
  MLMCPI_OR_HEAT_ATTR(1); MLMCPI_OR_HEAT_ATTR(2); MLMCPI_OR_HEAT_ATTR(3); MLMCPI_OR_HEAT_ATTR(4); MLMCPI_OR_HEAT_ATTR(5);
#undef MLMCPI_OR_HEAT_ATTR
#define MLMCPI_OR_HEAT_ATTR_W(KK) MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_or_heat_kernel<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OrHeatGeom<KK>::lds_bytes))
  MLMCPI_OR_HEAT_ATTR_W(6); MLMCPI_OR_HEAT_ATTR_W(7); MLMCPI_OR_HEAT_ATTR_W(8); MLMCPI_OR_HEAT_ATTR_W(9); MLMCPI_OR_HEAT_ATTR_W(10);
#undef MLMCPI_OR_HEAT_ATTR_W
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_perm_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPermPlaneMax));
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_perm_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPermPlaneMax));
#define MLMCPI_PERM_HEAT_ATTR(NN, SS) MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)schwinger_perm_heat_kernel<NN, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, NN == 1024 ? 156 * 1024 : (int)OrHeatGeom<1>::hb_bytes))
  MLMCPI_PERM_HEAT_ATTR(512, true); MLMCPI_PERM_HEAT_ATTR(512, false); MLMCPI_PERM_HEAT_ATTR(1024, true); MLMCPI_PERM_HEAT_ATTR(1024, false);
#undef MLMCPI_PERM_HEAT_ATTR
  g_lds_attr_set[dev] = true;
  return MLMCPI_OK;
}

}  // namespace mlmcpi

using namespace mlmcpi;

extern "C" {

int mlmcpi_lattice_state_size(const mlmcpi_lattice_action *act, uint32_t *n) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(n, "n is NULL");
  *n = (act->kind == MLMCPI_SCHWINGER ? 2u : 1u) * act->Mt * act->Mx;
  return MLMCPI_OK;
}

int mlmcpi_lattice_evaluate(const mlmcpi_lattice_action *act, const double *d_phi, uint32_t B, double *d_S,
                            void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_phi && d_S && B > 0, "bad arguments");
  if (act->kind == MLMCPI_GFF)
    return launch_lattice_reduce<L_GFF_ENERGY>(act->Mt, act->Mx, gff_mu2(*act), d_phi, B, 0.5, d_S, as_stream(stream));
  return launch_lattice_reduce<L_SCHW_ENERGY>(act->Mt, act->Mx, 0.0, d_phi, B, act->beta, d_S, as_stream(stream));
}

int mlmcpi_lattice_force(const mlmcpi_lattice_action *act, const double *d_phi, double *d_f, uint32_t B,
                         void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_phi && d_f && d_phi != d_f && B > 0, "bad arguments");
  dim3 grid(row_blocks(act->Mx, B), B), block(256);
  if (act->kind == MLMCPI_GFF)
    hipLaunchKernelGGL(gff_force_kernel, grid, block, 0, as_stream(stream), act->Mt, act->Mx, gff_mu2(*act), d_phi, d_f);
  else
    hipLaunchKernelGGL(schwinger_force_kernel, dim3((force_waves(act->Mt, act->Mx) + 3) / 4, B), block, 0, as_stream(stream),
                       act->Mt, act->Mx, act->beta, (const double2 *)d_phi, (double2 *)d_f);
  MLMCPI_LAUNCH_CHECK("lattice force kernel");
  return MLMCPI_OK;
}

static int gff_initialise_exact(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0,
                                hipStream_t st);

int mlmcpi_lattice_initialise(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed,
                              uint32_t chain0, void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_phi && B > 0, "bad arguments");
  if (act->kind == MLMCPI_GFF) return gff_initialise_exact(act, d_phi, B, seed, chain0, as_stream(stream));
  uint32_t n = 0;
  mlmcpi_lattice_state_size(act, &n);
  uint32_t nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(lattice_init_kernel, dim3(nb, B), dim3(256), 0, as_stream(stream), act->kind, n,
                     make_key(seed, chain0, 0), d_phi);
  MLMCPI_LAUNCH_CHECK("lattice_init_kernel");
  return MLMCPI_OK;
}

// The launches read `src` and write `dst`; after each one src <- dst and dst <- the other work buffer.  d_phi is only
// read unless it is also d_w1.  result_in (may be NULL: then the result is copied into d_phi, which must be writable):
// 0 -> the result is in d_w0, 1 -> in d_w1, -1 -> no sweep was run (result is the input).
// qoi_kind != 0 (1 average plaquette, 2 Q^2 / 4 pi^2): the QoI of the final state, summed inside the last launch (which
// has to be a launch of schwinger_sweep_kernel, i.e. the draw must end with a heat-bath sweep), into d_qoi[b].
static int sweep_draw_impl(const mlmcpi_lattice_action *act, double *d_phi, double *d_w0, double *d_w1, uint32_t B,
                           uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                           uint32_t fuse, int32_t *result_in, void *stream, int qoi_kind = 0, double *d_qoi = nullptr,
                           double *d_acc = nullptr) {
  if (int rc = check_lattice(act)) return rc;
  if (qoi_kind) {
    MLMCPI_REQUIRE(d_qoi && qoi_kind >= 1 && qoi_kind <= 3, "bad QoI arguments");
    if ((qoi_kind == 3) != (act->kind == MLMCPI_GFF))
      return fail(MLMCPI_ERR_UNSUPPORTED, "fused QoI %d does not belong to this action", qoi_kind);
    if (n_heatbath == 0) return fail(MLMCPI_ERR_UNSUPPORTED, "the fused QoI needs a draw that ends with a heat-bath sweep");
  }
  MLMCPI_REQUIRE(d_phi && d_w0 && d_w1 && d_phi != d_w0 && d_w0 != d_w1 && B > 0, "bad arguments");
  MLMCPI_REQUIRE(act->Mt % 2 == 0 && act->Mx % 2 == 0, "multicolour sweeps need even Mt, Mx (got %u x %u)", act->Mt,
                 act->Mx);
  // library default: best measured whole-step time -- up to 6 sweeps per launch where the 4 x 4
  // register-block kernel applies, 4 otherwise
  const Tuning tune = tuning();  // ONE snapshot per draw: mlmcpi_set_option on another thread cannot split a launch plan
  const bool or_blocks = !tune.or_lds && !tune.or_patch && !tune.tile_w && act->Mt % 64 == 0 && act->Mx % 64 == 0;
  const bool schw = act->kind == MLMCPI_SCHWINGER;
  // GFF lattices that 32 x 32 tiles divide and 64 x 64 ones do not (or that are below 128, where the 64-tile fused launch
  // does not apply): the register-block kernels on 32 x 32 tiles, same launch plan
  // (r05: also lattices no tile divides -- edge tiles computed whole and written in part -- unless the padding would more
  // than double the work)
  const uint32_t g32_tx = (act->Mt + 31) / 32, g32_ty = (act->Mx + 31) / 32;
  const bool gff_blocks32 = !schw && !tune.or_lds && !tune.or_patch && !tune.tile_w && act->Mt >= 64 && act->Mx >= 64 &&
                            !(or_blocks && act->Mt >= 128 && act->Mx >= 128) &&
                            (uint64_t)g32_tx * g32_ty * 1024 <= (uint64_t)2 * act->Mt * act->Mx + (uint64_t)act->Mt * act->Mx / 5;
  // One chain (at most one workgroup of the fused launch per CU: nothing to overlap a second launch's load and store
  // phases with): the whole draw in ONE launch of schwinger_or_heat_kernel<n_overrelax, wide> while its halo fits a
  // workgroup (n_overrelax <= 10) -- the library default only; a caller's `fuse` is kept.
  const bool whole_draw = fuse == 0 && schw && or_blocks && !tune.or_heat_split && tune.or_heat_wide >= 0 && n_heatbath >= 1 &&
                          n_overrelax >= 6 && n_overrelax <= 10 && 2. * act->beta <= kVsKappaMax && act->Mt >= 128 && act->Mx >= 128 &&
                          (uint64_t)(act->Mt / 64) * (act->Mx / 64) * B <= kComputeUnits;
  // Schwinger overrelaxation in closed form (schwinger_perm_kernel, schwinger_perm_heat_kernel): the default where 64 x 64
  // tiles divide the lattice; MLMCPI_OR_KERNEL=block|patch|lds select the sweep-by-sweep kernels
  // r05: any even lattice of at least one tile.  Tiles on the upper / right edge of a lattice the tiles do not divide reach
  // beyond it; what lies there are periodic images of vertices other tiles own (the plane wraps as often as needed):
  // computed like any halo, not written.  64 x 64 tiles where they divide Mx and wherever the fused launch applies (both
  // extents >= 128: its image must not wrap onto itself), else 64 x 32 (the heat bath is then a launch of its own);
  // lattices that would more than double the work through padding stay with the sweep-by-sweep kernels.
  const bool perm_shape = schw && act->Mt >= 64 && act->Mx >= 32 && (uint64_t)act->Mt * act->Mx < (1ull << 28);   // (32-bit byte offsets)
  const bool perm64 = perm_shape && (act->Mx % 64 == 0 || (act->Mt >= 128 && act->Mx >= 128));
  const uint32_t perm_th = perm64 ? 64 : 32, perm_tx = (act->Mt + 63) / 64, perm_ty = (act->Mx + perm_th - 1) / perm_th;
  const bool perm = perm_shape && !tune.or_block && !tune.or_lds && !tune.or_patch && !tune.tile_w &&
                    (uint64_t)perm_tx * 64 * perm_ty * perm_th <= (uint64_t)2 * act->Mt * act->Mx + (uint64_t)act->Mt * act->Mx / 5;
  const uint32_t fuse_arg = fuse;
  if (fuse == 0) fuse = whole_draw ? n_overrelax : (or_blocks || gff_blocks32) ? 6 : 4;
  if (fuse > kMaxFuse) fuse = kMaxFuse;
  hipStream_t st = as_stream(stream);
  const uint32_t total = n_overrelax + n_heatbath;
  const size_t state_bytes = (size_t)B * act->Mt * act->Mx * (schw ? 16 : 8);
  if (int rc = init_sweep_kernels()) return rc;
  double *src = d_phi, *dst = d_w0;
  auto advance = [&]() {  // the buffer just written becomes the input; the next output is the other work buffer
    src = dst;
    dst = (dst == d_w0) ? d_w1 : d_w0;
  };
  uint32_t s = 0;
  while (s < total) {
    // Overrelaxation sweeps are fused: they are bound by the passes over the state, and a fused launch trades halo
    // recomputation (cheap for them) for passes.  A heat-bath sweep is bound by its sampler arithmetic, which a wider halo
    // would only multiply: no sweep follows it inside a launch.  As the LAST sweep of an overrelaxation launch it needs no
    // halo of its own beyond the two rings it reads (the *_or_heat_kernel branches below); otherwise it gets a launch to itself.
    uint32_t n = 1;
    if (s < n_overrelax) {
      const uint32_t rem = n_overrelax - s;
      n = rem < fuse ? rem : fuse;
      if (or_blocks || gff_blocks32) {  // as few launches as `fuse` allows, of equal depth (10 sweeps, fuse 6: 5 + 5, not 6 + 4)
        const uint32_t launches = (rem + fuse - 1) / fuse;
        n = (rem + launches - 1) / launches;
      }
    }
    if (perm && s < n_overrelax) {
      // as few launches as kPermMaxK (or the caller's `fuse`) allows, of equal depth
      const uint32_t rem = n_overrelax - s, kmax = fuse_arg ? std::min(fuse_arg, kPermMaxK) : kPermMaxK;
      const uint32_t launches = (rem + kmax - 1) / kmax, K = (rem + launches - 1) / launches;
      const dim3 bgrid(perm_tx * perm_ty, B);
      const double2 *in2 = (const double2 *)src;
      double2 *out2 = (double2 *)dst;
      if (perm64 && !tune.or_heat_split && s + K == n_overrelax && n_heatbath >= 1 && act->Mt >= 128 && act->Mx >= 128) {
        // the last overrelaxation launch takes the heat-bath sweep behind it along, and the QoI if that ends the draw
        const bool step = 2. * act->beta <= kVsKappaMax;   // which sampler: a property of the action (device_common.hpp)
        const uint32_t *vs_table = nullptr;
        if (step)
          if (int rcv = vs_table_device(2. * act->beta, &vs_table)) return rcv;
        const bool with_qoi = qoi_kind && s + K + 1 == total;
        void *partial = nullptr;
        if (with_qoi)
          if (int rcs = scratch((size_t)B * bgrid.x * sizeof(double), &partial, st)) return rcs;
        const int op = !with_qoi ? 0 : qoi_kind == 1 ? (int)L_PLAQ : (int)L_CHARGE;
        const RngKey hkey = make_key(seed, chain0, sweep0 + s + K);
        // at most one workgroup per CU: sixteen waves (MLMCPI_OR_HEAT=wide|narrow forces)
        const bool wide = tune.or_heat_wide ? tune.or_heat_wide > 0 : (uint64_t)bgrid.x * B <= kComputeUnits;
        using PHG = PermHeatGeom<512, true>;
        // one plane for all 68 rows where it fits: beside a second workgroup (narrow) or in the whole LDS (wide)
        const size_t lds_max = wide ? (size_t)156 * 1024 : OrHeatGeom<1>::hb_bytes;
        const uint32_t NB = PHG::lds_bytes(K, 1) <= lds_max ? 1 : 2;
        const size_t lds = PHG::lds_bytes(K, NB);
        if (lds > lds_max) return fail(MLMCPI_ERR_INVALID, "closed-form plane of %u sweeps does not fit", K);
#define MLMCPI_PERM_HEAT(NN, SS) hipLaunchKernelGGL((schwinger_perm_heat_kernel<NN, SS>), bgrid, dim3(NN), lds, st, act->Mt, act->Mx, act->beta, in2, out2, perm_tx, K, NB, hkey, op, (double *)partial, vs_table)
        if (wide) { if (step) MLMCPI_PERM_HEAT(1024, true); else MLMCPI_PERM_HEAT(1024, false); }
        else { if (step) MLMCPI_PERM_HEAT(512, true); else MLMCPI_PERM_HEAT(512, false); }
#undef MLMCPI_PERM_HEAT
        MLMCPI_LAUNCH_CHECK("schwinger_perm_heat_kernel");
        if (with_qoi) {
          hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)partial, bgrid.x, B, op,
                             1.0 / ((double)act->Mx * act->Mt), d_qoi, d_acc);
          MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
        }
        advance();
        s += K + 1;
        continue;
      }
      if (perm64) {
        using PG0 = PermGeom<512, 0, 64>;
        const uint32_t NB = PG0::plane_bytes(K, 1) <= kPermPlaneMax ? 1 : 2;
        hipLaunchKernelGGL(schwinger_perm_kernel<64>, bgrid, dim3(512), perm_lds_bytes<64>(K, NB), st, act->Mt, act->Mx, in2, out2, perm_tx, K, NB);
      } else {
        using PG0 = PermGeom<512, 0, 32>;
        const uint32_t NB = PG0::plane_bytes(K, 1) <= kPermPlaneMax ? 1 : 2;
        hipLaunchKernelGGL(schwinger_perm_kernel<32>, bgrid, dim3(512), perm_lds_bytes<32>(K, NB), st, act->Mt, act->Mx, in2, out2, perm_tx, K, NB);
      }
      MLMCPI_LAUNCH_CHECK("schwinger_perm_kernel");
      advance();
      s += K;
      continue;
    }
    SweepGeom g;
    uint32_t kinds = 0;
    for (;;) {  // shrink the fused count until the tile + halo fits in LDS
      kinds = 0;
      for (uint32_t q = 0; q < n; ++q)
        if (s + q >= n_overrelax) kinds |= 1u << q;
      // Schwinger: two link angles per site; GFF heat bath: field + parked normal per site
      g = choose_geometry(tune, act->Mt, act->Mx, n, (schw || kinds) ? 16 : 8);
      if (g.lds_bytes <= 160 * 1024 - 256 || n == 1) break;
      --n;
    }
    const RngKey key = make_key(seed, chain0, sweep0 + s);
    dim3 grid(g.tg.tiles_x * g.tiles_y, B);
    int rc;
    if (schw && !kinds && !g.overridden && act->Mt % 64 == 0 && act->Mx % 32 == 0 && (n <= 6 || whole_draw)) {
      // specialised overrelaxation kernel (bit-identical to the generic one)
      const size_t lds = (size_t)2 * (32 + 4 * n) * (64 + 4 * n + 1) * sizeof(double);
      dim3 sgrid((act->Mt / 64) * (act->Mx / 32), B);
      const double2 *in2 = (const double2 *)src;
      double2 *out2 = (double2 *)dst;
      // more fused sweeps -> larger LDS image -> fewer resident workgroups: keep the wave count per CU up
      // with wider workgroups (MLMCPI_OR_THREADS overrides: tuning knob)
      const bool use_patch = !tune.or_lds;
      if (or_blocks) {  // 4 x 4 register blocks on 64 x 64 tiles (n <= 6)
        dim3 bgrid((act->Mt / 64) * (act->Mx / 64), B);
        // the last overrelaxation launch of the draw takes the heat-bath sweep behind it along (and the QoI, if that is
        // the draw's last sweep): schwinger_or_heat_kernel, bit-identical to the two launches (MLMCPI_OR_HEAT=split)
        if (!tune.or_heat_split && s + n == n_overrelax && n_heatbath >= 1 && (n <= 5 || whole_draw) && act->Mt >= 128 && act->Mx >= 128) {
          const bool step = 2. * act->beta <= kVsKappaMax;   // which sampler: a property of the action (device_common.hpp)
          const uint32_t *vs_table = nullptr;
          if (step)
            if (int rcv = vs_table_device(2. * act->beta, &vs_table)) return rcv;
          const bool with_qoi = qoi_kind && s + n + 1 == total;
          void *partial = nullptr;
          if (with_qoi)
            if (int rcs = scratch((size_t)B * bgrid.x * sizeof(double), &partial, st)) return rcs;
          const int op = !with_qoi ? 0 : qoi_kind == 1 ? (int)L_PLAQ : (int)L_CHARGE;
          const RngKey hkey = make_key(seed, chain0, sweep0 + s + n);
          // at most one workgroup per CU: sixteen waves for the heat-bath part (bit-identical; MLMCPI_OR_HEAT=wide|narrow forces)
          const bool wide = tune.or_heat_wide ? tune.or_heat_wide > 0 : (uint64_t)bgrid.x * B <= kComputeUnits;
#define MLMCPI_OR_HEAT_W(KK, WW) hipLaunchKernelGGL((schwinger_or_heat_kernel<KK, WW>), bgrid, dim3(WW ? 1024 : OrHeatGeom<KK>::NT), OrHeatGeom<KK>::lds_bytes, st, act->Mt, act->Mx, act->beta, in2, out2, act->Mt / 64, hkey, op, (double *)partial, vs_table)
#define MLMCPI_OR_HEAT_CAUCHY(KK) hipLaunchKernelGGL((schwinger_or_heat_kernel<KK, false, false>), bgrid, dim3(OrHeatGeom<KK>::NT), OrHeatGeom<KK>::lds_bytes, st, act->Mt, act->Mx, act->beta, in2, out2, act->Mt / 64, hkey, op, (double *)partial, vs_table)
#define MLMCPI_OR_HEAT(KK) do { if (!step) MLMCPI_OR_HEAT_CAUCHY(KK); else if (wide) MLMCPI_OR_HEAT_W(KK, true); else MLMCPI_OR_HEAT_W(KK, false); } while (0)
          switch (n) {
            case 1: MLMCPI_OR_HEAT(1); break;
            case 2: MLMCPI_OR_HEAT(2); break;
            case 3: MLMCPI_OR_HEAT(3); break;
            case 4: MLMCPI_OR_HEAT(4); break;
            case 5: MLMCPI_OR_HEAT(5); break;
            case 6: MLMCPI_OR_HEAT_W(6, true); break;  // whole_draw (wide workgroups only)
            case 7: MLMCPI_OR_HEAT_W(7, true); break;
            case 8: MLMCPI_OR_HEAT_W(8, true); break;
            case 9: MLMCPI_OR_HEAT_W(9, true); break;
            default: MLMCPI_OR_HEAT_W(10, true);
          }
#undef MLMCPI_OR_HEAT
#undef MLMCPI_OR_HEAT_CAUCHY
#undef MLMCPI_OR_HEAT_W
          MLMCPI_LAUNCH_CHECK("schwinger_or_heat_kernel");
          if (with_qoi) {
            hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)partial, bgrid.x, B, op,
                               1.0 / ((double)act->Mx * act->Mt), d_qoi, d_acc);
            MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
          }
          advance();
          s += n + 1;
          continue;
        }
#define MLMCPI_OR_BLOCK(KK) hipLaunchKernelGGL((schwinger_or_block_kernel<KK>), bgrid, dim3(OrBlockGeom<KK>::NT), OrBlockGeom<KK>::lds_bytes, st, act->Mt, act->Mx, in2, out2, act->Mt / 64)
        switch (n) {
          case 1: MLMCPI_OR_BLOCK(1); break;
          case 2: MLMCPI_OR_BLOCK(2); break;
          case 3: MLMCPI_OR_BLOCK(3); break;
          case 4: MLMCPI_OR_BLOCK(4); break;
          case 5: MLMCPI_OR_BLOCK(5); break;
          default: MLMCPI_OR_BLOCK(6);
        }
#undef MLMCPI_OR_BLOCK
        MLMCPI_LAUNCH_CHECK("schwinger_or_block_kernel");
        advance();
        s += n;
        continue;
      }
      if (use_patch && n <= 4) {  // register-tiled kernel (MLMCPI_OR_KERNEL=lds selects the LDS-resident one)
        const uint32_t np = ((64 + 4 * n) / 2) * ((32 + 4 * n) / 2);
        const dim3 pblock((np + 63) / 64 * 64);
        const size_t plds = (size_t)8 * np * sizeof(double);
        switch (n) {
          case 1: hipLaunchKernelGGL((schwinger_or_patch_kernel<1>), sgrid, pblock, plds, st, act->Mt, act->Mx, in2, out2, act->Mt / 64); break;
          case 2: hipLaunchKernelGGL((schwinger_or_patch_kernel<2>), sgrid, pblock, plds, st, act->Mt, act->Mx, in2, out2, act->Mt / 64); break;
          case 3: hipLaunchKernelGGL((schwinger_or_patch_kernel<3>), sgrid, pblock, plds, st, act->Mt, act->Mx, in2, out2, act->Mt / 64); break;
          default: hipLaunchKernelGGL((schwinger_or_patch_kernel<4>), sgrid, pblock, plds, st, act->Mt, act->Mx, in2, out2, act->Mt / 64);
        }
        MLMCPI_LAUNCH_CHECK("schwinger_or_patch_kernel");
        advance();
        s += n;
        continue;
      }
      uint32_t nt_or = n >= 4 ? 1024 : 512;  // measured best (tools/scan_or.sh): K <= 3: 512, K >= 4: 1024
      if (tune.or_threads) nt_or = tune.or_threads;
#define MLMCPI_OR(KK, NN) hipLaunchKernelGGL((schwinger_or_kernel<64, 32, KK, NN>), sgrid, dim3(NN), lds, st, act->Mt, act->Mx, in2, out2, act->Mt / 64)
#define MLMCPI_OR_K(KK) do { if (nt_or == 1024) MLMCPI_OR(KK, 1024); else if (nt_or == 512) MLMCPI_OR(KK, 512); else MLMCPI_OR(KK, 256); } while (0)
      switch (n) {
        case 1: MLMCPI_OR_K(1); break;
        case 2: MLMCPI_OR_K(2); break;
        case 3: MLMCPI_OR_K(3); break;
        case 4: MLMCPI_OR_K(4); break;
        case 5: MLMCPI_OR_K(5); break;
        default: MLMCPI_OR_K(6);
      }
#undef MLMCPI_OR_K
#undef MLMCPI_OR
      MLMCPI_LAUNCH_CHECK("schwinger_or_kernel");
      rc = MLMCPI_OK;
    } else if (!schw && !kinds && !g.overridden && gff_blocks32 && n <= 6) {
      const double mu2 = gff_mu2(*act);
      dim3 bgrid(g32_tx * g32_ty, B);
      if (!tune.or_heat_split && s + n == n_overrelax && n_heatbath >= 1 && n <= 5) {
        const bool with_qoi = qoi_kind && s + n + 1 == total;
        void *partial = nullptr;
        if (with_qoi)
          if (int rcs = scratch((size_t)B * bgrid.x * sizeof(double), &partial, st)) return rcs;
        const int op = with_qoi ? (int)L_PHI2 : 0;
        const RngKey hkey = make_key(seed, chain0, sweep0 + s + n);
#define MLMCPI_GFF_HEAT32(KK) hipLaunchKernelGGL((gff_or_heat_kernel<KK, 32>), bgrid, dim3((GffHeatGeom<KK, 32>::NT)), (GffHeatGeom<KK, 32>::lds_bytes), st, act->Mt, act->Mx, mu2, (const double *)src, dst, g32_tx, hkey, op, (double *)partial)
        switch (n) {
          case 1: MLMCPI_GFF_HEAT32(1); break;
          case 2: MLMCPI_GFF_HEAT32(2); break;
          case 3: MLMCPI_GFF_HEAT32(3); break;
          case 4: MLMCPI_GFF_HEAT32(4); break;
          default: MLMCPI_GFF_HEAT32(5);
        }
#undef MLMCPI_GFF_HEAT32
        MLMCPI_LAUNCH_CHECK("gff_or_heat_kernel<., 32>");
        if (with_qoi) {
          hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)partial, bgrid.x, B, op,
                             1.0 / ((double)act->Mx * act->Mt), d_qoi, d_acc);
          MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
        }
        advance();
        s += n + 1;
        continue;
      }
#define MLMCPI_GFF_BLOCK32(KK) hipLaunchKernelGGL((gff_or_block_kernel<KK, 32>), bgrid, dim3((GffBlockGeom<KK, 32>::NT)), (GffBlockGeom<KK, 32>::lds_bytes), st, act->Mt, act->Mx, mu2, (const double *)src, dst, g32_tx)
      switch (n) {
        case 1: MLMCPI_GFF_BLOCK32(1); break;
        case 2: MLMCPI_GFF_BLOCK32(2); break;
        case 3: MLMCPI_GFF_BLOCK32(3); break;
        case 4: MLMCPI_GFF_BLOCK32(4); break;
        case 5: MLMCPI_GFF_BLOCK32(5); break;
        default: MLMCPI_GFF_BLOCK32(6);
      }
#undef MLMCPI_GFF_BLOCK32
      MLMCPI_LAUNCH_CHECK("gff_or_block_kernel<., 32>");
      advance();
      s += n;
      continue;
    } else if (!schw && !kinds && !g.overridden && act->Mt % 64 == 0 && act->Mx % 32 == 0 && n <= (or_blocks ? 6u : 4u)) {
      const size_t lds = (size_t)(32 + 4 * n) * (64 + 4 * n + 1) * sizeof(double);
      dim3 sgrid((act->Mt / 64) * (act->Mx / 32), B);
      const double mu2 = gff_mu2(*act);
      const bool use_gff_patch = !tune.or_lds;
      if (or_blocks) {  // 4 x 4 register blocks on 64 x 64 tiles (n <= 6)
        dim3 bgrid((act->Mt / 64) * (act->Mx / 64), B);
        // as for the Schwinger action: the last overrelaxation launch takes the heat-bath sweep (and the QoI) along
        if (!tune.or_heat_split && s + n == n_overrelax && n_heatbath >= 1 && n <= 5 && act->Mt >= 128 && act->Mx >= 128) {
          const bool with_qoi = qoi_kind && s + n + 1 == total;
          void *partial = nullptr;
          if (with_qoi)
            if (int rcs = scratch((size_t)B * bgrid.x * sizeof(double), &partial, st)) return rcs;
          const int op = with_qoi ? (int)L_PHI2 : 0;
          const RngKey hkey = make_key(seed, chain0, sweep0 + s + n);
#define MLMCPI_GFF_HEAT(KK) hipLaunchKernelGGL((gff_or_heat_kernel<KK>), bgrid, dim3(GffHeatGeom<KK>::NT), GffHeatGeom<KK>::lds_bytes, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64, hkey, op, (double *)partial)
          switch (n) {
            case 1: MLMCPI_GFF_HEAT(1); break;
            case 2: MLMCPI_GFF_HEAT(2); break;
            case 3: MLMCPI_GFF_HEAT(3); break;
            case 4: MLMCPI_GFF_HEAT(4); break;
            default: MLMCPI_GFF_HEAT(5);
          }
#undef MLMCPI_GFF_HEAT
          MLMCPI_LAUNCH_CHECK("gff_or_heat_kernel");
          if (with_qoi) {
            hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)partial, bgrid.x, B, op,
                               1.0 / ((double)act->Mx * act->Mt), d_qoi, d_acc);
            MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
          }
          advance();
          s += n + 1;
          continue;
        }
#define MLMCPI_GFF_BLOCK(KK) hipLaunchKernelGGL((gff_or_block_kernel<KK>), bgrid, dim3(GffBlockGeom<KK>::NT), GffBlockGeom<KK>::lds_bytes, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64)
        switch (n) {
          case 1: MLMCPI_GFF_BLOCK(1); break;
          case 2: MLMCPI_GFF_BLOCK(2); break;
          case 3: MLMCPI_GFF_BLOCK(3); break;
          case 4: MLMCPI_GFF_BLOCK(4); break;
          case 5: MLMCPI_GFF_BLOCK(5); break;
          default: MLMCPI_GFF_BLOCK(6);
        }
#undef MLMCPI_GFF_BLOCK
        MLMCPI_LAUNCH_CHECK("gff_or_block_kernel");
        advance();
        s += n;
        continue;
      }
      if (use_gff_patch) {  // register-tiled kernel (MLMCPI_OR_KERNEL=lds selects the LDS-resident one)
        const uint32_t np = ((64 + 4 * n) / 2) * ((32 + 4 * n) / 2);
        const dim3 pblock((np + 63) / 64 * 64);
        const size_t plds = (size_t)4 * np * sizeof(double);
        switch (n) {
          case 1: hipLaunchKernelGGL((gff_or_patch_kernel<1>), sgrid, pblock, plds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
          case 2: hipLaunchKernelGGL((gff_or_patch_kernel<2>), sgrid, pblock, plds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
          case 3: hipLaunchKernelGGL((gff_or_patch_kernel<3>), sgrid, pblock, plds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
          default: hipLaunchKernelGGL((gff_or_patch_kernel<4>), sgrid, pblock, plds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64);
        }
        MLMCPI_LAUNCH_CHECK("gff_or_patch_kernel");
        advance();
        s += n;
        continue;
      }
      switch (n) {
        case 1: hipLaunchKernelGGL((gff_or_kernel<64, 32, 1, 256>), sgrid, dim3(256), lds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
        case 2: hipLaunchKernelGGL((gff_or_kernel<64, 32, 2, 256>), sgrid, dim3(256), lds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
        case 3: hipLaunchKernelGGL((gff_or_kernel<64, 32, 3, 256>), sgrid, dim3(256), lds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64); break;
        default: hipLaunchKernelGGL((gff_or_kernel<64, 32, 4, 256>), sgrid, dim3(256), lds, st, act->Mt, act->Mx, mu2, (const double *)src, dst, act->Mt / 64);
      }
      MLMCPI_LAUNCH_CHECK("gff_or_kernel");
      rc = MLMCPI_OK;
    } else
    // launches without a heat-bath sweep use the lean instantiation (no sampler code, fewer VGPRs)
    if (kinds && qoi_kind && s + n == total) {  // the last launch of the draw: sum the QoI while the tile is in LDS
      void *partial = nullptr;
      if (int rcs = scratch((size_t)B * grid.x * sizeof(double), &partial, st)) return rcs;
      const int op = qoi_kind == 1 ? (int)L_PLAQ : qoi_kind == 2 ? (int)L_CHARGE : (int)L_PHI2;
      rc = schw ? launch_sweep<true, true>(g, grid, st, act->Mt, act->Mx, act->beta, src, dst, n, kinds, key, op, (double *)partial)
                : launch_sweep<false, true>(g, grid, st, act->Mt, act->Mx, gff_mu2(*act), src, dst, n, kinds, key, op, (double *)partial);
      if (rc) return rc;
      hipLaunchKernelGGL(lattice_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const double *)partial, grid.x, B, op,
                         1.0 / ((double)act->Mx * act->Mt), d_qoi, d_acc);
      MLMCPI_LAUNCH_CHECK("lattice_finish_kernel");
    } else if (schw)
      rc = kinds ? launch_sweep<true, true>(g, grid, st, act->Mt, act->Mx, act->beta, src, dst, n, kinds, key)
                 : launch_sweep<true, false>(g, grid, st, act->Mt, act->Mx, act->beta, src, dst, n, kinds, key);
    else
      rc = kinds ? launch_sweep<false, true>(g, grid, st, act->Mt, act->Mx, gff_mu2(*act), src, dst, n, kinds, key)
                 : launch_sweep<false, false>(g, grid, st, act->Mt, act->Mx, gff_mu2(*act), src, dst, n, kinds, key);
    if (rc) return rc;
    advance();
    s += n;
  }
  if (result_in)
    *result_in = total == 0 ? -1 : (src == d_w0 ? 0 : 1);
  else if (src != d_phi)
    MLMCPI_HIP_TRY(hipMemcpyAsync(d_phi, src, state_bytes, hipMemcpyDeviceToDevice, st));
  return MLMCPI_OK;
}

int mlmcpi_lattice_sweep_draw(const mlmcpi_lattice_action *act, double *d_phi, double *d_scratch, uint32_t B,
                              uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                              uint32_t sweep0, uint32_t fuse, void *stream) {
  return sweep_draw_impl(act, d_phi, d_scratch, d_phi, B, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse, nullptr, stream);
}

int mlmcpi_lattice_sweep_draw_pingpong(const mlmcpi_lattice_action *act, double *d_a, double *d_b, uint32_t B,
                                       uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                       uint32_t sweep0, uint32_t fuse, int32_t *result_in_b, void *stream) {
  MLMCPI_REQUIRE(result_in_b, "result_in_b is NULL");
  int32_t where = 0;
  if (int rc = sweep_draw_impl(act, d_a, d_b, d_a, B, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse, &where, stream)) return rc;
  *result_in_b = where == 0 ? 1 : 0;  // work buffer 0 is d_b; no sweeps (-1) or work buffer 1: the result is in d_a
  return MLMCPI_OK;
}

int mlmcpi_lattice_sweep_draw_from(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1,
                                   uint32_t B, uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                   uint32_t sweep0, uint32_t fuse, int32_t *result_in, void *stream) {
  MLMCPI_REQUIRE(result_in, "result_in is NULL");
  MLMCPI_REQUIRE(n_overrelax + n_heatbath > 0, "no sweeps requested: the result would be the (read-only) input");
  // the input is only ever the `src` of the first launch (or a work buffer when the caller passes d_w1 == d_src)
  return sweep_draw_impl(act, const_cast<double *>(d_src), d_w0, d_w1, B, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse,
                         result_in, stream);
}

int mlmcpi_lattice_sweep_draw_qoi(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                                  uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                                  uint32_t fuse, int32_t qoi_kind, double *d_qoi, int32_t *result_in, void *stream) {
  MLMCPI_REQUIRE(result_in, "result_in is NULL");
  MLMCPI_REQUIRE(qoi_kind >= 1 && qoi_kind <= 3, "qoi_kind %d: 1 = average plaquette, 2 = Q^2 / (4 pi^2), 3 = phi^2 (GFF)", qoi_kind);
  return sweep_draw_impl(act, const_cast<double *>(d_src), d_w0, d_w1, B, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse,
                         result_in, stream, qoi_kind, d_qoi);
}

int mlmcpi_lattice_sweep_draw_qoi_record(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1,
                                         uint32_t B, uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                         uint32_t sweep0, uint32_t fuse, int32_t qoi_kind, double *d_qoi, double *d_acc,
                                         int32_t *result_in, void *stream) {
  MLMCPI_REQUIRE(result_in && d_acc, "result_in or d_acc is NULL");
  MLMCPI_REQUIRE(qoi_kind >= 1 && qoi_kind <= 3, "qoi_kind %d: 1 = average plaquette, 2 = Q^2 / (4 pi^2), 3 = phi^2 (GFF)", qoi_kind);
  return sweep_draw_impl(act, const_cast<double *>(d_src), d_w0, d_w1, B, n_overrelax, n_heatbath, seed, chain0, sweep0, fuse,
                         result_in, stream, qoi_kind, d_qoi, d_acc);
}

int mlmcpi_qoi_phi_squared(const double *d_phi, uint32_t n_vertices, uint32_t B, double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_phi && d_out && B > 0 && n_vertices > 0, "bad arguments");
  // treat the field as a 1 x n strip: the reduction does not need the geometry
  uint32_t Mt = n_vertices, Mx = 1;
  if (n_vertices > 4096)
    for (uint32_t w = 4096; w >= 64; w >>= 1)
      if (n_vertices % w == 0) { Mt = w; Mx = n_vertices / w; break; }
  return launch_lattice_reduce<L_PHI2>(Mt, Mx, 0.0, d_phi, B, 1.0 / n_vertices, d_out, as_stream(stream));
}

int mlmcpi_qoi_avg_plaquette(const double *d_theta, uint32_t Mt, uint32_t Mx, uint32_t B, double *d_out,
                             void *stream) {
  MLMCPI_REQUIRE(d_theta && d_out && B > 0 && Mt > 1 && Mx > 1, "bad arguments");
  return launch_lattice_reduce<L_PLAQ>(Mt, Mx, 0.0, d_theta, B, 1.0 / ((double)Mx * Mt), d_out, as_stream(stream));
}

int mlmcpi_qoi_2d_susceptibility(const double *d_theta, uint32_t Mt, uint32_t Mx, uint32_t B, double *d_out,
                                 void *stream) {
  MLMCPI_REQUIRE(d_theta && d_out && B > 0 && Mt > 1 && Mx > 1, "bad arguments");
  return launch_lattice_reduce<L_CHARGE>(Mt, Mx, 0.0, d_theta, B, 1.0, d_out, as_stream(stream));
}

int mlmcpi_lattice_site_updates(const mlmcpi_lattice_action *act, double *d_state, uint32_t B, const uint32_t *d_sites,
                                uint32_t n, uint32_t site, int32_t heat, uint64_t seed, uint32_t chain0, uint32_t step,
                                void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_state && B > 0, "bad arguments");
  uint32_t size = 0;
  mlmcpi_lattice_state_size(act, &size);
  if (!d_sites) {
    MLMCPI_REQUIRE(site < size, "site %u out of range (%u entries)", site, size);
    n = 1;
  }
  if (n == 0) return MLMCPI_OK;
  const bool schw = act->kind == MLMCPI_SCHWINGER;
  const uint32_t *vs_table = nullptr;
  if (schw && heat && 2. * act->beta <= kVsKappaMax)
    if (int rc = vs_table_device(2. * act->beta, &vs_table)) return rc;
  const dim3 grid((B + 63) / 64), block(64);
  const RngKey key = make_key(seed, chain0, step);
  if (schw)
    hipLaunchKernelGGL((lattice_site_update_kernel<true>), grid, block, 0, as_stream(stream), act->Mt, act->Mx, act->beta, d_state, B,
                       d_sites, n, site, (int)heat, key, vs_table);
  else
    hipLaunchKernelGGL((lattice_site_update_kernel<false>), grid, block, 0, as_stream(stream), act->Mt, act->Mx, gff_mu2(*act), d_state,
                       B, d_sites, n, site, (int)heat, key, vs_table);
  MLMCPI_LAUNCH_CHECK("lattice_site_update_kernel");
  return MLMCPI_OK;
}

int mlmcpi_stats_window_record(double *d_state, const double *d_q, uint32_t B, uint32_t window, void *stream) {
  MLMCPI_REQUIRE(d_state && d_q && B > 0 && window > 0 && window <= 1024, "bad arguments");
  hipLaunchKernelGGL(stats_window_record_kernel, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), d_state, d_q, B, window);
  MLMCPI_LAUNCH_CHECK("stats_window_record_kernel");
  return MLMCPI_OK;
}

int mlmcpi_stats_accumulate(double *d_acc, const double *d_q, uint32_t B, void *stream) {
  MLMCPI_REQUIRE(d_acc && d_q && B > 0, "bad arguments");
  hipLaunchKernelGGL(stats_accumulate_kernel, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), d_acc, d_q, B);
  MLMCPI_LAUNCH_CHECK("stats_accumulate_kernel");
  return MLMCPI_OK;
}

}  // extern "C"

// =================================================================================================
// Generic HMC for 2-D actions (sampler/hmcsampler.cc:8-69), streaming form: momenta and the trial
// state live in HBM, one fused force + momentum + position kernel per leapfrog step (ping-pong on
// the trial state because neighbours need the old positions).
// =================================================================================================
namespace mlmcpi {

// p ~ N(0,1) per entry (Philox site = entry index), trial <- current
__global__ void __launch_bounds__(256)
    lat_hmc_init_kernel(uint32_t n, const double *__restrict__ x_cur, double *__restrict__ x_trial,
                        double *__restrict__ p, const int32_t *__restrict__ done, RngKey key0) {
  const uint32_t b = blockIdx.y;
  if (done[b]) return;
  RngKey key = key0;
  key.chain += b;
  const size_t off = (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) {
    p[off + l] = rng_normal0(key, l, P_MOMENTUM, 0);
    x_trial[off + l] = x_cur[off + l];
  }
}

// one leapfrog step: F(x_in); p -= dtp F; x_out = x_in + dtx p
template <int KIND>
__global__ void __launch_bounds__(256)
    lat_hmc_step_kernel(uint32_t Mt, uint32_t Mx, double coupling, const double *__restrict__ x_in,
                        double *__restrict__ x_out, double *__restrict__ p_all, const int32_t *__restrict__ done,
                        double dtp, double dtx) {
  const uint32_t b = blockIdx.y;
  if (done[b]) return;
  if (KIND == MLMCPI_GFF) {
    const double *phi = x_in + (size_t)b * Mt * Mx;
    double *out = x_out + (size_t)b * Mt * Mx, *p = p_all + (size_t)b * Mt * Mx;
    const double kappa = 4. + coupling;
    for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x) {
      const uint32_t jm = j == 0 ? Mx - 1 : j - 1, jp = j + 1 == Mx ? 0 : j + 1;
      for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
        const uint32_t im = i == 0 ? Mt - 1 : i - 1, ip = i + 1 == Mt ? 0 : i + 1;
        const size_t o = (size_t)j * Mt + i;
        double F = kappa * phi[o];
        F -= phi[(size_t)j * Mt + ip];
        F -= phi[(size_t)j * Mt + im];
        F -= phi[(size_t)jp * Mt + i];
        F -= phi[(size_t)jm * Mt + i];
        const double pn = p[o] - dtp * F;
        p[o] = pn;
        out[o] = phi[o] + dtx * pn;
      }
    }
  } else {
    // grid (ceil(force_waves / 4), B): one sine per plaquette (schwinger_force_band)
    const uint32_t wave_id = blockIdx.x * 4 + threadIdx.x / kWave;
    if (wave_id >= force_waves(Mt, Mx)) return;
    const double2 *t = (const double2 *)x_in + (size_t)b * Mt * Mx;
    double2 *out = (double2 *)x_out + (size_t)b * Mt * Mx, *p = (double2 *)p_all + (size_t)b * Mt * Mx;
    schwinger_force_band(t, Mt, Mx, coupling, wave_id, [&](uint32_t j, uint32_t i, double f0, double f1) {
      const size_t o = (size_t)j * Mt + i;
      double2 pn = p[o];
      pn.x -= dtp * f0;
      pn.y -= dtp * f1;
      p[o] = pn;
      const double2 xo = t[o];
      out[o] = make_double2(xo.x + dtx * pn.x, xo.y + dtx * pn.y);
    });
  }
}

// en4 = [4][B]: S0, T0, S1, T1 (already scaled).  hmcsampler.cc:50-67.
__global__ void __launch_bounds__(256)
    lat_hmc_accept_kernel(uint32_t n, double *__restrict__ x_cur, const double *__restrict__ x_trial,
                          const double *__restrict__ en4, uint32_t B, const int32_t *__restrict__ done_in,
                          int32_t *__restrict__ done_out, double *__restrict__ energies, RngKey key0) {
  const uint32_t b = blockIdx.y;
  if (done_in[b]) {
    if (blockIdx.x == 0 && threadIdx.x == 0) done_out[b] = 1;
    return;
  }
  const double S0 = en4[b], T0 = en4[B + b], S1 = en4[2 * B + b], T1 = en4[3 * B + b];
  const double dH = (S1 - S0) + (T1 - T0);
  bool acc;
  if (dH < 0.0) {
    acc = true;
  } else {
    RngKey key = key0;
    key.chain += b;
    double u, v;
    rng_uniforms(key, 0, P_ACCEPT, 0, u, v);
    acc = u < exp(-dH);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    done_out[b] = acc ? 1 : 0;
    if (energies) {
      energies[4 * b + 0] = S0; energies[4 * b + 1] = T0; energies[4 * b + 2] = S1; energies[4 * b + 3] = T1;
    }
  }
  if (!acc) return;
  const size_t off = (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x)
    x_cur[off + l] = x_trial[off + l];
}

static size_t align256_(size_t n) { return (n + 255) & ~(size_t)255; }

static int lattice_energy(const mlmcpi_lattice_action *act, const double *d_phi, uint32_t B, double *d_S, hipStream_t st) {
  if (act->kind == MLMCPI_GFF)
    return launch_lattice_reduce<L_GFF_ENERGY>(act->Mt, act->Mx, gff_mu2(*act), d_phi, B, 0.5, d_S, st);
  return launch_lattice_reduce<L_SCHW_ENERGY>(act->Mt, act->Mx, 0.0, d_phi, B, act->beta, d_S, st);
}

static int kinetic_energy(const double *d_p, uint32_t n, uint32_t B, double *d_T, hipStream_t st) {
  uint32_t w = n, h = 1;
  if (n > 4096)
    for (uint32_t c = 4096; c >= 64; c >>= 1)
      if (n % c == 0) { w = c; h = n / c; break; }
  return launch_lattice_reduce<L_PHI2>(w, h, 0.0, d_p, B, 0.5, d_T, st);
}

}  // namespace mlmcpi

extern "C" {

// workspace: p | trial A | trial B | energies [4][B] | flags [2][B]
int mlmcpi_lattice_hmc_workspace_bytes(const mlmcpi_lattice_action *act, uint32_t B, size_t *bytes) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(bytes && B > 0, "bad arguments");
  uint32_t n = 0;
  mlmcpi_lattice_state_size(act, &n);
  *bytes = 3 * align256_((size_t)B * n * 8) + align256_((size_t)4 * B * 8) + align256_((size_t)2 * B * 4);
  return MLMCPI_OK;
}

int mlmcpi_lattice_hmc_draw(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint32_t nt, double dt,
                            uint32_t n_rep, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                            int32_t *d_accept, double *d_energies, void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_phi && d_work && B > 0 && n_rep > 0, "bad arguments");
  uint32_t n = 0;
  mlmcpi_lattice_state_size(act, &n);
  hipStream_t st = as_stream(stream);
  char *w = (char *)d_work;
  const size_t sb = align256_((size_t)B * n * 8);
  double *p = (double *)w, *xa = (double *)(w + sb), *xb = (double *)(w + 2 * sb);
  double *en4 = (double *)(w + 3 * sb);
  int32_t *flags = (int32_t *)(w + 3 * sb + align256_((size_t)4 * B * 8));
  MLMCPI_HIP_TRY(hipMemsetAsync(flags, 0, (size_t)2 * B * 4, st));
  uint32_t nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  const dim3 lin_grid(nb, B), row_grid(row_blocks(act->Mx, B), B), block(256);
  const double coupling = act->kind == MLMCPI_GFF ? gff_mu2(*act) : act->beta;
  for (uint32_t r = 0; r < n_rep; ++r) {
    const int32_t *done_in = flags + (size_t)(r & 1) * B;
    int32_t *done_out = flags + (size_t)((r + 1) & 1) * B;
    const RngKey key = make_key(seed, chain0, traj0 + r);
    hipLaunchKernelGGL(lat_hmc_init_kernel, lin_grid, block, 0, st, n, (const double *)d_phi, xa, p, done_in, key);
    MLMCPI_LAUNCH_CHECK("lat_hmc_init_kernel");
    if (int rc = lattice_energy(act, d_phi, B, en4, st)) return rc;
    if (int rc = kinetic_energy(p, n, B, en4 + B, st)) return rc;
    double *src = xa, *dst = xb;
    for (uint32_t k = 0; k <= nt; ++k) {
      const double dtp = (k == 0 || k == nt) ? 0.5 * dt : dt;
      const double dtx = (k == nt) ? 0.0 : dt;
      if (act->kind == MLMCPI_GFF)
        hipLaunchKernelGGL(lat_hmc_step_kernel<MLMCPI_GFF>, row_grid, block, 0, st, act->Mt, act->Mx, coupling,
                           (const double *)src, dst, p, done_in, dtp, dtx);
      else
        hipLaunchKernelGGL(lat_hmc_step_kernel<MLMCPI_SCHWINGER>, dim3((force_waves(act->Mt, act->Mx) + 3) / 4, B), block, 0, st, act->Mt, act->Mx, coupling,
                           (const double *)src, dst, p, done_in, dtp, dtx);
      MLMCPI_LAUNCH_CHECK("lat_hmc_step_kernel");
      double *tmp = src; src = dst; dst = tmp;
    }
    if (int rc = lattice_energy(act, src, B, en4 + 2 * (size_t)B, st)) return rc;
    if (int rc = kinetic_energy(p, n, B, en4 + 3 * (size_t)B, st)) return rc;
    hipLaunchKernelGGL(lat_hmc_accept_kernel, lin_grid, block, 0, st, n, d_phi, (const double *)src,
                       (const double *)en4, B, done_in, done_out, d_energies, key);
    MLMCPI_LAUNCH_CHECK("lat_hmc_accept_kernel");
  }
  if (d_accept)
    MLMCPI_HIP_TRY(hipMemcpyAsync(d_accept, flags + (size_t)(n_rep & 1) * B, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  return MLMCPI_OK;
}

}  // extern "C"

// =================================================================================================
// Transfers between lattice levels: Action::copy_from_fine / copy_from_coarse for the 2-D actions.
//   Schwinger  quenchedschwingeraction.cc:92-195 (three coarsening cases: both, temporal, spatial)
//   GFF        gffaction.cc:97-118 (fine2coarse_map of lattice2d.cc:126-134; unrotated coarsenings)
// Grid (rows, B); rt, rx in {1, 2} are the coarsening factors in the temporal / spatial direction.
// =================================================================================================
namespace mlmcpi {

__global__ void __launch_bounds__(256)
    schwinger_copy_from_fine_kernel(uint32_t Mt, uint32_t Mx, uint32_t rt, uint32_t rx, const double2 *__restrict__ fine_all,
                                    double2 *__restrict__ coarse_all) {
  const uint32_t b = blockIdx.y, Mtf = Mt * rt, Mxf = Mx * rx;  // Mt, Mx: coarse extents
  const double2 *fine = fine_all + (size_t)b * Mtf * Mxf;
  double2 *coarse = coarse_all + (size_t)b * Mt * Mx;
  for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x)
    for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
      const size_t f = (size_t)(rx * j) * Mtf + rt * i;
      // mu = 0 links add up along the temporal direction, mu = 1 links along the spatial one
      const double t0 = (rt == 2) ? fine[f].x + fine[f + 1].x : fine[f].x;
      const double t1 = (rx == 2) ? fine[f].y + fine[f + Mtf].y : fine[f].y;
      coarse[(size_t)j * Mt + i] = make_double2(mod_2pi(t0), mod_2pi(t1));
    }
}

// writes only the links the reference writes (the others are filled by the conditioned fine action)
__global__ void __launch_bounds__(256)
    schwinger_copy_from_coarse_kernel(uint32_t Mt, uint32_t Mx, uint32_t rt, uint32_t rx,
                                      const double2 *__restrict__ coarse_all, double *__restrict__ fine_all) {
  const uint32_t b = blockIdx.y, Mtf = Mt * rt, Mxf = Mx * rx;
  const double2 *coarse = coarse_all + (size_t)b * Mt * Mx;
  double *fine = fine_all + (size_t)b * 2 * Mtf * Mxf;
  for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x)
    for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
      const double2 c = coarse[(size_t)j * Mt + i];
      const size_t f = 2 * ((size_t)(rx * j) * Mtf + rt * i);  // link index of (rt i, rx j, 0)
      if (rt == 2) {
        fine[f] = 0.5 * c.x;
        fine[f + 2] = 0.5 * c.x;
      } else {
        fine[f] = c.x;
      }
      if (rx == 2) {
        fine[f + 1] = 0.5 * c.y;
        fine[f + 2 * Mtf + 1] = 0.5 * c.y;
      } else {
        fine[f + 1] = c.y;
      }
    }
}

// to_coarse != 0: coarse(i,j) = fine(rt i, rx j); else fine(rt i, rx j) = coarse(i,j)
__global__ void __launch_bounds__(256)
    vertex_transfer_kernel(uint32_t Mt, uint32_t Mx, uint32_t rt, uint32_t rx, double *__restrict__ fine_all,
                           double *__restrict__ coarse_all, int to_coarse) {
  const uint32_t b = blockIdx.y, Mtf = Mt * rt, Mxf = Mx * rx;
  double *fine = fine_all + (size_t)b * Mtf * Mxf, *coarse = coarse_all + (size_t)b * Mt * Mx;
  for (uint32_t j = blockIdx.x; j < Mx; j += gridDim.x)
    for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
      const size_t f = (size_t)(rx * j) * Mtf + rt * i, c = (size_t)j * Mt + i;
      if (to_coarse) coarse[c] = fine[f]; else fine[f] = coarse[c];
    }
}

}  // namespace mlmcpi

extern "C" {

static int check_levels(const mlmcpi_lattice_action *fine, uint32_t rt, uint32_t rx) {
  if (int rc = check_lattice_dims(fine)) return rc;
  if (!((rt == 1 || rt == 2) && (rx == 1 || rx == 2) && rt * rx > 1))
    return fail(MLMCPI_ERR_INVALID, "cannot copy between these lattices (coarsening factors %u x %u)", rt, rx);
  if (fine->Mt % rt || fine->Mx % rx) return fail(MLMCPI_ERR_INVALID, "fine lattice %u x %u cannot be coarsened by %u x %u", fine->Mt, fine->Mx, rt, rx);
  return MLMCPI_OK;
}

int mlmcpi_lattice_copy_from_fine(const mlmcpi_lattice_action *fine, uint32_t rt, uint32_t rx, const double *d_fine,
                                  double *d_coarse, uint32_t B, void *stream) {
  if (int rc = check_levels(fine, rt, rx)) return rc;
  MLMCPI_REQUIRE(d_fine && d_coarse && B > 0, "bad arguments");
  const uint32_t Mt = fine->Mt / rt, Mx = fine->Mx / rx;
  dim3 grid(row_blocks(Mx, B), B), block(256);
  if (fine->kind == MLMCPI_SCHWINGER)
    hipLaunchKernelGGL(schwinger_copy_from_fine_kernel, grid, block, 0, as_stream(stream), Mt, Mx, rt, rx,
                       (const double2 *)d_fine, (double2 *)d_coarse);
  else
    hipLaunchKernelGGL(vertex_transfer_kernel, grid, block, 0, as_stream(stream), Mt, Mx, rt, rx, (double *)d_fine, d_coarse, 1);
  MLMCPI_LAUNCH_CHECK("copy_from_fine kernel");
  return MLMCPI_OK;
}

int mlmcpi_lattice_copy_from_coarse(const mlmcpi_lattice_action *fine, uint32_t rt, uint32_t rx, const double *d_coarse,
                                    double *d_fine, uint32_t B, void *stream) {
  if (int rc = check_levels(fine, rt, rx)) return rc;
  MLMCPI_REQUIRE(d_fine && d_coarse && B > 0, "bad arguments");
  const uint32_t Mt = fine->Mt / rt, Mx = fine->Mx / rx;
  dim3 grid(row_blocks(Mx, B), B), block(256);
  if (fine->kind == MLMCPI_SCHWINGER)
    hipLaunchKernelGGL(schwinger_copy_from_coarse_kernel, grid, block, 0, as_stream(stream), Mt, Mx, rt, rx,
                       (const double2 *)d_coarse, d_fine);
  else
    hipLaunchKernelGGL(vertex_transfer_kernel, grid, block, 0, as_stream(stream), Mt, Mx, rt, rx, d_fine, (double *)d_coarse, 0);
  MLMCPI_LAUNCH_CHECK("copy_from_coarse kernel");
  return MLMCPI_OK;
}

}  // extern "C"

// =================================================================================================
// Two-level Metropolis step on the Schwinger lattice with semi-coarsening (one direction halved):
//   TwoLevelMetropolisStep::draw                               montecarlo/twolevelmetropolisstep.cc:35-89
//   QuenchedSchwingerAction::copy_from_{coarse,fine}           action/qft/quenchedschwingeraction.cc:92-195
//   QuenchedSchwingerSemiConditionedFineAction::{fill_fine_points,evaluate}
//                                                              action/qft/quenchedschwingerconditionedfineaction.cc:130-204,332-379
// One coarse cell (i, j) owns two fine vertices.  In the coarsened direction d the coarse link splits into a
// pair  a0 = theta_c/2 + dtheta, a1 = theta_c/2 - dtheta  (dtheta ~ U(-pi, pi)); the coarse link in the other
// direction is copied; the remaining fine link (direction 1-d, between the two halves) is drawn from its
// heat-bath conditional (ExpCos) given the staples  theta_p = s0 + b0 - a0,  theta_m = a1 + s2 - b1, where
// (b0, b1) is the pair of the neighbouring cell and s2 the copied link of the next cell.  The neighbour's pair
// is recomputed from its own Philox stream (no exchange).  RNG: dtheta of coarse cell c = Philox(site c,
// P_FILLIN, 0); the ExpCos draw of fine link l = von Mises stream (site l, kVmFillin).
// =================================================================================================
namespace mlmcpi {

// theta (fine, double2 per vertex) in reference order; rt = 2: temporal coarsening, else spatial (rx = 2).
// partial[(b * gridDim.x + blockIdx.x) * 2 + {0, 1}] = CFA sums of (theta', theta).
__global__ void __launch_bounds__(256)
    schwinger_twolevel_propose_kernel(uint32_t Mtc, uint32_t Mxc, uint32_t rt, double beta,
                                      const double2 *__restrict__ coarse_all, const double2 *__restrict__ theta_all,
                                      double2 *__restrict__ prime_all, double *__restrict__ partial, RngKey key0) {
  __shared__ double red[2 * 4];
  const uint32_t b = blockIdx.y;
  const uint32_t Mtf = (rt == 2) ? 2 * Mtc : Mtc, Mxf = (rt == 2) ? Mxc : 2 * Mxc;
  const double2 *coarse = coarse_all + (size_t)b * Mtc * Mxc;
  const double2 *theta = theta_all + (size_t)b * Mtf * Mxf;
  double2 *prime = prime_all + (size_t)b * Mtf * Mxf;
  RngKey key = key0;
  key.chain += b;
  double acc[2] = {0.0, 0.0};
  for (uint32_t j = blockIdx.x; j < Mxc; j += gridDim.x) {
    const uint32_t jp = j + 1 == Mxc ? 0 : j + 1;
    for (uint32_t i = threadIdx.x; i < Mtc; i += blockDim.x) {
      const uint32_t ip = i + 1 == Mtc ? 0 : i + 1;
      const uint32_t c = j * Mtc + i;
      // neighbour cell in the direction the pair does NOT point in, and the next cell along the pair
      const uint32_t cn = (rt == 2) ? jp * Mtc + i : j * Mtc + ip;
      const uint32_t cs = (rt == 2) ? j * Mtc + ip : jp * Mtc + i;
      const double2 lc = coarse[c], ln = coarse[cn], ls = coarse[cs];
      const double pair_c = (rt == 2) ? lc.x : lc.y, pair_n = (rt == 2) ? ln.x : ln.y;
      const double s0 = (rt == 2) ? lc.y : lc.x, s2 = (rt == 2) ? ls.y : ls.x;
      double u, v;
      rng_uniforms(key, c, P_FILLIN, 0, u, v);
      const double d_c = (2. * u - 1.) * kPi;
      rng_uniforms(key, cn, P_FILLIN, 0, u, v);
      const double d_n = (2. * u - 1.) * kPi;
      const double a0 = mod_2pi(0.5 * pair_c + d_c), a1 = mod_2pi(0.5 * pair_c - d_c);
      const double b0 = mod_2pi(0.5 * pair_n + d_n), b1 = mod_2pi(0.5 * pair_n - d_n);
      const double th_p = mod_2pi(s0 + b0 - a0), th_m = mod_2pi(a1 + s2 - b1);
      // fine vertices of this cell and the linear index of the filled link
      size_t v0, v1;
      uint32_t l_fill;
      if (rt == 2) {
        v0 = (size_t)j * Mtf + 2 * i;
        v1 = v0 + 1;
        l_fill = 2 * (uint32_t)v1 + 1;
      } else {
        v0 = (size_t)(2 * j) * Mtf + i;
        v1 = v0 + Mtf;
        l_fill = 2 * (uint32_t)v1;
      }
      const double fill = expcos_draw(key, l_fill, beta, th_p, th_m, kVmFillin);
      if (rt == 2) {
        prime[v0] = make_double2(a0, s0);
        prime[v1] = make_double2(a1, fill);
      } else {
        prime[v0] = make_double2(s0, a0);
        prime[v1] = make_double2(fill, a1);
      }
      acc[0] += expcos_neg_log_pdf(beta, fill, th_p, th_m);
      // the same term for the current fine state
      double t_a0, t_a1, t_b0, t_b1, t_s0, t_s2, t_x;
      if (rt == 2) {
        const uint32_t i2 = 2 * i, i2p = (i2 + 2 == Mtf) ? 0 : i2 + 2;
        const double2 q0 = theta[(size_t)j * Mtf + i2], q1 = theta[(size_t)j * Mtf + i2 + 1];
        const double2 n0 = theta[(size_t)jp * Mtf + i2], n1 = theta[(size_t)jp * Mtf + i2 + 1];
        t_a0 = q0.x; t_a1 = q1.x; t_s0 = q0.y; t_x = q1.y; t_b0 = n0.x; t_b1 = n1.x;
        t_s2 = theta[(size_t)j * Mtf + i2p].y;
      } else {
        const uint32_t j2 = 2 * j, j2p = (j2 + 2 == Mxf) ? 0 : j2 + 2;
        const double2 q0 = theta[(size_t)j2 * Mtf + i], q1 = theta[(size_t)(j2 + 1) * Mtf + i];
        const double2 n0 = theta[(size_t)j2 * Mtf + ip], n1 = theta[(size_t)(j2 + 1) * Mtf + ip];
        t_a0 = q0.y; t_a1 = q1.y; t_s0 = q0.x; t_x = q1.x; t_b0 = n0.y; t_b1 = n1.y;
        t_s2 = theta[(size_t)j2p * Mtf + i].x;
      }
      acc[1] += expcos_neg_log_pdf(beta, mod_2pi(t_x), mod_2pi(-t_a0 + t_s0 + t_b0), mod_2pi(t_a1 + t_s2 - t_b1));
    }
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    partial[((size_t)b * gridDim.x + blockIdx.x) * 2 + 0] = acc[0];
    partial[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = acc[1];
  }
}

// ---- coarsening in both directions: QuenchedSchwingerConditionedFineAction (quenchedschwingerconditionedfineaction.cc:7-78,
// 207-289).  Three kernels: (A) per coarse cell, steps 1 and 2 -- uniform shifts of the two split coarse links, then the
// two interior spatial links from the Bessel-product law of their sum (the staples of the 2 x 2 block need the split
// links of the cells (i+1, j) and (i, j+1), recomputed from those cells' Philox streams); (B) per fine link
// (i, 2 jc + 1, 0), step 3 -- the ExpCos heat-bath conditional, reading what (A) wrote; (C) the conditioned fine action
// of a state.
__device__ __forceinline__ void split_pair(const RngKey &key, uint32_t cell, double2 coarse_link, double (&t)[2], double (&x)[2]) {
  double u, v;
  rng_uniforms(key, cell, P_FILLIN, 0, u, v);
  const double dt = (2. * u - 1.) * kPi, dx = (2. * v - 1.) * kPi;
  t[0] = mod_2pi(0.5 * coarse_link.x + dt);
  t[1] = mod_2pi(0.5 * coarse_link.x - dt);
  x[0] = mod_2pi(0.5 * coarse_link.y + dx);
  x[1] = mod_2pi(0.5 * coarse_link.y - dx);
}

__global__ void __launch_bounds__(256)
    schwinger_both_fill_kernel(uint32_t Mtc, uint32_t Mxc, BesselFill P, const double2 *__restrict__ coarse_all,
                               double2 *__restrict__ prime_all, RngKey key0) {
  const uint32_t b = blockIdx.y, Mtf = 2 * Mtc;
  const double2 *coarse = coarse_all + (size_t)b * Mtc * Mxc;
  double *prime = (double *)(prime_all + (size_t)b * 4 * Mtc * Mxc);
  RngKey key = key0;
  key.chain += b;
  for (uint32_t j = blockIdx.x; j < Mxc; j += gridDim.x) {
    const uint32_t jp = j + 1 == Mxc ? 0 : j + 1;
    for (uint32_t i = threadIdx.x; i < Mtc; i += blockDim.x) {
      const uint32_t ip = i + 1 == Mtc ? 0 : i + 1;
      const uint32_t c = j * Mtc + i, c_t = j * Mtc + ip, c_x = jp * Mtc + i;
      double t[2], x[2], tt[2], tx[2], xt[2], xx[2];
      split_pair(key, c, coarse[c], t, x);         // this cell
      split_pair(key, c_t, coarse[c_t], tt, tx);   // cell (i+1, j): its spatial pair closes the block on the right
      split_pair(key, c_x, coarse[c_x], xt, xx);   // cell (i, j+1): its temporal pair closes the block on top
      // theta_p = th(2i+1,2j,0) + th(2i+2,2j,1) + th(2i+2,2j+1,1) - th(2i+1,2j+2,0)
      const double theta_p = mod_2pi(t[1] + tx[0] + tx[1] - xt[1]);
      // theta_m = th(2i,2j,1) + th(2i,2j+1,1) + th(2i,2j+2,0) - th(2i,2j,0)
      const double theta_m = mod_2pi(x[0] + x[1] + xt[0] - t[0]);
      const double tilde = P.approximate ? approx_bessel_draw(key, c, P.beta, theta_p, theta_m)
                                         : bessel_product_draw(key, c, P, theta_p, theta_m);
      double u, v;
      rng_uniforms(key, c, P_FILLIN, 1, u, v);
      const double d = (2. * u - 1.) * kPi;
      // fine vertices (2i, 2j), (2i+1, 2j), (2i, 2j+1), (2i+1, 2j+1); link index = 2 * vertex + mu
      const size_t v00 = (size_t)(2 * j) * Mtf + 2 * i, v01 = v00 + Mtf;
      prime[2 * v00] = t[0];
      prime[2 * v00 + 1] = x[0];
      prime[2 * (v00 + 1)] = t[1];
      prime[2 * (v00 + 1) + 1] = mod_2pi(0.5 * tilde + d);
      prime[2 * v01 + 1] = x[1];
      prime[2 * (v01 + 1) + 1] = mod_2pi(0.5 * tilde - d);
    }
  }
}

// step 3: links (i, 2 jc + 1, 0), i = 0..Mt-1, jc = 0..Mx/2-1
__global__ void __launch_bounds__(256)
    schwinger_both_rows_kernel(uint32_t Mt, uint32_t Mx, double beta, double2 *__restrict__ prime_all, RngKey key0) {
  const uint32_t b = blockIdx.y;
  double *prime = (double *)(prime_all + (size_t)b * Mt * Mx);
  RngKey key = key0;
  key.chain += b;
  for (uint32_t jc = blockIdx.x; jc < Mx / 2; jc += gridDim.x) {
    const uint32_t j0 = 2 * jc, j1 = j0 + 1, j2 = (j0 + 2 == Mx) ? 0 : j0 + 2;
    for (uint32_t i = threadIdx.x; i < Mt; i += blockDim.x) {
      const uint32_t ip = i + 1 == Mt ? 0 : i + 1;
      auto link = [&](uint32_t ii, uint32_t jj, uint32_t mu) { return prime[2 * ((size_t)jj * Mt + ii) + mu]; };
      const double theta_p = mod_2pi(link(i, j0, 0) + link(ip, j0, 1) - link(i, j0, 1));
      const double theta_m = mod_2pi(link(i, j1, 1) + link(i, j2, 0) - link(ip, j1, 1));
      const uint32_t l = 2 * (j1 * Mt + i);
      prime[l] = expcos_draw(key, l, beta, theta_p, theta_m, kVmFillin);
    }
  }
}

// partial[(b * gridDim.x + blockIdx.x) * 2 + slot] = conditioned fine action of `state`, one 2 x 2 block per thread
__global__ void __launch_bounds__(256)
    schwinger_both_cfa_kernel(uint32_t Mt, uint32_t Mx, BesselFill P, const double2 *__restrict__ state_all,
                              double *__restrict__ partial, uint32_t slot) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.y;
  const double *th = (const double *)(state_all + (size_t)b * Mt * Mx);
  auto link = [&](uint32_t ii, uint32_t jj, uint32_t mu) { return th[2 * ((size_t)jj * Mt + ii) + mu]; };
  double acc[1] = {0.0};
  for (uint32_t jc = blockIdx.x; jc < Mx / 2; jc += gridDim.x) {
    const uint32_t j0 = 2 * jc, j1 = j0 + 1, j2 = (j0 + 2 == Mx) ? 0 : j0 + 2;
    for (uint32_t ic = threadIdx.x; ic < Mt / 2; ic += blockDim.x) {
      const uint32_t i0 = 2 * ic, i1 = i0 + 1, i2 = (i0 + 2 == Mt) ? 0 : i0 + 2;
      if (!P.approximate) {
        const double phi_12 = +link(i0, j1, 1) + link(i0, j2, 0);
        const double phi_23 = +link(i1, j2, 0) - link(i2, j1, 1);
        const double phi_34 = -link(i1, j0, 0) - link(i2, j0, 1);
        const double phi_41 = -link(i0, j0, 0) + link(i0, j0, 1);
        const double theta_1 = +link(i0, j1, 0), theta_2 = -link(i1, j1, 1), theta_3 = -link(i1, j1, 0),
                     theta_4 = +link(i1, j0, 1);
        const double Phi = phi_12 + phi_23 + phi_34 + phi_41;
        acc[0] -= P.beta * (cos(theta_1 - theta_2 - phi_12) + cos(theta_2 - theta_3 - phi_23) +
                            cos(theta_3 - theta_4 - phi_34) + cos(theta_4 - theta_1 - phi_41));
        acc[0] -= log(bessel_znorm_inv_rescaled(P, Phi));
      } else {
        const double phi_p = mod_2pi(+link(i1, j0, 0) + link(i2, j0, 1) + link(i2, j1, 1) - link(i1, j2, 0));
        const double phi_m = mod_2pi(-link(i0, j0, 0) + link(i0, j0, 1) + link(i0, j1, 1) + link(i0, j2, 0));
        const double theta = mod_2pi(+link(i1, j0, 1) + link(i1, j1, 1));
        acc[0] -= log(approx_bessel_pdf(P.beta, theta, phi_p, phi_m));
        // the two horizontal links (i0, j1, 0), (i1, j1, 0) of this block
        for (uint32_t r = 0; r < 2; ++r) {
          const uint32_t i = i0 + r, ip = (r == 0) ? i1 : i2;
          const double hp = mod_2pi(-link(i, j0, 1) + link(i, j0, 0) + link(ip, j0, 1));
          const double hm = mod_2pi(+link(i, j1, 1) + link(i, j2, 0) - link(ip, j1, 1));
          acc[0] += expcos_neg_log_pdf(P.beta, mod_2pi(link(i, j1, 0)), hp, hm);
        }
      }
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) partial[((size_t)b * gridDim.x + blockIdx.x) * 2 + slot] = acc[0];
}

// QuenchedSchwingerGaussianConditionedFineAction (quenchedschwingerconditionedfineaction.cc:81-134, 293-327): the Gaussian
// variant of the fill-in for lattices coarsened in both directions.  One thread per coarse cell = one 2 x 2 block of fine
// vertices: the perimeter links come from the uniform splits of the coarse links (this cell's and, for the far sides, the
// cells (i+1, j) and (i, j+1), recomputed from their Philox streams as in schwinger_both_fill_kernel); the four interior
// links from GaussianFillinDistribution::draw.
__global__ void __launch_bounds__(256)
    schwinger_gauss_fill_kernel(uint32_t Mtc, uint32_t Mxc, double beta, const double2 *__restrict__ coarse_all,
                                double2 *__restrict__ prime_all, RngKey key0) {
  const uint32_t b = blockIdx.y, Mtf = 2 * Mtc;
  const double2 *coarse = coarse_all + (size_t)b * Mtc * Mxc;
  double2 *prime = prime_all + (size_t)b * 4 * Mtc * Mxc;
  RngKey key = key0;
  key.chain += b;
  for (uint32_t j = blockIdx.x; j < Mxc; j += gridDim.x) {
    const uint32_t jp = j + 1 == Mxc ? 0 : j + 1;
    for (uint32_t i = threadIdx.x; i < Mtc; i += blockDim.x) {
      const uint32_t ip = i + 1 == Mtc ? 0 : i + 1;
      const uint32_t c = j * Mtc + i, c_t = j * Mtc + ip, c_x = jp * Mtc + i;
      double t[2], x[2], tt[2], tx[2], xt[2], xx[2];
      split_pair(key, c, coarse[c], t, x);
      split_pair(key, c_t, coarse[c_t], tt, tx);
      split_pair(key, c_x, coarse[c_x], xt, xx);
      const double phi_12 = mod_2pi(+x[1] + xt[0]);    // th(2i, 2j+1, 1) + th(2i, 2j+2, 0)
      const double phi_23 = mod_2pi(+xt[1] - tx[1]);   // th(2i+1, 2j+2, 0) - th(2i+2, 2j+1, 1)
      const double phi_34 = mod_2pi(-tx[0] - t[1]);    // -th(2i+2, 2j, 1) - th(2i+1, 2j, 0)
      const double phi_41 = mod_2pi(-t[0] + x[0]);     // -th(2i, 2j, 0) + th(2i, 2j, 1)
      double th[4];
      gaussfill_draw(key, c, beta, phi_12, phi_23, phi_34, phi_41, th);
      const size_t v00 = (size_t)(2 * j) * Mtf + 2 * i, v01 = v00 + Mtf;
      prime[v00] = make_double2(t[0], x[0]);
      prime[v00 + 1] = make_double2(t[1], +th[3]);     // (2i+1, 2j): temporal half, interior spatial link theta_4
      prime[v01] = make_double2(+th[0], x[1]);         // (2i, 2j+1): interior temporal link theta_1, spatial half
      prime[v01 + 1] = make_double2(-th[2], -th[1]);   // (2i+1, 2j+1): -theta_3, -theta_2
    }
  }
}

// partial[(b * gridDim.x + blockIdx.x) * 2 + slot] = -sum log pdf over the 2 x 2 blocks of `state`
__global__ void __launch_bounds__(256)
    schwinger_gauss_cfa_kernel(uint32_t Mt, uint32_t Mx, double beta, const double2 *__restrict__ state_all, double *__restrict__ partial,
                               uint32_t slot) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.y;
  const double *th = (const double *)(state_all + (size_t)b * Mt * Mx);
  auto link = [&](uint32_t ii, uint32_t jj, uint32_t mu) { return th[2 * ((size_t)jj * Mt + ii) + mu]; };
  double acc[1] = {0.0};
  for (uint32_t jc = blockIdx.x; jc < Mx / 2; jc += gridDim.x) {
    const uint32_t j0 = 2 * jc, j1 = j0 + 1, j2 = (j0 + 2 == Mx) ? 0 : j0 + 2;
    for (uint32_t ic = threadIdx.x; ic < Mt / 2; ic += blockDim.x) {
      const uint32_t i0 = 2 * ic, i1 = i0 + 1, i2 = (i0 + 2 == Mt) ? 0 : i0 + 2;
      const double phi_12 = mod_2pi(+link(i0, j1, 1) + link(i0, j2, 0));
      const double phi_23 = mod_2pi(+link(i1, j2, 0) - link(i2, j1, 1));
      const double phi_34 = mod_2pi(-link(i2, j0, 1) - link(i1, j0, 0));
      const double phi_41 = mod_2pi(-link(i0, j0, 0) + link(i0, j0, 1));
      const double theta_1 = mod_2pi(+link(i0, j1, 0)), theta_2 = mod_2pi(-link(i1, j1, 1)), theta_3 = mod_2pi(-link(i1, j1, 0)),
                   theta_4 = mod_2pi(+link(i1, j0, 1));
      acc[0] -= log(gaussfill_pdf(beta, theta_1, theta_2, theta_3, theta_4, phi_12, phi_23, phi_34, phi_41));
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) partial[((size_t)b * gridDim.x + blockIdx.x) * 2 + slot] = acc[0];
}

// en4 = [4][B]: S_f(theta'), S_f(theta), S_c(theta_C), S_c(phi_c); twolevelmetropolisstep.cc:46-84
__global__ void __launch_bounds__(256)
    lattice_twolevel_accept_kernel(uint32_t n, double *__restrict__ theta, const double *__restrict__ theta_prime,
                                   const double *__restrict__ en4, const double *__restrict__ cfa_partial, uint32_t nblk,
                                   uint32_t B, int32_t *__restrict__ accept, double *__restrict__ terms, RngKey key0) {
  const uint32_t b = blockIdx.y;
  double cfa_p = 0.0, cfa_c = 0.0;
  for (uint32_t k = 0; k < nblk; ++k) {
    cfa_p += cfa_partial[((size_t)b * nblk + k) * 2 + 0];
    cfa_c += cfa_partial[((size_t)b * nblk + k) * 2 + 1];
  }
  const double dS_fine = en4[b] - en4[B + b];
  const double dS_coarse = en4[2 * B + b] - en4[3 * B + b];
  const double dS_trial = cfa_c - cfa_p;
  const double dS = dS_fine + dS_coarse + dS_trial;
  bool acc;
  if (dS < 0.0) {
    acc = true;
  } else {
    RngKey key = key0;
    key.chain += b;
    double u, v;
    rng_uniforms(key, 0, P_ACCEPT2, 0, u, v);
    acc = u < exp(-dS);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    accept[b] = acc ? 1 : 0;
    if (terms) {
      terms[3 * b + 0] = dS_fine; terms[3 * b + 1] = dS_coarse; terms[3 * b + 2] = dS_trial;
    }
  }
  if (!acc) return;
  const size_t off = (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x)
    theta[off + l] = theta_prime[off + l];
}

}  // namespace mlmcpi

extern "C" {

// besselproductdistribution.hh:44-72: I0(2 beta), the envelope width and the Fourier coefficients alpha_k of the
// normalisation constant (k <= 16, sums truncated at n, m <= 32)
static BesselFill make_bessel_fill(double beta) {
  static BesselFill cached;
  static bool have = false;
  static std::mutex guard;
  std::lock_guard<std::mutex> lock(guard);
  if (have && cached.beta == beta) return cached;
  BesselFill P;
  P.beta = beta;
  P.approximate = beta > 8.0 ? 1 : 0;
  P.I0_twobeta = std::cyl_bessel_i(0.0, 2. * beta);
  P.sigma_beta = kPi / std::sqrt(2. * std::log(P.I0_twobeta));
  double logfact[65];
  logfact[0] = logfact[1] = 0.0;
  for (int n = 2; n <= 64; ++n) logfact[n] = logfact[n - 1] + std::log((double)n);
  auto log_binom = [&](int n, int k) { return logfact[n] - logfact[k] - logfact[n - k]; };
  double alpha0 = 1.0;
  for (int k = 0; k <= 16; ++k) {
    double sum = 0.0;
    for (int n = k; n <= 32; ++n)
      for (int m = k; m <= 32; ++m)
        sum += std::pow(0.5 * beta, 2.0 * (n + m)) *
               std::exp(log_binom(2 * n, n - k) + log_binom(2 * m, m - k) - 2 * (logfact[n] + logfact[m]));
    const double alpha = ((k == 0) ? 2 : 4) * kPi * sum;
    if (k == 0) alpha0 = alpha;
    P.alphaZ[k] = (k == 0) ? alpha : alpha / alpha0;
  }
  cached = P;
  have = true;
  return P;
}

static int check_twolevel(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse, uint32_t *rt, uint32_t *rx) {
  if (int rc = check_lattice(fine)) return rc;
  if (int rc = check_lattice(coarse)) return rc;
  if (fine->kind != MLMCPI_SCHWINGER || coarse->kind != MLMCPI_SCHWINGER)
    return fail(MLMCPI_ERR_UNSUPPORTED, "two-level step: only the quenched Schwinger action has a device conditioned fine action");
  *rt = (coarse->Mt && fine->Mt == 2 * coarse->Mt) ? 2 : (fine->Mt == coarse->Mt ? 1 : 0);
  *rx = (coarse->Mx && fine->Mx == 2 * coarse->Mx) ? 2 : (fine->Mx == coarse->Mx ? 1 : 0);
  if (*rt == 0 || *rx == 0 || *rt * *rx == 1)
    return fail(MLMCPI_ERR_INVALID, "invalid coarsening for fill-in (%u x %u from %u x %u)", coarse->Mt, coarse->Mx, fine->Mt, fine->Mx);
  return MLMCPI_OK;
}

// workspace: theta' | theta_C | energies [4][B] | CFA partials [B * nblk * 2]
int mlmcpi_lattice_twolevel_workspace_bytes(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse,
                                            uint32_t B, size_t *bytes) {
  uint32_t rt, rx;
  if (int rc = check_twolevel(fine, coarse, &rt, &rx)) return rc;
  MLMCPI_REQUIRE(bytes && B > 0, "bad arguments");
  const size_t nf = (size_t)2 * fine->Mt * fine->Mx, nc = (size_t)2 * coarse->Mt * coarse->Mx;
  *bytes = align256_((size_t)B * nf * 8) + align256_((size_t)B * nc * 8) + align256_((size_t)4 * B * 8) +
           align256_((size_t)B * row_blocks(coarse->Mx, B) * 2 * 8);
  return MLMCPI_OK;
}

int mlmcpi_lattice_twolevel_draw(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse,
                                 const double *d_phi_coarse, double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0,
                                 uint32_t step, void *d_work, int32_t *d_accept, double *d_terms, void *stream) {
  return mlmcpi_lattice_twolevel_draw_cfa(fine, coarse, 0, d_phi_coarse, d_theta, B, seed, chain0, step, d_work, d_accept, d_terms,
                                          stream);
}

int mlmcpi_lattice_twolevel_draw_cfa(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse, int32_t cfa_kind,
                                     const double *d_phi_coarse, double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0,
                                     uint32_t step, void *d_work, int32_t *d_accept, double *d_terms, void *stream) {
  uint32_t rt, rx;
  if (int rc = check_twolevel(fine, coarse, &rt, &rx)) return rc;
  MLMCPI_REQUIRE(d_phi_coarse && d_theta && d_work && d_accept && B > 0, "bad arguments");
  MLMCPI_REQUIRE(cfa_kind == 0 || cfa_kind == 1, "unknown conditioned fine action %d", cfa_kind);
  MLMCPI_REQUIRE(cfa_kind == 0 || rt * rx == 4, "the Gaussian conditioned fine action needs a lattice coarsened in both directions");
  hipStream_t st = as_stream(stream);
  const size_t nf = (size_t)2 * fine->Mt * fine->Mx, nc = (size_t)2 * coarse->Mt * coarse->Mx;
  char *w = (char *)d_work;
  double *theta_prime = (double *)w;
  w += align256_((size_t)B * nf * 8);
  double *theta_c = (double *)w;
  w += align256_((size_t)B * nc * 8);
  double *en4 = (double *)w;
  w += align256_((size_t)4 * B * 8);
  double *cfa = (double *)w;
  const RngKey key = make_key(seed, chain0, step);
  const uint32_t nblk = row_blocks(coarse->Mx, B);
  if (rt * rx == 4 && cfa_kind == 1) {  // QuenchedSchwingerGaussianConditionedFineAction
    const dim3 grid(nblk, B), block(256);
    hipLaunchKernelGGL(schwinger_gauss_fill_kernel, grid, block, 0, st, coarse->Mt, coarse->Mx, fine->beta, (const double2 *)d_phi_coarse,
                       (double2 *)theta_prime, key);
    MLMCPI_LAUNCH_CHECK("schwinger_gauss_fill_kernel");
    hipLaunchKernelGGL(schwinger_gauss_cfa_kernel, grid, block, 0, st, fine->Mt, fine->Mx, fine->beta, (const double2 *)theta_prime, cfa, 0u);
    hipLaunchKernelGGL(schwinger_gauss_cfa_kernel, grid, block, 0, st, fine->Mt, fine->Mx, fine->beta, (const double2 *)d_theta, cfa, 1u);
    MLMCPI_LAUNCH_CHECK("schwinger_gauss_cfa_kernel");
  } else if (rt * rx == 4) {
    const BesselFill P = make_bessel_fill(fine->beta);
    const dim3 grid(nblk, B), block(256);
    hipLaunchKernelGGL(schwinger_both_fill_kernel, grid, block, 0, st, coarse->Mt, coarse->Mx, P, (const double2 *)d_phi_coarse,
                       (double2 *)theta_prime, key);
    MLMCPI_LAUNCH_CHECK("schwinger_both_fill_kernel");
    hipLaunchKernelGGL(schwinger_both_rows_kernel, grid, block, 0, st, fine->Mt, fine->Mx, fine->beta, (double2 *)theta_prime, key);
    MLMCPI_LAUNCH_CHECK("schwinger_both_rows_kernel");
    hipLaunchKernelGGL(schwinger_both_cfa_kernel, grid, block, 0, st, fine->Mt, fine->Mx, P, (const double2 *)theta_prime, cfa, 0u);
    hipLaunchKernelGGL(schwinger_both_cfa_kernel, grid, block, 0, st, fine->Mt, fine->Mx, P, (const double2 *)d_theta, cfa, 1u);
    MLMCPI_LAUNCH_CHECK("schwinger_both_cfa_kernel");
  } else {
    hipLaunchKernelGGL(schwinger_twolevel_propose_kernel, dim3(nblk, B), dim3(256), 0, st, coarse->Mt, coarse->Mx, rt,
                       fine->beta, (const double2 *)d_phi_coarse, (const double2 *)d_theta, (double2 *)theta_prime, cfa, key);
    MLMCPI_LAUNCH_CHECK("schwinger_twolevel_propose_kernel");
  }
  if (int rc = lattice_energy(fine, theta_prime, B, en4, st)) return rc;
  if (int rc = lattice_energy(fine, d_theta, B, en4 + B, st)) return rc;
  if (int rc = mlmcpi_lattice_copy_from_fine(fine, rt, rx, d_theta, theta_c, B, stream)) return rc;
  if (int rc = lattice_energy(coarse, theta_c, B, en4 + 2 * (size_t)B, st)) return rc;
  if (int rc = lattice_energy(coarse, d_phi_coarse, B, en4 + 3 * (size_t)B, st)) return rc;
  uint32_t nb = (uint32_t)((nf + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(lattice_twolevel_accept_kernel, dim3(nb, B), dim3(256), 0, st, (uint32_t)nf, d_theta,
                     (const double *)theta_prime, (const double *)en4, (const double *)cfa, nblk, B, d_accept, d_terms, key);
  MLMCPI_LAUNCH_CHECK("lattice_twolevel_accept_kernel");
  return MLMCPI_OK;
}

}  // extern "C"

// =================================================================================================
// Exact sampler of the Gaussian free field: GFFAction::draw / initialise_state (action/qft/gffaction.cc:121-123,
// 200-213).  The reference solves with a sparse Cholesky factor of the precision matrix Q = (4 + mu2) 1 - A built by
// Eigen (infeasible beyond ~64^2).  On the periodic lattice Q is diagonal in Fourier space,
//   lambda(k) = 4 + mu2 - 2 cos(2 pi k_t / Mt) - 2 cos(2 pi k_x / Mx),
// so a draw from N(0, Q^-1) is  phi(x) = Re sum_k w_k e^{+i k x} / sqrt(N lambda(k))  with w_k = n0 + i n1 complex
// white noise (E |w|^2 = 2; then E phi(x) phi(y) = (1/N) sum_k cos(k (x - y)) / lambda(k) = (Q^-1)_xy): one kernel
// fills the spectrum from Philox (site = mode index, purpose P_EXACT), hipFFT does the batched 2-D inverse
// transform in place, one kernel keeps the real part.  O(N log N) per chain at any lattice size.
// =================================================================================================
namespace mlmcpi {

__global__ void __launch_bounds__(256)
    gff_spectrum_kernel(uint32_t Mt, uint32_t Mx, double mu2, double2 *__restrict__ w_all, RngKey key0, uint32_t sub) {
  const uint32_t b = blockIdx.y, n = Mt * Mx;
  RngKey key = key0;
  key.chain += b;
  double2 *w = w_all + (size_t)b * n;
  const double inv_n = 1.0 / (double)n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) {
    const uint32_t kx = l / Mt, kt = l - kx * Mt;  // same layout as the field: mode (kt, kx) at kx * Mt + kt
    const double lambda = 4.0 + mu2 - 2.0 * cos(kTwoPi * kt / Mt) - 2.0 * cos(kTwoPi * kx / Mx);
    double n0, n1;
    rng_normals(key, l, P_EXACT, sub, n0, n1);
    const double s = sqrt(inv_n / lambda);
    w[l] = make_double2(s * n0, s * n1);
  }
}

__global__ void __launch_bounds__(256)
    gff_real_part_kernel(uint32_t n, const double2 *__restrict__ w_all, double *__restrict__ phi_all) {
  const uint32_t b = blockIdx.y;
  const double2 *w = w_all + (size_t)b * n;
  double *phi = phi_all + (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) phi[l] = w[l].x;
}

}  // namespace mlmcpi

extern "C" {

int mlmcpi_lattice_exact_workspace_bytes(const mlmcpi_lattice_action *act, uint32_t B, size_t *bytes) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(bytes && B > 0, "bad arguments");
  if (act->kind != MLMCPI_GFF) return fail(MLMCPI_ERR_UNSUPPORTED, "exact sampler only for the GFF action");
  *bytes = (size_t)B * act->Mt * act->Mx * sizeof(double2);
  return MLMCPI_OK;
}

// one batch of chains: spectrum -> inverse FFT (in place, d_work) -> real part
static int gff_exact_batch(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0,
                           uint32_t step, uint32_t sub, void *d_work, hipStream_t st) {
  const uint32_t n = act->Mt * act->Mx;
  // one cached plan per device and (Mt, Mx, B): plan creation costs milliseconds; a plan is bound to the device that was
  // current when it was made.  The lock is held across the enqueue: hipfftSetStream + Exec on a shared plan is not
  // re-entrant.
  struct PlanSlot { hipfftHandle plan = 0; uint32_t mt = 0, mx = 0, b = 0; };
  static std::mutex guard;
  static PlanSlot slots[64];
  int dev = 0;
  MLMCPI_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(MLMCPI_ERR_INVALID, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(guard);
  hipfftHandle &plan = slots[dev].plan;
  uint32_t &p_mt = slots[dev].mt, &p_mx = slots[dev].mx, &p_b = slots[dev].b;
  if (!plan || p_mt != act->Mt || p_mx != act->Mx || p_b != B) {
    if (plan) hipfftDestroy(plan);
    plan = 0;
    int dims[2] = {(int)act->Mx, (int)act->Mt};  // slowest index first
    if (hipfftPlanMany(&plan, 2, dims, nullptr, 1, (int)n, nullptr, 1, (int)n, HIPFFT_Z2Z, (int)B) != HIPFFT_SUCCESS) {
      plan = 0;
      return fail(MLMCPI_ERR_HIP, "hipfftPlanMany failed for %u x %u x %u", act->Mt, act->Mx, B);
    }
    p_mt = act->Mt; p_mx = act->Mx; p_b = B;
  }
  if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) return fail(MLMCPI_ERR_HIP, "hipfftSetStream failed");
  uint32_t nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(gff_spectrum_kernel, dim3(nb, B), dim3(256), 0, st, act->Mt, act->Mx, gff_mu2(*act), (double2 *)d_work,
                     make_key(seed, chain0, step), sub);
  MLMCPI_LAUNCH_CHECK("gff_spectrum_kernel");
  if (hipfftExecZ2Z(plan, (hipfftDoubleComplex *)d_work, (hipfftDoubleComplex *)d_work, HIPFFT_BACKWARD) != HIPFFT_SUCCESS)
    return fail(MLMCPI_ERR_HIP, "hipfftExecZ2Z failed");
  hipLaunchKernelGGL(gff_real_part_kernel, dim3(nb, B), dim3(256), 0, st, n, (const double2 *)d_work, d_phi);
  MLMCPI_LAUNCH_CHECK("gff_real_part_kernel");
  return MLMCPI_OK;
}

int mlmcpi_lattice_exact_draw(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0,
                              uint32_t step, void *d_work, void *stream) {
  if (int rc = check_lattice(act)) return rc;
  MLMCPI_REQUIRE(d_phi && d_work && B > 0, "bad arguments");
  if (act->kind != MLMCPI_GFF) return fail(MLMCPI_ERR_UNSUPPORTED, "exact sampler only for the GFF action");
  return gff_exact_batch(act, d_phi, B, seed, chain0, step, 0, d_work, as_stream(stream));
}

// GFFAction::initialise_state = draw (gffaction.cc:121-123): the exact sampler with its own Philox sub-stream, in
// batches of chains that keep the library scratch below 256 MiB
static int gff_initialise_exact(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0,
                                hipStream_t st) {
  const size_t per_chain = (size_t)act->Mt * act->Mx * sizeof(double2);
  uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)256 << 20) / per_chain);
  if (chunk > B) chunk = B;
  void *work = nullptr;
  if (int rc = scratch((size_t)chunk * per_chain, &work, st)) return rc;
  for (uint32_t b0 = 0; b0 < B; b0 += chunk) {
    const uint32_t nb = std::min(chunk, B - b0);
    if (int rc = gff_exact_batch(act, d_phi + (size_t)b0 * act->Mt * act->Mx, nb, seed, chain0 + b0, 0, 1, work, st)) return rc;
  }
  return MLMCPI_OK;
}

#ifdef MLMCPI_STAMPS
// instrumentation build only: the stamps of the last launch of schwinger_or_heat_kernel, 16 words per workgroup
int mlmcpi_debug_read_stamps(unsigned long long *h_out, uint32_t n_workgroups) {
  MLMCPI_HIP_TRY(hipDeviceSynchronize());
  MLMCPI_HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(mlmcpi::g_stamps), (size_t)n_workgroups * 16 * sizeof(unsigned long long)));
  return MLMCPI_OK;
}
#endif

}  // extern "C"
