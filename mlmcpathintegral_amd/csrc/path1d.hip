// path1d.hip -- 1-D path kernels (harmonic / quartic oscillator, topological rotor), batched over
// B chains laid out chain-major x[b*M + j]:
//   * Action::evaluate / Action::force                      (standalone, drop-in methods)
//   * QoIXsquared / QoISusceptibility
//   * fused HMC trajectory: state AND momenta stay in registers for the whole trajectory, one
//     workgroup per chain segment, halo of nt+1 sites recomputed redundantly, so HBM sees one read
//     and one write of the path per trajectory instead of 4 x 8 B per site per leapfrog step
//   * rotor overrelaxation / heat-bath sweeps (even / odd colouring) on LDS-resident segments
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace mlmcpi {

struct PathP {
  int kind;
  uint32_t M;
  double a, m0, mu2, lambda, x0;
  double c1;      // m0 / a
  double c2;      // 2 + a^2 mu2
  double c3;      // a lambda
  double inv_a2;  // 1 / a^2
  double T_final;
};

static PathP make_params(const mlmcpi_path_action &A) {
  PathP P;
  P.kind = A.kind;
  P.M = A.M;
  P.T_final = A.T_final;
  P.a = A.T_final / A.M;  // lattice/lattice1d.cc:9
  P.m0 = A.m0;
  P.mu2 = A.mu2;
  P.lambda = A.lambda;
  P.x0 = A.x0;
  P.c1 = A.m0 / P.a;
  P.c2 = 2. + P.a * P.a * A.mu2;
  P.c3 = P.a * A.lambda;
  P.inv_a2 = 1. / (P.a * P.a);
  return P;
}

// Site term of the action involving x_j and its left neighbour; S = energy_scale * sum_j term_j.
//   HO       harmonicoscillatoraction.cc:8-18     S = (a m0/2) sum [ (dx)^2/a^2 + mu2 x^2 ]
//   quartic  quarticoscillatoraction.cc:7-27      S = (a/2) sum [ m0((dx)^2/a^2 + mu2 x^2) + (lambda/2)(x-x0)^4 ]
//   rotor    rotoraction.cc:9-18                  S = (m0/a) sum [ 1 - cos(dx) ]
template <int KIND>
__device__ __forceinline__ double site_energy(const PathP &P, double x, double xl) {
  const double d = x - xl;
  if (KIND == MLMCPI_HARMONIC) return P.inv_a2 * d * d + P.mu2 * x * x;
  if (KIND == MLMCPI_QUARTIC) {
    const double sh = x - P.x0, sh2 = sh * sh;
    return P.m0 * (P.inv_a2 * d * d + P.mu2 * (x * x)) + 0.5 * P.lambda * sh2 * sh2;
  }
  return 1. - cos(d);
}

__host__ __device__ inline double energy_scale(const PathP &P) {
  if (P.kind == MLMCPI_HARMONIC) return 0.5 * P.a * P.m0;
  if (P.kind == MLMCPI_QUARTIC) return 0.5 * P.a;
  return P.m0 / P.a;
}

// Force on site j.  HO harmonicoscillatoraction.cc:21-35, quartic quarticoscillatoraction.cc:30-53,
// rotor rotoraction.cc:59-79.
template <int KIND>
__device__ __forceinline__ double site_force(const PathP &P, double xl, double x, double xr) {
  if (KIND == MLMCPI_ROTOR) return P.c1 * (sin(x - xl) + sin(x - xr));
  double f = P.c1 * (P.c2 * x - xl - xr);
  if (KIND == MLMCPI_QUARTIC) {
    const double sh = x - P.x0;
    f += P.c3 * sh * sh * sh;
  }
  return f;
}

// ---- standalone evaluate / force / QoI ------------------------------------------------------------
enum ReduceOp { R_ENERGY = 0, R_XSQUARED = 1, R_WINDING = 2 };

// grid (nsplit, B); partial[b*nsplit + s] = sum over the split's sites of the site term
// `stride` > 1 evaluates on every stride-th entry of a longer path (the coarse points of a fine path,
// action/qm/qmaction.cc:16-24, without materialising the copy)
template <int KIND, int OP>
__global__ void __launch_bounds__(256) path_reduce_kernel(PathP P, const double *__restrict__ x,
                                                          double *__restrict__ partial, uint32_t stride, double scale,
                                                          double *__restrict__ out) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.y, M = P.M;
  const double *xb = x + (size_t)b * M * stride;
  const uint32_t per = (M + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(M, lo + per);
  double acc[1] = {0.0};
  for (uint32_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    const double xj = xb[(size_t)j * stride], xl = xb[(size_t)(j == 0 ? M - 1 : j - 1) * stride];
    if (OP == R_ENERGY) acc[0] += site_energy<KIND>(P, xj, xl);
    if (OP == R_XSQUARED) acc[0] += xj * xj;
    if (OP == R_WINDING) acc[0] += mod_2pi(xj - xl);
  }
  block_sum<1>(acc, red);
  if (threadIdx.x != 0) return;
  if (gridDim.x == 1)  // one workgroup per chain: the sum is complete, finish here (what path_finish_kernel does)
    out[b] = (OP == R_WINDING) ? (1. / (4. * kPi * kPi)) * (acc[0] * acc[0]) * scale : scale * acc[0];
  else
    partial[(size_t)b * gridDim.x + blockIdx.x] = acc[0];
}

// out[b] = finish(sum_s partial[b][s]); one thread per chain, fixed summation order
__global__ void path_finish_kernel(const double *__restrict__ partial, uint32_t nsplit, uint32_t B, int op,
                                   double scale, double *__restrict__ out, double *__restrict__ acc = nullptr) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (uint32_t k = 0; k < nsplit; ++k) s += partial[(size_t)b * nsplit + k];
  // R_WINDING: chi = Q^2 / (4 pi^2 T)  (qoisusceptibility.cc:20-22); others: scale * sum
  const double v = (op == R_WINDING) ? (1. / (4. * kPi * kPi)) * (s * s) * scale : scale * s;
  out[b] = v;
  if (acc) {  // stats->record_sample of the value as well (the recurrence of stats_accumulate_kernel)
    double *a = acc + 5 * (size_t)b;
    a[0] += 1.0;
    a[1] += v;
    a[2] += v * v;
    a[3] += v * v * v;
    a[4] += v * v * v * v;
  }
}

template <int KIND>
__global__ void __launch_bounds__(256) path_force_kernel(PathP P, const double *__restrict__ x,
                                                         double *__restrict__ f) {
  const uint32_t b = blockIdx.y, M = P.M;
  const double *xb = x + (size_t)b * M;
  double *fb = f + (size_t)b * M;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
    const double xl = xb[j == 0 ? M - 1 : j - 1], xr = xb[j + 1 == M ? 0 : j + 1];
    fb[j] = site_force<KIND>(P, xl, xb[j], xr);
  }
}

__global__ void __launch_bounds__(256) path_init_kernel(int kind, uint32_t M, RngKey key0, double *__restrict__ x) {
  const uint32_t b = blockIdx.y;
  RngKey key = key0;
  key.chain += b;
  double *xb = x + (size_t)b * M;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
    if (kind == MLMCPI_ROTOR) {
      double u, v;
      rng_uniforms(key, j, P_INIT, 0, u, v);
      xb[j] = -kPi + 2.0 * kPi * u;
    } else {
      xb[j] = 0.0;
    }
  }
}

// ---- fused HMC trajectory ---------------------------------------------------------------------------
// Grid (nseg, B), NT threads, R consecutive sites per thread: buffer site k = t*R + r maps to global
// site (g0 + k) mod M.  halo == 0 means the buffer IS the periodic path (NT*R == M).  Otherwise the
// first / last `halo` = nt+1 buffer sites are recomputed copies of the neighbouring segments; the
// error front entering from the clamped buffer ends advances one site per leapfrog step and never
// reaches an owned site.  Only boundary values move through LDS (2 doubles per thread per step,
// double buffered -> one barrier per step).
//
// sampler/hmcsampler.cc:22-57: p ~ N(0,1); T0; nt+1 force evaluations with half steps for p at
// both ends and no position update after the last; T1; S(x_trial), S(x_cur).
template <int KIND, int R>
__global__ void __launch_bounds__(R >= 8 ? 512 : 1024)
    hmc_trajectory_kernel(PathP P, const double *__restrict__ x_cur, double *__restrict__ x_trial,
                          double *__restrict__ partials, const int32_t *__restrict__ done, uint32_t owned_len,
                          uint32_t halo, uint32_t nt, double dt, RngKey key0) {
  extern __shared__ double lds[];  // [2][2][NT] boundary exchange | 4*NT/64 reduction scratch | [R][NT] staging
  const uint32_t b = blockIdx.y, seg = blockIdx.x, t = threadIdx.x, NT = blockDim.x, M = P.M;
  if (done && done[b]) return;  // reference: repetitions after an acceptance are not run (hmcsampler.cc:10-12); null: none yet
  const bool periodic = (halo == 0);
  const uint32_t o0 = seg * owned_len;
  const uint32_t olen = min(owned_len, M - o0);
  const uint32_t g0 = (uint32_t)(((uint64_t)o0 + M - (halo % M)) % M);
  const uint32_t kbase = t * R;
  const double *xb = x_cur + (size_t)b * M;
  RngKey key = key0;
  key.chain += b;

  // Momenta are generated in a rolled loop through LDS: unrolled, the R Box-Muller chains get
  // interleaved and their temporaries push the 2R doubles of state out of the register file.
  double *ex_first = lds, *ex_last = lds + 2 * NT;  // [2][NT] each
  double *stage = lds + 4 * NT + 4 * (NT / kWave);  // [R][NT]
  {
    uint32_t g = (uint32_t)(((uint64_t)g0 + kbase) % M);
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      stage[r * NT + t] = rng_normal0(key, g, P_MOMENTUM, 0);
      g = (g + 1 == M) ? 0 : g + 1;
    }
  }
  double x[R], p[R];
  uint32_t g = (uint32_t)(((uint64_t)g0 + kbase) % M);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    x[r] = xb[g];
    p[r] = stage[r * NT + t];  // written by this thread: no barrier needed
    g = (g + 1 == M) ? 0 : g + 1;
  }

  int buf = 0;
  double xl, xr;
  auto exchange = [&]() {
    ex_first[buf * NT + t] = x[0];
    ex_last[buf * NT + t] = x[R - 1];
    __syncthreads();
    if (t == 0)
      xl = periodic ? ex_last[buf * NT + NT - 1] : x[0];
    else
      xl = ex_last[buf * NT + t - 1];
    if (t == NT - 1)
      xr = periodic ? ex_first[buf * NT] : x[R - 1];
    else
      xr = ex_first[buf * NT + t + 1];
    buf ^= 1;
  };

  // owned mask of buffer site k: halo <= k < halo + olen
  auto owned = [&](int r) { return (kbase + r - halo) < olen; };

  double sums[4] = {0.0, 0.0, 0.0, 0.0};  // S_cur, T0, S_trial, T1 (raw site sums)
  exchange();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (owned(r)) {
      sums[0] += site_energy<KIND>(P, x[r], r == 0 ? xl : x[r - 1]);
      sums[1] += p[r] * p[r];
    }
    if (KIND == MLMCPI_ROTOR) __builtin_amdgcn_sched_barrier(0);
  }

  for (uint32_t k = 0; k <= nt; ++k) {
    const double dtp = (k == 0 || k == nt) ? 0.5 * dt : dt;
    const double dtx = (k == nt) ? 0.0 : dt;
    if (KIND == MLMCPI_ROTOR) {
      // one sine per link: d_r = sin(x_r - x_{r-1}); F_r = c1 (d_r - d_{r+1}), identical to
      // c1 (sin(x-x_m) + sin(x-x_p)) because sin is odd
      double dprev = sin_reduced(x[0] - xl);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double dnext = sin_reduced((r == R - 1 ? xr : x[r + 1]) - x[r]);
        p[r] -= dtp * (P.c1 * (dprev - dnext));
        dprev = dnext;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      double left = xl;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double right = (r == R - 1) ? xr : x[r + 1];
        const double f = site_force<KIND>(P, left, x[r], right);
        left = x[r];
        p[r] -= dtp * f;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] += dtx * p[r];
    if (k < nt) exchange();  // positions do not change in the last step: xl stays valid
  }

  double *xt = x_trial + (size_t)b * M;
  g = (uint32_t)(((uint64_t)g0 + kbase) % M);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (owned(r)) {
      sums[2] += site_energy<KIND>(P, x[r], r == 0 ? xl : x[r - 1]);
      sums[3] += p[r] * p[r];
      xt[g] = x[r];
    }
    g = (g + 1 == M) ? 0 : g + 1;
    if (KIND == MLMCPI_ROTOR) __builtin_amdgcn_sched_barrier(0);
  }
  block_sum<4>(sums, lds + 4 * NT);
  if (t == 0) {
    double *out = partials + ((size_t)b * gridDim.x + seg) * 4;
    out[0] = sums[0]; out[1] = sums[1]; out[2] = sums[2]; out[3] = sums[3];
  }
}

// Whole chains on the device: n_draws x n_rep trajectories of a periodic, register-resident path
// (one workgroup per chain) in ONE launch, with the Metropolis test, the copy-on-accept and the QoI
// of every draw done in-kernel.  Same arithmetic, same Philox counters (trajectory index
// traj0 + d*n_rep + r) as n_draws calls of mlmcpi_path_hmc_draw followed by the QoI kernel, so the
// two forms agree to rounding; this one removes ~2 launches and a host round trip per draw, which is
// what dominates for short paths (BASELINE config 1: M_lat = 128; the coarse levels of config 5).
// qoi_kind: 0 none, 1 <x^2> (qoixsquared.cc:7-20), 2 susceptibility (qoisusceptibility.cc:8-23).
template <int KIND, int R>
__global__ void __launch_bounds__(R >= 8 ? 512 : 1024)
    hmc_chain_kernel(PathP P, double *__restrict__ x_state, double *__restrict__ q_out,
                     int32_t *__restrict__ acc_count, double *__restrict__ energies, uint32_t nt, double dt,
                     uint32_t n_rep, uint32_t n_draws, int qoi_kind, RngKey key0) {
  extern __shared__ double lds[];  // [2][2][NT] exchange | 4*NT/64 scratch | [R][NT] staging | flag
  const uint32_t b = blockIdx.x, t = threadIdx.x, NT = blockDim.x, M = P.M;
  const uint32_t kbase = t * R;
  double *xb = x_state + (size_t)b * M;
  RngKey key = key0;
  key.chain += b;
  double *ex_first = lds, *ex_last = lds + 2 * NT;
  double *scratch = lds + 4 * NT;
  double *stage = scratch + 4 * (NT / kWave);
  double *flag = stage + (size_t)R * NT;
  const double escale = energy_scale(P);

  double xc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) xc[r] = xb[kbase + r];
  int buf = 0;
  // A chain that one wave holds (M = 64 R: BASELINE config 1, M_lat = 128) exchanges its boundary values by rotating the
  // wave one lane with DPP (wave_ror / wave_rol wrap around, which is the periodic boundary): four v_mov_b32_dpp instead
  // of two LDS writes, a barrier and two LDS reads per leapfrog step -- the step of such a chain is nothing but this latency.
  const bool one_wave = NT == kWave;
  auto exchange = [&](const double (&v)[R], double &xl, double &xr) {
    if (one_wave) {
      xl = wave_rotate_up(v[R - 1]);  // lane t gets lane t - 1 (lane 0: lane 63)
      xr = wave_rotate_down(v[0]);    // lane t gets lane t + 1 (lane 63: lane 0)
      return;
    }
    ex_first[buf * NT + t] = v[0];
    ex_last[buf * NT + t] = v[R - 1];
    __syncthreads();
    xl = ex_last[buf * NT + (t == 0 ? NT - 1 : t - 1)];
    xr = ex_first[buf * NT + (t == NT - 1 ? 0 : t + 1)];
    buf ^= 1;
  };
  int32_t n_acc = 0;
  for (uint32_t d = 0; d < n_draws; ++d) {
    bool accepted = false;
    for (uint32_t rep = 0; rep < n_rep && !accepted; ++rep) {
      key.step = key0.step + d * n_rep + rep;
#pragma unroll 1
      for (int r = 0; r < R; ++r) stage[r * NT + t] = rng_normal0(key, kbase + r, P_MOMENTUM, 0);
      double x[R], p[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        x[r] = xc[r];
        p[r] = stage[r * NT + t];
      }
      double sums[4] = {0.0, 0.0, 0.0, 0.0};
      double xl, xr;
      exchange(x, xl, xr);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        sums[0] += site_energy<KIND>(P, x[r], r == 0 ? xl : x[r - 1]);
        sums[1] += p[r] * p[r];
        if (KIND == MLMCPI_ROTOR) __builtin_amdgcn_sched_barrier(0);
      }
      for (uint32_t k = 0; k <= nt; ++k) {
        const double dtp = (k == 0 || k == nt) ? 0.5 * dt : dt;
        const double dtx = (k == nt) ? 0.0 : dt;
        if (KIND == MLMCPI_ROTOR) {
          double dprev = sin_reduced(x[0] - xl);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double dnext = sin_reduced((r == R - 1 ? xr : x[r + 1]) - x[r]);
            p[r] -= dtp * (P.c1 * (dprev - dnext));
            dprev = dnext;
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          double left = xl;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double right = (r == R - 1) ? xr : x[r + 1];
            const double f = site_force<KIND>(P, left, x[r], right);
            left = x[r];
            p[r] -= dtp * f;
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] += dtx * p[r];
        if (k < nt) exchange(x, xl, xr);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        sums[2] += site_energy<KIND>(P, x[r], r == 0 ? xl : x[r - 1]);
        sums[3] += p[r] * p[r];
        if (KIND == MLMCPI_ROTOR) __builtin_amdgcn_sched_barrier(0);
      }
      block_sum<4>(sums, scratch);
      if (t == 0) {
        const double S0 = escale * sums[0], T0 = 0.5 * sums[1], S1 = escale * sums[2], T1 = 0.5 * sums[3];
        const double dH = (S1 - S0) + (T1 - T0);
        bool acc;
        if (dH < 0.0) {
          acc = true;
        } else {
          double u, v;
          rng_uniforms(key, 0, P_ACCEPT, 0, u, v);
          acc = u < exp(-dH);
        }
        flag[0] = acc ? 1.0 : 0.0;
        if (energies) {
          energies[4 * b + 0] = S0; energies[4 * b + 1] = T0; energies[4 * b + 2] = S1; energies[4 * b + 3] = T1;
        }
      }
      __syncthreads();
      accepted = flag[0] != 0.0;
      __syncthreads();  // flag is rewritten by the next repetition
      if (accepted) {
#pragma unroll
        for (int r = 0; r < R; ++r) xc[r] = x[r];
      }
    }
    n_acc += accepted ? 1 : 0;
    if (qoi_kind) {
      double q[1] = {0.0};
      if (qoi_kind == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) q[0] += xc[r] * xc[r];
      } else {
        double xl, xr;
        exchange(xc, xl, xr);
#pragma unroll
        for (int r = 0; r < R; ++r) q[0] += mod_2pi(xc[r] - (r == 0 ? xl : xc[r - 1]));
      }
      block_sum<1>(q, scratch);
      if (t == 0)
        q_out[(size_t)b * n_draws + d] =
            (qoi_kind == 1) ? (1.0 / M) * q[0] : (1. / (4. * kPi * kPi)) * (q[0] * q[0]) * (1.0 / P.T_final);
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) xb[kbase + r] = xc[r];
  if (t == 0 && acc_count) acc_count[b] = n_acc;
}

// Global Metropolis test + copy of accepted trial states.  Grid (nblk, B).  Every workgroup of a
// chain recomputes the (cheap) decision from the segment partials in the same order, so no
// inter-workgroup hand-off is needed.  sampler/hmcsampler.cc:50-67.
__global__ void __launch_bounds__(256)
    hmc_accept_kernel(uint32_t M, double escale, double *__restrict__ x_cur, const double *__restrict__ x_trial,
                      const double *__restrict__ partials, uint32_t nseg, const int32_t *__restrict__ done_in,
                      int32_t *__restrict__ done_out, double *__restrict__ energies, RngKey key0) {
  const uint32_t b = blockIdx.y;
  if (done_in && done_in[b]) {  // (null: first repetition, no chain has accepted yet)
    if (blockIdx.x == 0 && threadIdx.x == 0) done_out[b] = 1;
    return;
  }
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint32_t k = 0; k < nseg; ++k) {
    const double *q = partials + ((size_t)b * nseg + k) * 4;
    s[0] += q[0]; s[1] += q[1]; s[2] += q[2]; s[3] += q[3];
  }
  const double S0 = escale * s[0], T0 = 0.5 * s[1], S1 = escale * s[2], T1 = 0.5 * s[3];
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
  double *dst = x_cur + (size_t)b * M;
  const double *src = x_trial + (size_t)b * M;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) dst[j] = src[j];
}

__global__ void add_flags_kernel(int32_t *__restrict__ acc, const int32_t *__restrict__ flags, uint32_t B) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) acc[b] += flags[b];
}

// QMAction::copy_from_fine / copy_from_coarse (action/qm/qmaction.cc:7-24): even sites of the fine path
__global__ void __launch_bounds__(256) path_transfer_kernel(uint32_t Mc, double *__restrict__ fine_all,
                                                            double *__restrict__ coarse_all, int to_coarse) {
  const uint32_t b = blockIdx.y;
  double *fine = fine_all + (size_t)b * 2 * Mc, *coarse = coarse_all + (size_t)b * Mc;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < Mc; j += gridDim.x * blockDim.x) {
    if (to_coarse) coarse[j] = fine[2 * j]; else fine[2 * j] = coarse[j];
  }
}

// ---- rotor sweeps -----------------------------------------------------------------------------------
// Grid (nseg, B), 256 threads.  The segment plus a halo of 2 sites per fused sweep lives in LDS;
// every site whose two neighbours are inside the buffer is updated, so stale values creep inwards
// by at most two sites per sweep and never reach the owned range.  in != out (halo reads race with
// the neighbours' writes otherwise).  kinds bit s = 1 -> sweep s is a heat-bath sweep.
// rotoraction.cc:20-56, rotoraction.hh:195-213.
// HEAT = false: overrelaxation-only instantiation (no sampler code, few registers).  STEP: heat-bath draws from the step
// envelope (2 m0 / a <= kVsKappaMax, device_common.hpp) instead of the wrapped-Cauchy one; pool_cap then counts VsPool entries.
#ifndef MLMCPI_ROTOR_LEAN
#define MLMCPI_ROTOR_LEAN 2
#endif
#ifndef MLMCPI_ROTOR_WAVES
#define MLMCPI_ROTOR_WAVES 1
#endif
template <bool HEAT, bool STEP = false>
__global__ void __launch_bounds__(256, HEAT && STEP ? MLMCPI_ROTOR_WAVES : 1)
    rotor_sweep_kernel(PathP P, const double *__restrict__ in, double *__restrict__ out, uint32_t owned_len,
                       uint32_t nsweeps, uint32_t kinds, RngKey key0, uint32_t pool_cap, const uint32_t *__restrict__ vs_table,
                       double *__restrict__ winding_partial = nullptr, uint32_t n_closed = 0) {
  extern __shared__ double lds_all[];
  __shared__ double qoi_red[4];
  // winding_partial != NULL: the segment's share of sum_j mod_2pi(x_j - x_{j-1}) (qoi/qm/qoisusceptibility.cc:8-23) of the
  // NEW state goes out with it; the left neighbour of the first owned site has to be exact for that: one more pair of halo sites
  const uint32_t b = blockIdx.y, seg = blockIdx.x, M = P.M, halo = 2 * nsweeps + (winding_partial ? 2u : 0u);
  // the sampler's tables and the list of open cells at the START of the LDS (table look-ups are then instruction offsets, not
  // additions of a wave-uniform base at half rate: r04, -4 % on the sweeps), the segment image behind them
  HbPool pool = HbPool::carve(lds_all, HEAT && !STEP ? pool_cap : 0u);
  VsPool<uint32_t> vpool = VsPool<uint32_t>::carve(lds_all, HEAT && STEP ? pool_cap : 0u, STEP ? vs_table : nullptr);
  double *const buf = lds_all + (HEAT ? (STEP ? VsPool<uint32_t>::bytes(pool_cap) : HbPool::bytes(pool_cap)) / sizeof(double) : 0);
  const uint32_t o0 = seg * owned_len, olen = min(owned_len, M - o0);
  const uint32_t L = olen + 2 * halo;
  const uint32_t g0 = (uint32_t)(((uint64_t)o0 + M - (halo % M)) % M);
  const double *xin = in + (size_t)b * M;
  RngKey key = key0;
  key.chain += b;
  for (uint32_t k = threadIdx.x; k < L; k += blockDim.x) {
    uint32_t g = g0 + k;  // g0 < M and M < 2^31 for every supported lattice: no overflow; no 64-bit modulo per element
    while (g >= M) g -= M;
    buf[k] = xin[g];
  }
  __syncthreads();
  const double sig_scale = 2.0 * P.m0 / P.a;  // W'' = (2 m0 / a) |cos((x+ - x-)/2)|
  // The first n_closed sweeps of the launch -- overrelaxation sweeps -- in closed form.  The update x_j <- x_{j-1} + x_{j+1} - x_j
  // (rotoraction.cc:40-56) adds d_j - d_{j-1} to x_j, d_j = x_{j+1} - x_j, and leaves the two differences exchanged; in
  // even / odd order a sweep moves the difference at an even index two down and the one at an odd index two up, whatever
  // the path is, so K sweeps add to the pair of sites (j, j + 1), j = 2 p even,
  //     x_j     += X - S,     S = sum_{s<K} d(j - 1 - 2 s) = sum_s do[p - 1 - s],    X  = sum_{s<K} d(j + 2 s) = sum_s de[p + s],
  //     x_{j+1} += X' - S,                                                           X' = X - de[p] + de[p + K]
  // with the differences of the path the launch started from, split by parity (de[i] = d(2 i), do[i] = d(2 i + 1): every
  // sum is a run of consecutive LDS words, consecutive lanes read consecutive words).  The same map as K sweeps to the
  // rounding of 2 K additions (the 2-D counterpart: lattice2d.hip, schwinger_perm_kernel); exact where the sweeps are
  // (buffer sites [2 K, L - 2 K)), the edge sites keep their values as they do under the sweeps' creeping halo.
  if (n_closed) {
    const uint32_t H2 = L / 2, K = n_closed;   // L is even (owned lengths, halos and M are) and at most 2048: <= 4 pairs per thread
    double xa[4], xb[4], dev[4], dov[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint32_t p = threadIdx.x + 256 * m;
      if (p < H2) {
        xa[m] = buf[2 * p];
        xb[m] = buf[2 * p + 1];
        dev[m] = xb[m] - xa[m];
        dov[m] = (2 * p + 2 < L ? buf[2 * p + 2] : xb[m]) - xb[m];
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint32_t p = threadIdx.x + 256 * m;
      if (p < H2) {
        buf[p] = dev[m];
        buf[H2 + p] = dov[m];
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint32_t p = threadIdx.x + 256 * m;
      if (p >= K && p + K < H2) {
        double S = 0.0, X = 0.0;
        const double *od = buf + H2 + p - 1, *ev = buf + p;
        for (uint32_t q = 0; q < K; ++q) {
          S += od[-(int)q];
          X += ev[q];
        }
        const double X2 = (X - ev[0]) + ev[K];
        xa[m] = mod_2pi_fast(xa[m] + (X - S));
        xb[m] = mod_2pi_fast(xb[m] + (X2 - S));
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint32_t p = threadIdx.x + 256 * m;
      if (p < H2) {
        buf[2 * p] = xa[m];
        buf[2 * p + 1] = xb[m];
      }
    }
    __syncthreads();
  }
  for (uint32_t s = n_closed; s < nsweeps; ++s) {
    const bool heat = HEAT && ((kinds >> s) & 1u);
    RngKey skey = key;
    skey.step += s;
    for (uint32_t colour = 0; colour < 2; ++colour) {
      // buffer parity == global parity (g0 is even because o0, halo and M are); sites k = k0, k0 + 2, ... < L - 1
      // with k0 = 2 for colour 0 and 1 for colour 1.  getWminimum (rotoraction.hh:206-213) in closed form:
      // atan2(sin x+ + sin x-, cos x+ + cos x-) = (x+ + x-)/2 (+ pi when cos((x+ - x-)/2) < 0), so that
      //   overrelaxation  mod_2pi(2 x0 - x) = mod_2pi(x+ + x- - x)                 (no transcendental at all)
      //   heat bath       mod_2pi(x0 + ExpSin2(2 W'')),  kappa = W'' = (2 m0/a) |cos((x+ - x-)/2)|  (one cosine)
      const uint32_t k0 = 2 - colour;
      const uint32_t count = (L > k0 + 1) ? (L - 1 - k0 + 1) / 2 : 0;
      if (!heat) {
        for (uint32_t idx = threadIdx.x; idx < count; idx += blockDim.x) {
          const uint32_t k = k0 + 2 * idx;
          buf[k] = mod_2pi_fast(buf[k - 1] + buf[k + 1] - buf[k]);
        }
      } else if (HEAT && STEP) {
        auto global_site = [&](uint32_t k) {
          uint32_t g = g0 + k;
          while (g >= M) g -= M;
          return g;
        };
        heatbath_cells_step<256, 4, uint32_t, MLMCPI_ROTOR_LEAN>(
            count, skey, vpool, [&](uint32_t idx) { return k0 + 2 * idx; },
            [&](uint32_t k, VsCell &cell) {
              vs_cell(sig_scale, buf[k + 1], buf[k - 1], cell);
              cell.site = global_site(k);
            },
            [&](uint32_t k) { return vs_kappa_exact(sig_scale, buf[k + 1], buf[k - 1]); },
            [&](uint32_t k, double angle) { buf[k] = angle; });
      } else if (HEAT) {
        heatbath_cells<256, 4, true>(
            count, skey, pool,
            [&](uint32_t idx, double &tau, double &centre, uint32_t &site, uint32_t &off) {
              const uint32_t k = k0 + 2 * idx;
              const double xm = buf[k - 1], xp = buf[k + 1];
              // |x+ - x-| / 2 <= pi: cos(d) = cos(pi u), u = |x+ - x-| / (2 pi) in [0, 1] (no libm range reduction)
              const double c = cospi_unit(fmin(fabs(xp - xm) * (0.5 / kPi), 1.0));
              tau = sig_scale * fabs(c);
              centre = 0.5 * (xp + xm) + (c < 0.0 ? kPi : 0.0);
              uint32_t g = g0 + k;
              while (g >= M) g -= M;
              site = g;
              off = k;
            },
            [&](uint32_t off, double angle) { buf[off] = angle; });
      }
      __syncthreads();
    }
  }
  double *xout = out + (size_t)b * M;
  double acc[1] = {0.0};
  for (uint32_t k = threadIdx.x; k < olen; k += blockDim.x) {
    xout[o0 + k] = buf[halo + k];
    if (winding_partial) acc[0] += mod_2pi(buf[halo + k] - buf[halo + k - 1]);
  }
  if (winding_partial) {
    block_sum<1>(acc, qoi_red);
    if (threadIdx.x == 0) winding_partial[(size_t)b * gridDim.x + seg] = acc[0];
  }
}

// Site-at-a-time rotor updates (rotoraction.cc:20-56 through Action::heatbath_update / overrelaxation_update,
// action/action.hh:73-96): one thread per chain walks the site list in order, on the state in global memory; arithmetic
// and random numbers of rotor_sweep_kernel.
__global__ void __launch_bounds__(64)
    rotor_site_update_kernel(PathP P, double *__restrict__ x_all, uint32_t B, const uint32_t *__restrict__ sites, uint32_t n,
                             uint32_t single, int heat, RngKey key0, const uint32_t *__restrict__ vs_table) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  RngKey key = key0;
  key.chain += b;
  double *x = x_all + (size_t)b * P.M;
  const double sig_scale = 2.0 * P.m0 / P.a;
  const bool step = sig_scale <= kVsKappaMax;
  const VsTable tab = VsTable::in_global(vs_table);
  for (uint32_t q = 0; q < n; ++q) {
    const uint32_t l = sites ? sites[q] : single;
    const double xm = x[l == 0 ? P.M - 1 : l - 1], xp = x[l + 1 == P.M ? 0 : l + 1];
    if (!heat) {
      x[l] = mod_2pi_fast(xm + xp - x[l]);
    } else if (step) {
      x[l] = vs_draw(key, l, sig_scale, xp, xm, tab);
    } else {
      const double c = cospi_unit(fmin(fabs(xp - xm) * (0.5 / kPi), 1.0));
      const double centre = 0.5 * (xp + xm) + (c < 0.0 ? kPi : 0.0);
      x[l] = mod_2pi_fast(vonmises_draw(key, l, sig_scale * fabs(c)) + centre);
    }
  }
}

// ---- two-level Metropolis step (montecarlo/twolevelmetropolisstep.cc:35-89) ----------------------------------
// Conditioned-action quantities of the Gaussian fill-in (action/qm/gaussianconditionedfineaction.cc:7-43):
// HO  harmonicoscillatoraction.hh:163-189: W'' = 2 m0/a + a m0 mu2, x0 = (x- + x+) / (2 + a^2 mu2)
// quartic quarticoscillatoraction.hh:160-194: W'' = (2/a + a mu2) m0 + 3 lambda a (xbar - x0)^2, x0 by 4 fixed-point steps
template <int KIND>
__device__ __forceinline__ void w_conditioned(const PathP &P, double x_m, double x_p, double &w_min, double &w_curv) {
  if (KIND == MLMCPI_ROTOR) {  // rotoraction.hh:195-213
    double sm, cm, sp, cp;
    sincos(x_m, &sm, &cm);
    sincos(x_p, &sp, &cp);
    w_min = atan2(sp + sm, cp + cm);
    w_curv = 2.0 * P.m0 / P.a * fabs(cos(0.5 * (x_p - x_m)));
  } else if (KIND == MLMCPI_HARMONIC) {
    w_curv = (2. / P.a + P.a * P.mu2) * P.m0;
    w_min = (0.5 / (1. + 0.5 * P.a * P.a * P.mu2)) * (x_m + x_p);
  } else {
    const double xbar = 0.5 * (x_m + x_p);
    const double rho = 1. / (1. + 0.5 * P.a * P.a * P.mu2);
    double x = xbar;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double sh = x - P.x0;
      x = rho * (xbar - 0.5 * P.a * P.a * P.lambda / P.m0 * sh * sh * sh);
    }
    w_min = x;
    w_curv = (2. / P.a + P.a * P.mu2) * P.m0 + 3. * P.lambda * P.a * (xbar - P.x0) * (xbar - P.x0);
  }
}

// TwoLevelMetropolisStep::draw (montecarlo/twolevelmetropolisstep.cc:35-89) for one chain per workgroup, in ONE launch (r02:
// a propose kernel, four reductions of two launches each and an accept kernel -- ten launches for an O(M) streaming job,
// 27 % of the multilevel kernel time).  The thread of coarse site j builds theta'[2j] = x_c[j] and the fill-in
// theta'[2j+1] (Gaussian around Wminimum with Wcurvature, gaussianconditionedfineaction.cc:7-43; ExpSin2 for the rotor,
// rotorconditionedfineaction.cc:7-43; Philox normals / von Mises draws of site 2j+1) and adds up, for the sites 2j+1 and
// 2j+2 of the fine paths and for coarse site j+1, the six sums of the step:
//     S_f(theta'), S_f(theta), S_c(theta_C), S_c(x_c), S_cfa(theta'), S_cfa(theta).
// The workgroup reduces them in a fixed order, takes the decision (exp(-dS) against the chain's P_ACCEPT2 uniform) and,
// if accepted, copies theta' over theta -- every thread the entries it wrote itself.
template <int KIND>
__global__ void __launch_bounds__(KIND == MLMCPI_ROTOR ? 512 : 1024)  // the rotor's libm calls want more than 128 registers
    twolevel_fused_kernel(PathP Pf, PathP Pc, const double *__restrict__ x_coarse, double *__restrict__ theta,
                          double *__restrict__ theta_prime, int32_t *__restrict__ accept, double *__restrict__ terms, RngKey key0,
                          const int32_t *__restrict__ mask) {
  __shared__ double red[6 * 16];
  __shared__ int decision;
  const uint32_t b = blockIdx.x, M = Pf.M, Mc = M / 2;
  if (mask && mask[b] == 0) {  // hierarchicalsampler.cc:62-76: a chain rejected further down does not move on this level
    if (threadIdx.x == 0) {
      accept[b] = 0;
      if (terms) terms[3 * b + 0] = terms[3 * b + 1] = terms[3 * b + 2] = 0.0;
    }
    return;
  }
  const double *xc = x_coarse + (size_t)b * Mc;
  double *th = theta + (size_t)b * M, *tp = theta_prime + (size_t)b * M;
  RngKey key = key0;
  key.chain += b;
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (uint32_t j = threadIdx.x; j < Mc; j += blockDim.x) {
    const uint32_t jn = (j + 1 == Mc) ? 0 : j + 1;
    const double x_m = xc[j], x_p = xc[jn];
    double w_min, w_curv;
    w_conditioned<KIND>(Pf, x_m, x_p, w_min, w_curv);
    double fill;
    if (KIND == MLMCPI_ROTOR) {
      const double sigma = 2. * w_curv;
      fill = mod_2pi(w_min + vonmises_draw(key, 2 * j + 1, 0.5 * sigma, kVmFillin));
      const double sh = sin(0.5 * (fill - w_min));
      acc[4] += sigma * sh * sh + log(two_pi_i0_scaled(0.5 * sigma));
    } else {
      const double sigma = 1. / sqrt(w_curv);
      fill = w_min + rng_normal0(key, 2 * j + 1, P_FILLIN, 0) * sigma;
      const double dxp = fill - w_min;
      acc[4] += 0.5 * w_curv * dxp * dxp - 0.5 * log(w_curv);
    }
    tp[2 * j] = x_m;
    tp[2 * j + 1] = fill;
    const double t_m = th[2 * j], t_o = th[2 * j + 1], t_p = th[2 * jn];
    w_conditioned<KIND>(Pf, t_m, t_p, w_min, w_curv);
    const double dx = t_o - w_min;
    if (KIND == MLMCPI_ROTOR) {
      const double sigma = 2. * w_curv, sh = sin(0.5 * dx);
      acc[5] += sigma * sh * sh + log(two_pi_i0_scaled(0.5 * sigma));
    } else {
      acc[5] += 0.5 * w_curv * dx * dx - 0.5 * log(w_curv);
    }
    // actions: sites 2j+1 and 2j+2 of the fine paths, site j+1 of the coarse ones (each site once over all j)
    acc[0] += site_energy<KIND>(Pf, fill, x_m) + site_energy<KIND>(Pf, x_p, fill);
    acc[1] += site_energy<KIND>(Pf, t_o, t_m) + site_energy<KIND>(Pf, t_p, t_o);
    acc[2] += site_energy<KIND>(Pc, t_p, t_m);
    acc[3] += site_energy<KIND>(Pc, x_p, x_m);
  }
  block_sum<6>(acc, red);
  if (threadIdx.x == 0) {
    const double dS_fine = energy_scale(Pf) * acc[0] - energy_scale(Pf) * acc[1];
    const double dS_coarse = energy_scale(Pc) * acc[2] - energy_scale(Pc) * acc[3];
    const double dS_trial = acc[5] - acc[4];
    const double dS = dS_fine + dS_coarse + dS_trial;
    bool ok = dS < 0.0;
    if (!ok) {
      double u, v;
      rng_uniforms(key, 0, P_ACCEPT2, 0, u, v);
      ok = u < exp(-dS);
    }
    decision = ok ? 1 : 0;
    accept[b] = decision;
    if (terms) {
      terms[3 * b + 0] = dS_fine; terms[3 * b + 1] = dS_coarse; terms[3 * b + 2] = dS_trial;
    }
  }
  __syncthreads();
  if (!decision) return;
  for (uint32_t j = threadIdx.x; j < Mc; j += blockDim.x) {  // the entries this thread wrote above
    th[2 * j] = tp[2 * j];
    th[2 * j + 1] = tp[2 * j + 1];
  }
}

// ---- host dispatch ---------------------------------------------------------------------------------------
static int check_action(const mlmcpi_path_action *act) {
  if (!act) return fail(MLMCPI_ERR_INVALID, "action is NULL");
  if (act->kind < MLMCPI_HARMONIC || act->kind > MLMCPI_ROTOR)
    return fail(MLMCPI_ERR_INVALID, "kind %d is not a 1-D path action", act->kind);
  if (act->M < 2) return fail(MLMCPI_ERR_INVALID, "M_lat = %u too small", act->M);
  if (!(act->T_final > 0.0)) return fail(MLMCPI_ERR_INVALID, "T_final must be positive");
  return MLMCPI_OK;
}

static uint32_t choose_split(uint32_t sites, uint32_t B) {
  // enough workgroups to fill 256 CUs, at least ~1024 sites each
  uint32_t want = (2048 + B - 1) / B, cap = (sites + 1023) / 1024;
  uint32_t n = want < cap ? want : cap;
  return n ? n : 1;
}

template <int OP>
static int launch_reduce(const PathP &P, const double *d_x, uint32_t B, double scale, double *d_out, hipStream_t st,
                         uint32_t stride = 1) {
  const uint32_t nsplit = choose_split(P.M, B);
  void *ws = nullptr;
  int rc = scratch((size_t)B * nsplit * sizeof(double), &ws, st);
  if (rc) return rc;
  dim3 grid(nsplit, B), block(256);
  switch (P.kind) {
    case MLMCPI_HARMONIC:
      hipLaunchKernelGGL((path_reduce_kernel<MLMCPI_HARMONIC, OP>), grid, block, 0, st, P, d_x, (double *)ws, stride, scale, d_out);
      break;
    case MLMCPI_QUARTIC:
      hipLaunchKernelGGL((path_reduce_kernel<MLMCPI_QUARTIC, OP>), grid, block, 0, st, P, d_x, (double *)ws, stride, scale, d_out);
      break;
    default:
      hipLaunchKernelGGL((path_reduce_kernel<MLMCPI_ROTOR, OP>), grid, block, 0, st, P, d_x, (double *)ws, stride, scale, d_out);
  }
  MLMCPI_LAUNCH_CHECK("path_reduce_kernel");
  if (nsplit == 1) return MLMCPI_OK;
  hipLaunchKernelGGL(path_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, st, (const double *)ws, nsplit, B, OP,
                     scale, d_out);
  MLMCPI_LAUNCH_CHECK("path_finish_kernel");
  return MLMCPI_OK;
}

struct HmcPlan {
  uint32_t R, NT, nseg, owned_len, halo;
};

// Register-resident geometry: the whole periodic path in one workgroup when M = NT*R fits
// (NT a multiple of 64, R in {1,2,4,8,16}, NT <= 512 for R >= 8 so that 2R doubles of state plus
// the sine's temporaries stay in VGPRs, else <= 1024), i.e. M <= 8192; longer paths are cut into
// segments of NT*R buffer sites with a halo of nt+1.
static int plan_hmc(int kind, uint32_t M, uint32_t B, uint32_t nt, HmcPlan *plan) {
  const uint32_t maxR = (kind == MLMCPI_ROTOR) ? 8 : 16;  // the rotor's sines need the registers
  static const uint32_t Rs[5] = {16, 8, 4, 2, 1};
  uint32_t best = 0, fallback = 0, best_nt = 0, fallback_nt = 0;
  for (uint32_t R : Rs) {
    if (R > maxR || M % R) continue;
    uint32_t NT = M / R;
    if (NT % 64 || NT < 64 || NT > (R >= 8 ? 512u : 1024u)) continue;  // register budget: see launch bounds
    if (!fallback || NT > fallback_nt) { fallback = R; fallback_nt = NT; }  // most parallel
    if (!best && (uint64_t)B * (NT / 64) >= 2048) { best = R; best_nt = NT; }  // largest R that still fills the chip
  }
  if (best || fallback) {
    plan->R = best ? best : fallback;
    plan->NT = best ? best_nt : fallback_nt;
    plan->nseg = 1;
    plan->owned_len = M;
    plan->halo = 0;
    return MLMCPI_OK;
  }
  const uint32_t halo = nt + 1;
  uint32_t R = maxR, NT = 4096 / maxR;
  if (M + 2 * halo <= 1024) { R = 4; NT = 256; }  // short odd-sized paths
  const uint32_t L = NT * R;
  if (2 * halo + 64 > L) return fail(MLMCPI_ERR_INVALID, "nt = %u too long for the fused trajectory (max %u)", nt, (L - 64) / 2 - 1);
  const uint32_t owned_max = L - 2 * halo;
  const uint32_t nseg = (M + owned_max - 1) / owned_max;
  plan->R = R;
  plan->NT = NT;
  plan->nseg = nseg;
  plan->owned_len = (M + nseg - 1) / nseg;
  plan->halo = halo;
  return MLMCPI_OK;
}

template <int KIND>
static int launch_traj(const HmcPlan &pl, const PathP &P, const double *x_cur, double *x_trial, double *partials,
                       const int32_t *done, uint32_t B, uint32_t nt, double dt, RngKey key, hipStream_t st) {
  dim3 grid(pl.nseg, B), block(pl.NT);
  const size_t lds = (4 * pl.NT + 4 * (pl.NT / 64) + (size_t)pl.R * pl.NT) * sizeof(double);
#define MLMCPI_TRAJ(RR)                                                                                          \
  hipLaunchKernelGGL((hmc_trajectory_kernel<KIND, RR>), grid, block, lds, st, P, x_cur, x_trial, partials, done, \
                     pl.owned_len, pl.halo, nt, dt, key)
  switch (pl.R) {
    case 16: MLMCPI_TRAJ(16); break;
    case 8: MLMCPI_TRAJ(8); break;
    case 4: MLMCPI_TRAJ(4); break;
    case 2: MLMCPI_TRAJ(2); break;
    default: MLMCPI_TRAJ(1);
  }
#undef MLMCPI_TRAJ
  MLMCPI_LAUNCH_CHECK("hmc_trajectory_kernel");
  return MLMCPI_OK;
}

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace mlmcpi

using namespace mlmcpi;

extern "C" {

int mlmcpi_path_evaluate(const mlmcpi_path_action *act, const double *d_x, uint32_t B, double *d_S, void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_x && d_S && B > 0, "bad arguments");
  PathP P = make_params(*act);
  return launch_reduce<R_ENERGY>(P, d_x, B, energy_scale(P), d_S, as_stream(stream));
}

int mlmcpi_path_force(const mlmcpi_path_action *act, const double *d_x, double *d_f, uint32_t B, void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_x && d_f && B > 0 && d_x != d_f, "bad arguments");
  PathP P = make_params(*act);
  dim3 grid(choose_split(P.M, B) , B), block(256);
  hipStream_t st = as_stream(stream);
  switch (P.kind) {
    case MLMCPI_HARMONIC: hipLaunchKernelGGL(path_force_kernel<MLMCPI_HARMONIC>, grid, block, 0, st, P, d_x, d_f); break;
    case MLMCPI_QUARTIC: hipLaunchKernelGGL(path_force_kernel<MLMCPI_QUARTIC>, grid, block, 0, st, P, d_x, d_f); break;
    default: hipLaunchKernelGGL(path_force_kernel<MLMCPI_ROTOR>, grid, block, 0, st, P, d_x, d_f);
  }
  MLMCPI_LAUNCH_CHECK("path_force_kernel");
  return MLMCPI_OK;
}

int mlmcpi_path_initialise(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint64_t seed, uint32_t chain0,
                           void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_x && B > 0, "bad arguments");
  dim3 grid(choose_split(act->M, B), B), block(256);
  hipLaunchKernelGGL(path_init_kernel, grid, block, 0, as_stream(stream), act->kind, act->M, make_key(seed, chain0, 0),
                     d_x);
  MLMCPI_LAUNCH_CHECK("path_init_kernel");
  return MLMCPI_OK;
}

int mlmcpi_qoi_xsquared(const double *d_x, uint32_t M, uint32_t B, double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_x && d_out && B > 0 && M > 1, "bad arguments");
  PathP P = {};
  P.kind = MLMCPI_HARMONIC;
  P.M = M;
  return launch_reduce<R_XSQUARED>(P, d_x, B, 1.0 / M, d_out, as_stream(stream));
}

int mlmcpi_qoi_susceptibility(const double *d_x, uint32_t M, double T_final, uint32_t B, double *d_out,
                              void *stream) {
  MLMCPI_REQUIRE(d_x && d_out && B > 0 && M > 1 && T_final > 0, "bad arguments");
  PathP P = {};
  P.kind = MLMCPI_ROTOR;
  P.M = M;
  return launch_reduce<R_WINDING>(P, d_x, B, 1.0 / T_final, d_out, as_stream(stream));
}

// workspace layout: x_trial [B*M] | partials [B*nseg*4] | flags [2][B] int32
int mlmcpi_path_hmc_workspace_bytes(const mlmcpi_path_action *act, uint32_t B, uint32_t nt, size_t *bytes) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(bytes && B > 0, "bad arguments");
  HmcPlan pl;
  if (int rc = plan_hmc(act->kind, act->M, B, nt, &pl)) return rc;
  *bytes = align256((size_t)B * act->M * 8) + align256((size_t)B * pl.nseg * 4 * 8) + align256((size_t)2 * B * 4);
  return MLMCPI_OK;
}

int mlmcpi_path_hmc_draw(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint32_t nt, double dt,
                         uint32_t n_rep, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                         int32_t *d_accept, double *d_energies, void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_x && d_work && B > 0 && n_rep > 0, "bad arguments");
  HmcPlan pl;
  if (int rc = plan_hmc(act->kind, act->M, B, nt, &pl)) return rc;
  PathP P = make_params(*act);
  hipStream_t st = as_stream(stream);
  char *w = (char *)d_work;
  double *x_trial = (double *)w;
  w += align256((size_t)B * P.M * 8);
  double *partials = (double *)w;
  w += align256((size_t)B * pl.nseg * 4 * 8);
  int32_t *flags = (int32_t *)w;  // [2][B]
  // accept flags ping-pong between the two halves of `flags`; the first repetition has none to read (null) and the last
  // one writes the caller's array directly: a draw is n_rep x (trajectory, accept) and nothing else on the stream
  const uint32_t copy_blocks = choose_split(P.M, B);
  for (uint32_t r = 0; r < n_rep; ++r) {
    const int32_t *done_in = r == 0 ? nullptr : flags + (size_t)(r & 1) * B;
    int32_t *done_out = (r + 1 == n_rep && d_accept) ? d_accept : flags + (size_t)((r + 1) & 1) * B;
    RngKey key = make_key(seed, chain0, traj0 + r);
    int rc;
    switch (P.kind) {
      case MLMCPI_HARMONIC: rc = launch_traj<MLMCPI_HARMONIC>(pl, P, d_x, x_trial, partials, done_in, B, nt, dt, key, st); break;
      case MLMCPI_QUARTIC: rc = launch_traj<MLMCPI_QUARTIC>(pl, P, d_x, x_trial, partials, done_in, B, nt, dt, key, st); break;
      default: rc = launch_traj<MLMCPI_ROTOR>(pl, P, d_x, x_trial, partials, done_in, B, nt, dt, key, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(hmc_accept_kernel, dim3(copy_blocks, B), dim3(256), 0, st, P.M, energy_scale(P), d_x,
                       (const double *)x_trial, (const double *)partials, pl.nseg, done_in, done_out, d_energies, key);
    MLMCPI_LAUNCH_CHECK("hmc_accept_kernel");
  }
  return MLMCPI_OK;
}

// see sweep_draw_impl of lattice2d.hip: reads d_x first, then alternates between d_w0 and d_w1 (which may be d_x)
static int path_sweep_impl(const mlmcpi_path_action *act, double *d_x, double *d_w0, double *d_w1, uint32_t B,
                           uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                           uint32_t sweep0, int32_t *result_in, void *stream, double *d_qoi = nullptr, double *d_acc = nullptr) {
  if (int rc = check_action(act)) return rc;
  // action/action.hh:73-96: only the rotor implements local updates among the 1-D actions
  if (act->kind != MLMCPI_ROTOR)
    return fail(MLMCPI_ERR_UNSUPPORTED, "heat bath / overrelaxation update not implemented for this action");
  MLMCPI_REQUIRE(d_x && d_w0 && d_w1 && d_x != d_w0 && d_w0 != d_w1 && B > 0, "bad arguments");
  MLMCPI_REQUIRE(act->M % 2 == 0, "even/odd sweeps need an even number of sites (M_lat = %u)", act->M);
  PathP P = make_params(*act);
  hipStream_t st = as_stream(stream);
  const uint32_t total = n_overrelax + n_heatbath;
  const Tuning tune = tuning();
  const bool split_heat = tune.or_heat_split, closed = !tune.or_block;
  double *src = d_x, *dst = d_w0;
  uint32_t s = 0;
  while (s < total) {
    // overrelaxation sweeps (they come first, sampler order) are fused up to 8 per launch: in one dimension the halo
    // of 2 sites per sweep costs next to nothing; a heat-bath sweep gets a launch of its own (sampler-bound)
    uint32_t n = 1, kinds = 0, n_closed = 0;
    if (s < n_overrelax) {
      // overrelaxation in closed form (rotor_sweep_kernel; MLMCPI_OR_KERNEL=block: sweep by sweep): up to 16 sweeps per launch
      const uint32_t cap = closed ? 16u : 8u;
      n = n_overrelax - s < cap ? n_overrelax - s : cap;
      if (closed) n_closed = n;
      // the last overrelaxation launch takes the heat-bath sweep behind it along (one pass over the state less; the sweeps
      // of a launch are numbered on from its key, so the draws are those of two launches: MLMCPI_OR_HEAT=split)
      if (!split_heat && s + n == n_overrelax && n_heatbath >= 1 && (n < cap || closed)) {
        kinds = 1u << n;
        ++n;
      }
    } else
      kinds = 1u;
    // the draw's last launch can sum the topological charge of the new state while the segment is in LDS (d_qoi)
    const bool with_qoi = d_qoi && s + n == total;
    const uint32_t halo = 2 * n + (with_qoi ? 2 : 0);
    uint32_t owned = 2048 - 2 * halo;  // even
    if (owned > P.M) owned = P.M;
    const uint32_t nseg = (P.M + owned - 1) / owned;
    owned = (P.M + nseg - 1) / nseg;
    owned += owned & 1;  // keep segment starts even
    const uint32_t nseg2 = (P.M + owned - 1) / owned;
    const size_t lds = (size_t)(owned + 2 * halo) * sizeof(double);
    // retry pool of the heat-bath phases (device_common.hpp); which sampler: a property of the action (2 m0 / a), not a knob
    const bool step = 2.0 * P.m0 / P.a <= kVsKappaMax;
    const uint32_t pool_cap = 256;
    const uint32_t *vs_table = nullptr;
    if (kinds && step)
      if (int rc = vs_table_device(2.0 * P.m0 / P.a, &vs_table)) return rc;
    void *partial = nullptr;
    if (with_qoi)
      if (int rc = scratch((size_t)B * nseg2 * sizeof(double), &partial, st)) return rc;
    if (kinds && step)
      hipLaunchKernelGGL((rotor_sweep_kernel<true, true>), dim3(nseg2, B), dim3(256), lds + VsPool<uint32_t>::bytes(pool_cap), st, P,
                         (const double *)src, dst, owned, n, kinds, make_key(seed, chain0, sweep0 + s), pool_cap, vs_table, (double *)partial, n_closed);
    else if (kinds)
      hipLaunchKernelGGL(rotor_sweep_kernel<true>, dim3(nseg2, B), dim3(256), lds + HbPool::bytes(pool_cap), st, P,
                         (const double *)src, dst, owned, n, kinds, make_key(seed, chain0, sweep0 + s), pool_cap, vs_table, (double *)partial, n_closed);
    else
      hipLaunchKernelGGL(rotor_sweep_kernel<false>, dim3(nseg2, B), dim3(256), lds, st, P, (const double *)src, dst, owned, n,
                         kinds, make_key(seed, chain0, sweep0 + s), 0u, vs_table, (double *)partial, n_closed);
    MLMCPI_LAUNCH_CHECK("rotor_sweep_kernel");
    if (with_qoi) {
      hipLaunchKernelGGL(path_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, st, (const double *)partial, nseg2, B,
                         (int)R_WINDING, 1.0 / act->T_final, d_qoi, d_acc);
      MLMCPI_LAUNCH_CHECK("path_finish_kernel");
    }
    src = dst;
    dst = (dst == d_w0) ? d_w1 : d_w0;
    s += n;
  }
  if (result_in)
    *result_in = total == 0 ? -1 : (src == d_w0 ? 0 : 1);
  else if (src != d_x)
    MLMCPI_HIP_TRY(hipMemcpyAsync(d_x, src, (size_t)B * P.M * 8, hipMemcpyDeviceToDevice, st));
  return MLMCPI_OK;
}

int mlmcpi_path_site_updates(const mlmcpi_path_action *act, double *d_x, uint32_t B, const uint32_t *d_sites, uint32_t n,
                             uint32_t site, int32_t heat, uint64_t seed, uint32_t chain0, uint32_t step, void *stream) {
  if (int rc = check_action(act)) return rc;
  if (act->kind != MLMCPI_ROTOR)
    return fail(MLMCPI_ERR_UNSUPPORTED, "heat bath / overrelaxation update not implemented for this action");
  MLMCPI_REQUIRE(d_x && B > 0, "bad arguments");
  if (!d_sites) {
    MLMCPI_REQUIRE(site < act->M, "site %u out of range (%u sites)", site, act->M);
    n = 1;
  }
  if (n == 0) return MLMCPI_OK;
  const PathP P = make_params(*act);
  const uint32_t *vs_table = nullptr;
  if (heat && 2.0 * P.m0 / P.a <= kVsKappaMax)
    if (int rc = vs_table_device(2.0 * P.m0 / P.a, &vs_table)) return rc;
  hipLaunchKernelGGL(rotor_site_update_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), P, d_x, B, d_sites, n, site,
                     (int)heat, make_key(seed, chain0, step), vs_table);
  MLMCPI_LAUNCH_CHECK("rotor_site_update_kernel");
  return MLMCPI_OK;
}

int mlmcpi_path_sweep_draw(const mlmcpi_path_action *act, double *d_x, double *d_scratch, uint32_t B,
                           uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                           uint32_t sweep0, void *stream) {
  return path_sweep_impl(act, d_x, d_scratch, d_x, B, n_overrelax, n_heatbath, seed, chain0, sweep0, nullptr, stream);
}

int mlmcpi_path_sweep_draw_from(const mlmcpi_path_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                                uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                                int32_t *result_in, void *stream) {
  MLMCPI_REQUIRE(result_in, "result_in is NULL");
  MLMCPI_REQUIRE(n_overrelax + n_heatbath > 0, "no sweeps requested: the result would be the (read-only) input");
  return path_sweep_impl(act, const_cast<double *>(d_src), d_w0, d_w1, B, n_overrelax, n_heatbath, seed, chain0, sweep0,
                         result_in, stream);
}

int mlmcpi_path_sweep_draw_qoi(const mlmcpi_path_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                               uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                               double *d_qoi, double *d_acc, int32_t *result_in, void *stream) {
  MLMCPI_REQUIRE(result_in && d_qoi, "result_in or d_qoi is NULL");
  MLMCPI_REQUIRE(n_overrelax + n_heatbath > 0, "no sweeps requested: the result would be the (read-only) input");
  return path_sweep_impl(act, const_cast<double *>(d_src), d_w0, d_w1, B, n_overrelax, n_heatbath, seed, chain0, sweep0,
                         result_in, stream, d_qoi, d_acc);
}

// workspace: theta' [B*M] (the trial state; kept at the r02 size, which also held reduction partials)
static uint32_t twolevel_blocks(uint32_t M, uint32_t B) { return choose_split(M / 2, B); }

int mlmcpi_path_twolevel_workspace_bytes(const mlmcpi_path_action *fine, uint32_t B, size_t *bytes) {
  if (int rc = check_action(fine)) return rc;
  MLMCPI_REQUIRE(bytes && B > 0, "bad arguments");
  *bytes = align256((size_t)B * fine->M * 8) + align256((size_t)4 * B * 8) +
           align256((size_t)B * twolevel_blocks(fine->M, B) * 2 * 8);
  return MLMCPI_OK;
}

int mlmcpi_path_twolevel_draw(const mlmcpi_path_action *fine, const mlmcpi_path_action *coarse, const double *d_x_coarse,
                              double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step, void *d_work,
                              int32_t *d_accept, double *d_terms, void *stream) {
  return mlmcpi_path_twolevel_draw_masked(fine, coarse, d_x_coarse, d_theta, B, seed, chain0, step, d_work, nullptr, d_accept, d_terms,
                                          stream);
}

int mlmcpi_path_twolevel_draw_masked(const mlmcpi_path_action *fine, const mlmcpi_path_action *coarse, const double *d_x_coarse,
                                     double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step, void *d_work,
                                     const int32_t *d_mask, int32_t *d_accept, double *d_terms, void *stream) {
  if (int rc = check_action(fine)) return rc;
  if (int rc = check_action(coarse)) return rc;
  MLMCPI_REQUIRE(d_x_coarse && d_theta && d_work && d_accept && B > 0, "bad arguments");
  MLMCPI_REQUIRE(fine->M % 2 == 0 && coarse->M == fine->M / 2 && coarse->kind == fine->kind,
                 "coarse action must live on the lattice with half the sites (M %u vs %u)", coarse->M, fine->M);
  hipStream_t st = as_stream(stream);
  const PathP Pf = make_params(*fine), Pc = make_params(*coarse);
  double *theta_prime = (double *)d_work;
  const RngKey key = make_key(seed, chain0, step);
  // one workgroup per chain; as many threads as there are coarse sites, up to 1024 (then several sites per thread)
  uint32_t nt = 64;
  while (nt < (Pf.kind == MLMCPI_ROTOR ? 512u : 1024u) && nt < Pf.M / 2) nt *= 2;
  const dim3 grid(B), block(nt);
  if (Pf.kind == MLMCPI_HARMONIC)
    hipLaunchKernelGGL(twolevel_fused_kernel<MLMCPI_HARMONIC>, grid, block, 0, st, Pf, Pc, d_x_coarse, d_theta, theta_prime, d_accept, d_terms, key, d_mask);
  else if (Pf.kind == MLMCPI_QUARTIC)
    hipLaunchKernelGGL(twolevel_fused_kernel<MLMCPI_QUARTIC>, grid, block, 0, st, Pf, Pc, d_x_coarse, d_theta, theta_prime, d_accept, d_terms, key, d_mask);
  else
    hipLaunchKernelGGL(twolevel_fused_kernel<MLMCPI_ROTOR>, grid, block, 0, st, Pf, Pc, d_x_coarse, d_theta, theta_prime, d_accept, d_terms, key, d_mask);
  MLMCPI_LAUNCH_CHECK("twolevel_fused_kernel");
  return MLMCPI_OK;
}

int mlmcpi_path_hmc_run(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint32_t nt, double dt, uint32_t n_rep,
                        uint32_t n_draws, int qoi_kind, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                        double *d_qoi, int32_t *d_accept_count, void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_x && d_work && B > 0 && n_rep > 0 && n_draws > 0, "bad arguments");
  MLMCPI_REQUIRE(qoi_kind >= 0 && qoi_kind <= 2 && (qoi_kind == 0 || d_qoi), "bad QoI selection");
  HmcPlan pl;
  if (int rc = plan_hmc(act->kind, act->M, B, nt, &pl)) return rc;
  PathP P = make_params(*act);
  hipStream_t st = as_stream(stream);
  if (pl.halo == 0) {
    // register-resident periodic path: everything in one launch, one workgroup per chain
    const size_t lds = (4 * pl.NT + 4 * (pl.NT / 64) + (size_t)pl.R * pl.NT + 1) * sizeof(double);
    const RngKey key = make_key(seed, chain0, traj0);
#define MLMCPI_CHAIN(KK, RR) hipLaunchKernelGGL((hmc_chain_kernel<KK, RR>), dim3(B), dim3(pl.NT), lds, st, P, d_x, d_qoi, d_accept_count, (double *)nullptr, nt, dt, n_rep, n_draws, qoi_kind, key)
#define MLMCPI_CHAIN_R(KK) switch (pl.R) { case 16: MLMCPI_CHAIN(KK, 16); break; case 8: MLMCPI_CHAIN(KK, 8); break; case 4: MLMCPI_CHAIN(KK, 4); break; case 2: MLMCPI_CHAIN(KK, 2); break; default: MLMCPI_CHAIN(KK, 1); }
    switch (P.kind) {
      case MLMCPI_HARMONIC: MLMCPI_CHAIN_R(MLMCPI_HARMONIC); break;
      case MLMCPI_QUARTIC: MLMCPI_CHAIN_R(MLMCPI_QUARTIC); break;
      default: MLMCPI_CHAIN_R(MLMCPI_ROTOR);
    }
#undef MLMCPI_CHAIN_R
#undef MLMCPI_CHAIN
    MLMCPI_LAUNCH_CHECK("hmc_chain_kernel");
    return MLMCPI_OK;
  }
  // segmented paths (M > 8192 or not a multiple of 64): same sequence through the per-draw entry points
  int32_t *acc_tmp = nullptr;
  if (d_accept_count) MLMCPI_HIP_TRY(hipMemsetAsync(d_accept_count, 0, (size_t)B * 4, st));
  MLMCPI_HIP_TRY(hipMalloc((void **)&acc_tmp, (size_t)B * 4));
  int rc = MLMCPI_OK;
  for (uint32_t d = 0; d < n_draws && !rc; ++d) {
    rc = mlmcpi_path_hmc_draw(act, d_x, B, nt, dt, n_rep, seed, chain0, traj0 + d * n_rep, d_work, acc_tmp, nullptr, stream);
    if (!rc && d_accept_count) {
      hipLaunchKernelGGL(add_flags_kernel, dim3((B + 255) / 256), dim3(256), 0, st, d_accept_count, (const int32_t *)acc_tmp, B);
    }
    if (!rc && qoi_kind == 1) rc = launch_reduce<R_XSQUARED>(P, d_x, B, 1.0 / P.M, d_qoi + (size_t)d * B, st);
    if (!rc && qoi_kind == 2) rc = launch_reduce<R_WINDING>(P, d_x, B, 1.0 / P.T_final, d_qoi + (size_t)d * B, st);
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(acc_tmp);
  return rc;
}

int mlmcpi_path_hmc_run_layout(const mlmcpi_path_action *act, uint32_t B, uint32_t nt, int32_t *chain_major) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(chain_major && B > 0, "bad arguments");
  HmcPlan pl;
  if (int rc = plan_hmc(act->kind, act->M, B, nt, &pl)) return rc;
  *chain_major = pl.halo == 0 ? 1 : 0;
  return MLMCPI_OK;
}

int mlmcpi_path_copy_from_fine(const double *d_fine, double *d_coarse, uint32_t M_coarse, uint32_t B, void *stream) {
  MLMCPI_REQUIRE(d_fine && d_coarse && M_coarse > 0 && B > 0, "bad arguments");
  hipLaunchKernelGGL(path_transfer_kernel, dim3(choose_split(M_coarse, B), B), dim3(256), 0, as_stream(stream), M_coarse,
                     (double *)d_fine, d_coarse, 1);
  MLMCPI_LAUNCH_CHECK("path_transfer_kernel");
  return MLMCPI_OK;
}

int mlmcpi_path_copy_from_coarse(const double *d_coarse, double *d_fine, uint32_t M_coarse, uint32_t B, void *stream) {
  MLMCPI_REQUIRE(d_fine && d_coarse && M_coarse > 0 && B > 0, "bad arguments");
  hipLaunchKernelGGL(path_transfer_kernel, dim3(choose_split(M_coarse, B), B), dim3(256), 0, as_stream(stream), M_coarse,
                     d_fine, (double *)d_coarse, 0);
  MLMCPI_LAUNCH_CHECK("path_transfer_kernel");
  return MLMCPI_OK;
}

}  // extern "C"

// =================================================================================================
// Exact sampler of the harmonic oscillator: HarmonicOscillatorAction::build_covariance / draw
// (action/qm/harmonicoscillatoraction.cc:38-66).  x = L y with y ~ N(0, 1)^M and L the lower Cholesky factor of the
// covariance (the inverse of the circulant precision matrix) -- for B chains a dense [B x M] . [M x M] product,
// the one fp64 matrix-core job of the path: v_mfma_f64_16x16x4_f64, one wave per 16 chains x 16 sites output tile.
//   A operand: y tile, lane l holds y[chain l & 15][k = k0 + (l >> 4)]      (from LDS; Philox, purpose P_EXACT)
//   B operand: L^T tile, lane l holds L[site j0 + (l & 15)][k = k0 + (l >> 4)] = LT[k][j] (coalesced along j)
//   C / D:     lane l, register r holds x[chain (l >> 4) + 4 r][site j0 + (l & 15)]   (f64 map, not the f32 one)
// L is lower triangular, so a site tile stops at k <= j0 + 15.
// =================================================================================================
namespace mlmcpi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int TJ>  // site tiles per wave
__global__ void __launch_bounds__(256)
    ho_exact_draw_kernel(uint32_t M, uint32_t B, const double *__restrict__ LT, double *__restrict__ x, RngKey key0) {
  constexpr int KC = 64;  // k values generated per round
  __shared__ double ylds[KC * 16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t c0 = blockIdx.y * 16u;                  // first chain of this workgroup
  const uint32_t jw0 = (blockIdx.x * 4u + wave) * TJ * 16u;  // first site of this wave
  const uint32_t j_hi_block = min(M, (blockIdx.x + 1u) * 4u * TJ * 16u);  // sites of the workgroup end here
  v4f64 acc[TJ];
#pragma unroll
  for (int t = 0; t < TJ; ++t) acc[t] = (v4f64){0., 0., 0., 0.};
  for (uint32_t k0 = 0; k0 < j_hi_block; k0 += KC) {
    __syncthreads();
    // y[chain][k0 .. k0 + KC): 16 chains x KC/2 Box-Muller pairs = 512 pairs for 256 threads
    for (uint32_t p = threadIdx.x; p < 16u * (KC / 2); p += 256u) {
      const uint32_t r = p & 15u, q = p >> 4;
      const uint32_t k = k0 + 2u * q;
      double n0 = 0.0, n1 = 0.0;
      if (c0 + r < B && k < M) {
        RngKey key = key0;
        key.chain += c0 + r;
        rng_normals(key, k >> 1, P_EXACT, 0, n0, n1);
      }
      ylds[(2u * q) * 16u + r] = n0;
      ylds[(2u * q + 1u) * 16u + r] = n1;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const uint32_t j0 = jw0 + t * 16u;
      if (j0 >= M || k0 > j0 + 15u) continue;             // wave-uniform: tile outside the lattice / above the diagonal
      const uint32_t j = j0 + (lane & 15u);
      for (uint32_t kk = 0; kk < (uint32_t)KC && k0 + kk <= j0 + 15u; kk += 4u) {
        const uint32_t k = k0 + kk + (lane >> 4);
        const double a = ylds[(kk + (lane >> 4)) * 16u + (lane & 15u)];
        const double bv = (k < M && j < M) ? LT[(size_t)k * M + j] : 0.0;
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TJ; ++t) {
    const uint32_t j = jw0 + t * 16u + (lane & 15u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t chain = c0 + (lane >> 4) + 4u * r;
      if (chain < B && j < M) x[(size_t)chain * M + j] = acc[t][r];
    }
  }
}

}  // namespace mlmcpi

extern "C" {

// harmonicoscillatoraction.cc:38-56.  The precision matrix Q = circ(d, c, 0, ..., 0, c) is circulant, so its inverse is
// the circulant with first row C_k = (1/M) sum_m cos(2 pi m k / M) / (d + 2 c cos(2 pi m / M)); then a Cholesky
// decomposition C = L L^T (Cholesky-Banachiewicz, O(M^3 / 3) on the host, as in the reference).  h_LT receives L^T
// row-major, i.e. h_LT[k * M + j] = L[j][k], the layout the device product reads.
int mlmcpi_ho_cholesky_factor(const mlmcpi_path_action *act, double *h_LT) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(h_LT, "h_LT is NULL");
  if (act->kind != MLMCPI_HARMONIC) return fail(MLMCPI_ERR_UNSUPPORTED, "exact sampler only for the harmonic oscillator action");
  const uint32_t M = act->M;
  MLMCPI_REQUIRE(M <= 4096, "exact sampler: M_lat = %u needs a dense %u x %u factor (limit 4096)", M, M, M);
  const double a = act->T_final / M, d = a * act->m0 * act->mu2 + 2.0 * act->m0 / a, c = -act->m0 / a;
  std::vector<double> row(M), L((size_t)M * M, 0.0);
  for (uint32_t k = 0; k < M; ++k) {
    double s = 0.0;
    for (uint32_t m = 0; m < M; ++m) s += std::cos(2.0 * kPi * (double)((uint64_t)m * k % M) / M) / (d + 2.0 * c * std::cos(2.0 * kPi * m / M));
    row[k] = s / M;
  }
  auto C = [&](uint32_t i, uint32_t j) { return row[(i + M - j) % M]; };
  for (uint32_t i = 0; i < M; ++i)
    for (uint32_t j = 0; j <= i; ++j) {
      double s = C(i, j);
      for (uint32_t k = 0; k < j; ++k) s -= L[(size_t)i * M + k] * L[(size_t)j * M + k];
      if (i == j) {
        if (!(s > 0.0)) return fail(MLMCPI_ERR_INVALID, "covariance matrix is not positive definite");
        L[(size_t)i * M + i] = std::sqrt(s);
      } else {
        L[(size_t)i * M + j] = s / L[(size_t)j * M + j];
      }
    }
  for (uint32_t j = 0; j < M; ++j)
    for (uint32_t k = 0; k < M; ++k) h_LT[(size_t)k * M + j] = L[(size_t)j * M + k];
  return MLMCPI_OK;
}

int mlmcpi_path_exact_draw(const mlmcpi_path_action *act, const double *d_LT, double *d_x, uint32_t B, uint64_t seed,
                           uint32_t chain0, uint32_t step, void *stream) {
  if (int rc = check_action(act)) return rc;
  MLMCPI_REQUIRE(d_LT && d_x && B > 0, "bad arguments");
  if (act->kind != MLMCPI_HARMONIC) return fail(MLMCPI_ERR_UNSUPPORTED, "exact sampler only for the harmonic oscillator action");
  const uint32_t M = act->M;
  const dim3 grid((M + 127) / 128, (B + 15) / 16);  // 4 waves x 2 site tiles x 16 sites = 128 sites per workgroup
  hipLaunchKernelGGL(ho_exact_draw_kernel<2>, grid, dim3(256), 0, as_stream(stream), M, B, d_LT, d_x, make_key(seed, chain0, step));
  MLMCPI_LAUNCH_CHECK("ho_exact_draw_kernel");
  return MLMCPI_OK;
}

}  // extern "C"
