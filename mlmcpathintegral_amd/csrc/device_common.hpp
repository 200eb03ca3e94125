// device_common.hpp -- gfx950 device helpers shared by the sweep / HMC / QoI kernels:
// counter-based RNG (Philox4x32-10), the two rejection samplers used by the heat-bath updates,
// mod_2pi, and wave64 / workgroup reductions.  Device code only; the CPU oracle has its own,
// independently written restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlmcpi {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 6.28318530717958647692;
constexpr int kWave = 64;

// common/auxilliary.hh:42-44
__device__ __forceinline__ double mod_2pi(double x) { return x - 2. * kPi * floor(0.5 * (x + kPi) / kPi); }

// ---- RNG contract ------------------------------------------------------------------
enum Purpose : uint32_t {
  P_MOMENTUM = 1,
  P_ACCEPT = 2,
  P_GFF_NORMAL = 3,
  P_REJ_NORMAL = 4,
  P_REJ_UNIFORM = 5,
  P_INIT = 6,
};

struct RngKey {
  uint32_t k0, k1;  // seed low / high word
  uint32_t chain;   // global chain index
  uint32_t step;    // sweep or trajectory counter
};

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u01(uint32_t lo, uint32_t hi) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void rng_uniforms(const RngKey &k, uint32_t site, uint32_t purpose, uint32_t sub,
                                             double &a, double &b) {
  U4 r = philox4x32_10(site, k.chain, k.step, (purpose << 24) | (sub & 0xFFFFFFu), k.k0, k.k1);
  a = u01(r.x, r.y);
  b = u01(r.z, r.w);
}

__device__ __forceinline__ void rng_normals(const RngKey &k, uint32_t site, uint32_t purpose, uint32_t sub,
                                            double &n0, double &n1) {
  double u, v;
  rng_uniforms(k, site, purpose, sub, u, v);
  double r = sqrt(-2.0 * log(1.0 - u));
  double s, c;
  sincos(kTwoPi * v, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// cosine branch only (one normal per call)
__device__ __forceinline__ double rng_normal0(const RngKey &k, uint32_t site, uint32_t purpose, uint32_t sub) {
  double u, v;
  rng_uniforms(k, site, purpose, sub, u, v);
  return sqrt(-2.0 * log(1.0 - u)) * cos(kTwoPi * v);
}

// ---- rejection samplers -----------------------------------------------------------------------
// Attempt a uses the (a&1) branch of normal call a>>1 and of uniform call a>>1, so a wave never
// needs more than one Box-Muller evaluation per two attempts.  The loops are bounded (2^21
// attempts) so that every wave reaches its exit even on NaN input, where the reference would spin
// forever; with finite input the bound is unreachable in practice.
constexpr uint32_t kMaxAttemptPairs = 1u << 20;
//
// distribution/expcosdistribution.hh:51-65
__device__ __forceinline__ double expcos_draw(const RngKey &k, uint32_t site, double beta, double x_p,
                                              double x_m) {
  const double dx = x_m - x_p;
  const double tau = 2. * beta * fabs(cos(0.5 * dx));
  const double sigma = kPi * sqrt(2. / tau);
  const double inv4pi2 = 1. / (4. * kPi * kPi);
  double x = 0.0;
  for (uint32_t pair = 0; pair < kMaxAttemptPairs; ++pair) {
    double n0, n1, u0, u1;
    rng_normals(k, site, P_REJ_NORMAL, pair, n0, n1);
    rng_uniforms(k, site, P_REJ_UNIFORM, pair, u0, u1);
    x = sigma * n0;
    if (-kPi <= x && x < kPi && u0 <= exp(tau * (cos(x) - 1. + inv4pi2 * x * x))) break;
    x = sigma * n1;
    if (-kPi <= x && x < kPi && u1 <= exp(tau * (cos(x) - 1. + inv4pi2 * x * x))) break;
    x = 0.0;  // only reached when the bound is exhausted
  }
  return mod_2pi(x + 0.5 * (x_p + x_m) + (fabs(dx) > kPi ? kPi : 0.0));
}

// distribution/expsin2distribution.hh:45-58
__device__ __forceinline__ double expsin2_draw(const RngKey &k, uint32_t site, double sigma) {
  const double scale = kPi / sqrt(2. * sigma);
  for (uint32_t pair = 0; pair < kMaxAttemptPairs; ++pair) {
    double n0, n1, u0, u1;
    rng_normals(k, site, P_REJ_NORMAL, pair, n0, n1);
    rng_uniforms(k, site, P_REJ_UNIFORM, pair, u0, u1);
    double r = scale * n0;
    if (fabs(r) < kPi) {
      double s = sin(0.5 * r);
      if (u0 < exp(-sigma * (s * s - r * r / (kPi * kPi)))) return r;
    }
    r = scale * n1;
    if (fabs(r) < kPi) {
      double s = sin(0.5 * r);
      if (u1 < exp(-sigma * (s * s - r * r / (kPi * kPi)))) return r;
    }
  }
  return 0.0;
}

// ---- reductions -------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;  // valid in lane 0
}

// Sum over the workgroup in a fixed order (lane tree, then waves 0..n-1): bitwise reproducible.
// `scratch` needs blockDim.x/64 doubles per value.  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwave = (blockDim.x + kWave - 1) / kWave;
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) scratch[q * nwave + wave] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double s = 0.0;
      for (int w = 0; w < nwave; ++w) s += scratch[q * nwave + w];
      v[q] = s;
    }
  }
}

}  // namespace mlmcpi
