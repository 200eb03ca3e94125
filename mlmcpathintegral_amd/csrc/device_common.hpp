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

// p * x + c with the polynomial coefficient c in an SGPR pair (v_fma_f64 v, v, v, s[..]).  The compiler's own choice
// for fma(p, x, constant) is the two-address v_fmac_f64, which needs the constant copied into the destination VGPR
// pair first (one or two extra VALU instructions per Horner step); with the coefficient on the scalar side the step is
// a single VALU instruction and the copies become scalar moves, which issue beside the vector pipe.
__device__ __forceinline__ double fma_k(double p, double x, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c));
  return r;
}

// common/auxilliary.hh:42-44, operation for operation (used where the VALUE matters: QoIs)
__device__ __forceinline__ double mod_2pi(double x) { return x - 2. * kPi * floor(0.5 * (x + kPi) / kPi); }

// Same map with the division replaced by a multiplication with 1/(2 pi): 3 fp64 instructions instead
// of ~16 (an fp64 division is a v_rcp_f64 plus two Newton steps plus scale / fixup).  The two forms
// can differ only when (x + pi)/(2 pi) lies within an ulp of an integer, and then by exactly 2 pi,
// i.e. they return the same angle.  Used inside the sweeps, where link angles only ever enter
// 2 pi-periodic functions or another mod_2pi.
__device__ __forceinline__ double mod_2pi_fast(double x) {
  return fma(-kTwoPi, floor(fma(x, 1.0 / kTwoPi, 0.5)), x);  // 3 instructions: v_fma, v_floor, v_fma
}

// sin(d) for the leapfrog force of the rotor: two-term Cody-Waite reduction to |r| <= pi/2 (exact
// enough for |d| up to ~1e6, far beyond any angle difference a stable trajectory produces) and the
// degree-21 Taylor polynomial (truncation < 1.3e-18): ~20 fp64 instructions, against ~175 for the
// general sin() with its Payne-Hanek path.  Absolute error ~1e-16.
__device__ __forceinline__ double sin_reduced(double d) {
  const double n = rint(d * 0.31830988618379067154);           // d / pi
  double r = fma(-n, 3.14159265358979311600e+00, d);            // pi, high part
  r = fma(-n, 1.22464679914735317723e-16, r);                   // pi - high part
  const double r2 = r * r;
  double p = -1.9572941063391261231e-20;                        // -1/21!
  p = fma_k(p, r2, 8.2206352466243297170e-18);                    //  1/19!
  p = fma_k(p, r2, -2.8114572543455207632e-15);                   // -1/17!
  p = fma_k(p, r2, 7.6471637318198164759e-13);                    //  1/15!
  p = fma_k(p, r2, -1.6059043836821614599e-10);                   // -1/13!
  p = fma_k(p, r2, 2.5052108385441718775e-08);                    //  1/11!
  p = fma_k(p, r2, -2.7557319223985890653e-06);                   // -1/9!
  p = fma_k(p, r2, 1.9841269841269841270e-04);                    //  1/7!
  p = fma_k(p, r2, -8.3333333333333333333e-03);                   // -1/5!
  p = fma_k(p, r2, 1.6666666666666666667e-01);                    //  1/3!
  const double sr = fma(-r * r2, p, r);                          // r - r^3 (1/3! - r^2/5! + ...)
  // (-1)^n: n is an integer-valued double; its parity is the low bit of the converted integer
  return ((long long)n & 1) ? -sr : sr;
}

// ---- lean fp64 primitives for the heat-bath sampler ---------------------------------------------------
// The ocml division / sqrt / acos are correctly rounded over the whole double range (19 / 31 / ~95
// instructions).  The sampler's operands are benign (finite, far from the denormal range), so the
// scaling, fix-up and special-case code is dead weight; these versions keep the Newton / Goldschmidt
// cores only and stay within ~1 ulp.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

__device__ __forceinline__ double fast_div(double a, double b) {
  const double y = fast_rcp(b);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
}

__device__ __forceinline__ double fast_sqrt(double x) {  // x > 0, normal
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  return fma(fma(-g, g, x), h, g);
}

// acos on [-1, 1] after fdlibm's e_acos.c (rational approximation of (asin(x) - x) / x^3 in z),
// evaluated branch-free: z = x^2 for |x| < 1/2, z = (1 - |x|) / 2 otherwise.
__device__ __forceinline__ double fast_acos(double x) {
  const double ax = fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : 0.5 * (1.0 - ax);
  double p = 3.47933107596021167570e-05;
  p = fma_k(p, z, 7.91534994289814532176e-04);
  p = fma_k(p, z, -4.00555345006794114027e-02);
  p = fma_k(p, z, 2.01212532134862925881e-01);
  p = fma_k(p, z, -3.25565818622400915405e-01);
  p = fma_k(p, z, 1.66666666666666657415e-01);
  p *= z;
  double q = 7.70381505559019352791e-02;
  q = fma_k(q, z, -6.88283971605453293030e-01);
  q = fma_k(q, z, 2.02094576023350569471e+00);
  q = fma_k(q, z, -2.40339491173441421878e+00);
  q = fma_k(q, z, 1.0);
  const double R = fast_div(p, q);
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  // |x| < 1/2: pi/2 - (x + x R)
  const double r_small = pio2_hi - (x - (pio2_lo - x * R));
  // |x| >= 1/2: 2 (s + s R) for x > 0, pi - 2 (s + s R) for x < 0, s = sqrt(z)
  const double s = (z > 0.0) ? fast_sqrt(z) : 0.0;
  const double t = fma(s, R, s);
  const double r_large = (x > 0.0) ? 2.0 * t : 2.0 * (pio2_hi - (t - pio2_lo));
  return small ? r_small : r_large;
}

// log(x) for x in (0, 1] (the Box-Muller radius): x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m - 1)/(m + 1), |s| <= 0.1716: 10 odd terms (< 2e-17 truncation).  ~35
// instructions instead of ~100; relative accuracy also near x = 1, where log -> 0.
__device__ __forceinline__ double log_unit(double x) {
  int e;
  double m = frexp(x, &e);  // m in [0.5, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double s = fast_div(m - 1.0, m + 1.0);
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = fma_k(p, z, 1.0 / 17.0);
  p = fma_k(p, z, 1.0 / 15.0);
  p = fma_k(p, z, 1.0 / 13.0);
  p = fma_k(p, z, 1.0 / 11.0);
  p = fma_k(p, z, 1.0 / 9.0);
  p = fma_k(p, z, 1.0 / 7.0);
  p = fma_k(p, z, 1.0 / 5.0);
  p = fma_k(p, z, 1.0 / 3.0);
  p = fma_k(p, z, 1.0);
  return fma((double)e, 0.69314718055994530942, 2.0 * s * p);
}

// (cos, sin)(2 pi v) for v in [0, 1): quadrant from the nearest multiple of 1/4 turn, Taylor kernels on
// |x| <= pi/4, rotation by the quadrant.  ~35 instructions instead of ~190 for sincos().
__device__ __forceinline__ void sincos_2pi_unit(double v, double &sn, double &cs) {
  const double a = 4.0 * v;           // quarter turns, [0, 4)
  const double q = rint(a);           // 0..4
  const double x = (0.5 * kPi) * (a - q);  // |x| <= pi/4
  const double x2 = x * x;
  double sp = -7.6471637318198164759e-13;
  sp = fma_k(sp, x2, 1.6059043836821614599e-10);
  sp = fma_k(sp, x2, -2.5052108385441718775e-08);
  sp = fma_k(sp, x2, 2.7557319223985890653e-06);
  sp = fma_k(sp, x2, -1.9841269841269841270e-04);
  sp = fma_k(sp, x2, 8.3333333333333333333e-03);
  sp = fma_k(sp, x2, -1.6666666666666666667e-01);
  const double s0 = fma(x * x2, sp, x);
  double cp = 4.7794773323873852974e-14;
  cp = fma_k(cp, x2, -1.1470745597729724714e-11);
  cp = fma_k(cp, x2, 2.0876756987868098979e-09);
  cp = fma_k(cp, x2, -2.7557319223985890653e-07);
  cp = fma_k(cp, x2, 2.4801587301587301587e-05);
  cp = fma_k(cp, x2, -1.3888888888888888889e-03);
  cp = fma_k(cp, x2, 4.1666666666666666667e-02);
  cp = fma_k(cp, x2, -0.5);
  const double c0 = fma(cp, x2, 1.0);
  const int k = (int)q & 3;  // rotation by k quarter turns
  const double cr = (k & 1) ? -s0 : c0, sr = (k & 1) ? c0 : s0;
  cs = (k & 2) ? -cr : cr;
  sn = (k & 2) ? -sr : sr;
}

// Box-Muller from two uniforms in [0, 1): radius from 1 - u (in (0, 1]), angle 2 pi v
__device__ __forceinline__ void box_muller(double u, double v, double &n0, double &n1) {
  const double t = -2.0 * log_unit(1.0 - u);
  const double r = (t > 0.0) ? fast_sqrt(t) : 0.0;
  double sn, cs;
  sincos_2pi_unit(v, sn, cs);
  n0 = r * cs;
  n1 = r * sn;
}

// ---- RNG contract ------------------------------------------------------------------
enum Purpose : uint32_t {
  P_MOMENTUM = 1,
  P_ACCEPT = 2,
  P_GFF_NORMAL = 3,
  P_VONMISES = 4,
  P_INIT = 6,
  P_FILLIN = 7,   // two-level step: Gaussian fill-in of the fine-only sites
  P_ACCEPT2 = 8,  // two-level step: Metropolis uniform
  P_BESSEL = 9,   // two-level step, Schwinger coarsened in both directions: Bessel-product fill-in, sub = call counter
  P_EXACT = 10,   // exact Gaussian sampler of the harmonic oscillator: normals of entries (2 m, 2 m + 1) from site m
  P_GAUSSFILL = 13,  // two-level step, Schwinger coarsened in both directions, Gaussian fill-in: sub 0 (xi, omega), 1, 2 normals
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
    // one v_mad_u64_u32 per product (hi and lo together) instead of v_mul_hi_u32 + v_mul_lo_u32
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    c1 = (uint32_t)p1;
    c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// Round keys in VECTOR registers.  MI355X issues a 32-bit VALU instruction whose operands are all VGPRs (or literals) at
// 2.4 - 2.8 cycles per wave, the same instruction with an SGPR operand at 4.1 (tools/valu_issue_bench.hip,
// profiles/r04_valu_issue_cost.txt: v_xor_b32 2.4 / 4.1, v_bitop3_b32 2.8 / 4.1).  The compiler keeps wave-uniform values --
// the round keys -- in SGPRs, where the scalar unit bumps them for free, and pays for that in every one of the 2 x 10
// key xors of a call.  Here the keys of rounds 2 .. 9 sit in 16 VGPRs (an opaque v_mov, so that they stay there) and a
// round's  hi ^ counter ^ key  is ONE three-input v_bitop3_b32: 16 instructions at 2.8 cycles instead of 32 at 2.4 / 4.1.
// Rounds 0 and 1 stay as they are: half of their operands are uniform and fold away on the scalar side.  Same function.
struct PhiloxVKeys { uint32_t k[16]; };
__device__ __forceinline__ uint32_t to_vgpr(uint32_t s) {
  uint32_t v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
__device__ __forceinline__ PhiloxVKeys philox_vkeys(uint32_t k0, uint32_t k1) {
  PhiloxVKeys vk;
#pragma unroll
  for (int r = 2; r < 10; ++r) {
    vk.k[2 * (r - 2)] = to_vgpr(k0 + (uint32_t)r * 0x9E3779B9u);
    vk.k[2 * (r - 2) + 1] = to_vgpr(k1 + (uint32_t)r * 0xBB67AE85u);
  }
  return vk;
}
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                            const PhiloxVKeys &vk) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    if (r < 2) {
      c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
      c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    } else {
      c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, vk.k[2 * (r - 2)], 0x96);       // a ^ b ^ c
      c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, vk.k[2 * (r - 2) + 1], 0x96);
    }
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
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
  box_muller(u, v, n0, n1);
}

// the same with the round keys in vector registers (philox_vkeys: hot loops that draw once per cell)
__device__ __forceinline__ void rng_normals(const RngKey &k, const PhiloxVKeys &vk, uint32_t site, uint32_t purpose, uint32_t sub,
                                            double &n0, double &n1) {
  const U4 r = philox4x32_10(site, k.chain, k.step, (purpose << 24) | (sub & 0xFFFFFFu), k.k0, k.k1, vk);
  box_muller(u01(r.x, r.y), u01(r.z, r.w), n0, n1);
}

// cosine branch only (one normal per call)
__device__ __forceinline__ double rng_normal0(const RngKey &k, uint32_t site, uint32_t purpose, uint32_t sub) {
  double u, v, n0, n1;
  rng_uniforms(k, site, purpose, sub, u, v);
  box_muller(u, v, n0, n1);  // (the unused sine branch is dead code the compiler removes)
  return n0;
}

// ---- LDS reads that stay ds_read_b64 ---------------------------------------------------------------------
// hipcc merges neighbouring 8-byte LDS loads into ds_read2_b64, which the LDS serves at a quarter of the
// ds_read_b64 rate (MI355X_MICROARCH.md, LDS table: 16 cycles for 16 bytes per lane against 2 x 2).  The
// stencil kernels are LDS-issue bound, so their loads are issued through inline asm, which the merger
// does not see.  The caller issues a group of reads and then ONE lds_wait7() before the first use (the
// compiler does not track inline-asm loads, cdna_hip_programming.md 5.7).  `addr` is the LDS byte address
// (the callers add the LDS address of their dynamic array, see schwinger_or_kernel).
template <int OFF>
__device__ __forceinline__ double lds_read_f64(uint32_t addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// The wait names every loaded value as an in/out operand: the compiler sees the asm outputs of the reads
// as ready immediately and would otherwise schedule their consumers ABOVE the s_waitcnt.
__device__ __forceinline__ void lds_wait7(double &a, double &b, double &c, double &d, double &e, double &f, double &g) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g) : : "memory");
}

// the same for the six values of a heat-bath stencil, and for those of two cells at once
__device__ __forceinline__ void lds_wait6(double (&a)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : : "memory");
}
__device__ __forceinline__ void lds_wait12(double (&a)[6], double (&b)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(b[4]), "+v"(b[5])
               :
               : "memory");
}

// ---- heat-bath angle draws ------------------------------------------------------------------------
// Both heat-bath conditionals of the reference are von Mises laws p(x) ~ exp(kappa cos(x - c)):
//   ExpCosDistribution   kappa = tau = 2 beta |cos(dx/2)|      distribution/expcosdistribution.{hh:51-65,cc:7-21}
//   ExpSin2Distribution  kappa = sigma / 2                      distribution/expsin2distribution.{hh:45-58,cc:20-24}
// The reference draws them by rejection from a Gaussian envelope with acceptance rate
// sqrt(kappa/pi) I0(kappa) e^-kappa <= 0.27, -> 0 like sqrt(kappa) for flat conditionals: fine one
// site at a time on a CPU, but on a 64-wide wave the slowest lane sets the pace, and in a
// 1024^2 x batch sweep some link always has kappa ~ 1e-8 (~1e4 attempts) and stalls the whole
// launch.  The device samples the SAME distribution with the wrapped-Cauchy envelope of Best &
// Fisher (Appl. Statist. 28 (1979) 152-157): acceptance >= 0.65 for every kappa, one cosine per
// attempt, one arccosine per draw.
//
// Arithmetic.  Best & Fisher's envelope parameter r = (1 + rho^2) / (2 rho) simplifies to r = (1 + s) / (2 kappa) with
// s = sqrt(1 + 4 kappa^2); everything is written in R = kappa r = (1 + s) / 2 (one square root, no division, finite as
// kappa -> 0):   z = cos(pi u1),  f = cos(theta) = (kappa + R z) / (R + kappa z),  c = R - kappa f in (1/2, R + kappa],
// accept with probability c exp(1 - c).
//
// Random numbers.  ONE Philox call (counter word 3 = P_VONMISES << 24 | sub0 | t, t = 0, 1, ...) feeds TWO attempts,
// 2t from words (x, y) and 2t + 1 from (z, w).  Of the 64 bits v = hi:lo of an attempt
//     bits 12..63  u1 = (v >> 12) 2^-52           the proposal,
//     bit  0       the sign of the angle,
//     bits 1..11   b                               the leading 11 bits of the acceptance uniform u2 = (b + u2') / 2048.
// b alone decides the test unless c exp(1 - c) falls into [b, b + 1) / 2048 (about one attempt in 10^3); only then is
// the tail u2' (53 bits) taken from a second call (word 3 | kVmRefine), and the decision is the exact fp64 one
// (c (2 - c) > u2 or log(c / u2) + 1 - c >= 0).  The screening test runs in fp32 (hardware exp) with a guard band that
// covers the fp32 rounding, so whichever tier decides, the decision is the one the exact test would take.

// cos(pi u) for u in [0, 1]: cos(pi u) = sin(x), x = pi (1/2 - u), |x| <= pi/2; degree-21 Taylor polynomial
// (truncation < 1.3e-18).  15 fp64 instructions against ~175 for the general cos().
__device__ __forceinline__ double cospi_unit(double u) {
  const double x = kPi * (0.5 - u);
  const double x2 = x * x;
  double p = -1.9572941063391261231e-20;                        // -1/21!
  p = fma_k(p, x2, 8.2206352466243297170e-18);                    //  1/19!
  p = fma_k(p, x2, -2.8114572543455207632e-15);                   // -1/17!
  p = fma_k(p, x2, 7.6471637318198164759e-13);                    //  1/15!
  p = fma_k(p, x2, -1.6059043836821614599e-10);                   // -1/13!
  p = fma_k(p, x2, 2.5052108385441718775e-08);                    //  1/11!
  p = fma_k(p, x2, -2.7557319223985890653e-06);                   // -1/9!
  p = fma_k(p, x2, 1.9841269841269841270e-04);                    //  1/7!
  p = fma_k(p, x2, -8.3333333333333333333e-03);                   // -1/5!
  p = fma_k(p, x2, 1.6666666666666666667e-01);                    //  1/3!
  return fma(-x * x2, p, x);                                     // x - x^3 (1/3! - x^2/5! + ...)
}

// cos(d / 2) for any |d| < 2^30: d / (4 pi) reduced to t in [-1/2, 1/2], cos(2 pi t) = cos(pi * 2|t|)
__device__ __forceinline__ double cos_half(double d) {
  const double v = d * (0.25 / kPi);
  const double t = v - rint(v);
  return cospi_unit(2.0 * fabs(t));
}

// cos(x) for |x| < 2^30 (plaquette angles: |x| <= 4 pi): x / (2 pi) reduced to t in [-1/2, 1/2], cos(2 pi t) = cos(pi * 2|t|).
// Absolute error ~|x| 2e-16 + 1e-16: ~25 instructions against ~130 for the general cos() with its Payne-Hanek path.
__device__ __forceinline__ double cos_reduced(double x) {
  const double v = x * (0.5 / kPi);
  const double t = v - rint(v);
  return cospi_unit(2.0 * fabs(t));
}

// The sampler in three pieces so that callers can run the (divergent) attempt loop as a per-lane
// work queue: vm_envelope once per draw, vm_attempt_pair until it returns true, vm_angle once.
__device__ __forceinline__ double vm_clamp(double kappa) {
  return fmax(kappa, 1e-12);  // also maps NaN to a finite concentration: every wave reaches its exit
}

__device__ __forceinline__ double vm_envelope(double kappa) {  // R = kappa r = (1 + sqrt(1 + 4 kappa^2)) / 2
  return fma(0.5, fast_sqrt(fma(4. * kappa, kappa, 1.)), 0.5);
}

constexpr uint32_t kMaxVmPairs = 512u;     // attempt bound (2 x 512 attempts): every lane leaves the loop
constexpr uint32_t kVmFillin = 1u << 23;   // sub0 of the two-level fill-in draws (sweeps: 0)
constexpr uint32_t kVmRefine = 1u << 22;   // the call that supplies the tails u2' of a pair's acceptance uniforms

// proposal uniform: the top 52 bits of hi:lo as the mantissa of a double in [1, 2), minus 1
__device__ __forceinline__ double u01_52(uint32_t lo, uint32_t hi) {
  return __hiloint2double((int)((hi >> 12) | 0x3FF00000u), (int)__builtin_amdgcn_alignbit(hi, lo, 12)) - 1.0;
}

// One attempt from the word pair (lo, hi): proposal f = cos(theta), c, and the screening decision:
// 1 accepted, 0 rejected, -1 open (the 11 leading bits of u2 do not decide).
__device__ __forceinline__ int vm_try(uint32_t lo, uint32_t hi, double kappa, double R, double &f, double &c) {
  const double z = cospi_unit(u01_52(lo, hi));
  f = fast_div(fma(R, z, kappa), fma(kappa, z, R));
  c = fma(-kappa, f, R);
  const float cf = (float)c;
  const float af = cf * __expf(1.0f - cf);              // acceptance probability c exp(1 - c), fp32
  const float band = af * (1e-5f * (1.0f + cf));        // >> its fp32 error (~4e-7 (1 + c) relative)
  const float lo_s = (float)((lo >> 1) & 0x7FFu) * (1.0f / 2048.0f), hi_s = lo_s + (1.0f / 2048.0f);  // u2 in [lo_s, hi_s)
  return hi_s <= af - band ? 1 : (lo_s >= af + band ? 0 : -1);
}

// the exact test with the full acceptance uniform u2 = (b + tail) / 2048
__device__ __forceinline__ int vm_exact(uint32_t lo, double tail, double c) {
  const double u2 = ((double)((lo >> 1) & 0x7FFu) + tail) * (1.0 / 2048.0);
  return (c * (2. - c) - u2 > 0. || log(c / u2) + 1. - c >= 0.) ? 1 : 0;
}

// Attempts 2 pair and 2 pair + 1; returns true when one of them is accepted (or when the attempt bound is hit).
// f = cos(theta).  sub0 separates streams that share (site, chain, step): 0 for sweeps, kVmFillin for two-level fill-ins.
template <class Keys>   // RngKey alone, or RngKey + PhiloxVKeys (hot loops)
__device__ __forceinline__ bool vm_attempt_pair_impl(const RngKey &k, const Keys *vk, uint32_t site, uint32_t pair, double kappa, double R,
                                                     double &f, bool &negative, uint32_t sub0) {
  const uint32_t w3 = (P_VONMISES << 24) | sub0 | pair;
  const U4 q = vk ? philox4x32_10(site, k.chain, k.step, w3, k.k0, k.k1, *vk) : philox4x32_10(site, k.chain, k.step, w3, k.k0, k.k1);
  double fa, ca, fb, cb;
  int sa = vm_try(q.x, q.y, kappa, R, fa, ca), sb = vm_try(q.z, q.w, kappa, R, fb, cb);
  if (sa < 0 || (sa == 0 && sb < 0)) {  // a decision that matters is open: fetch the tails
    const U4 e = philox4x32_10(site, k.chain, k.step, w3 | kVmRefine, k.k0, k.k1);
    if (sa < 0) sa = vm_exact(q.x, u01(e.x, e.y), ca);
    if (sa == 0 && sb < 0) sb = vm_exact(q.z, u01(e.z, e.w), cb);
  }
  f = sa == 1 ? fa : fb;
  negative = ((sa == 1 ? q.x : q.z) & 1u) != 0;
  return sa == 1 || sb == 1 || pair + 1 >= kMaxVmPairs;
}
__device__ __forceinline__ bool vm_attempt_pair(const RngKey &k, uint32_t site, uint32_t pair, double kappa, double R,
                                                double &f, bool &negative, uint32_t sub0 = 0) {
  return vm_attempt_pair_impl<PhiloxVKeys>(k, nullptr, site, pair, kappa, R, f, negative, sub0);
}
__device__ __forceinline__ bool vm_attempt_pair(const RngKey &k, const PhiloxVKeys *vk, uint32_t site, uint32_t pair, double kappa,
                                                double R, double &f, bool &negative, uint32_t sub0 = 0) {
  return vm_attempt_pair_impl<PhiloxVKeys>(k, vk, site, pair, kappa, R, f, negative, sub0);
}

__device__ __forceinline__ double vm_angle(double f, bool negative) {
  const double theta = fast_acos(fmin(1.0, fmax(-1.0, f)));
  return negative ? -theta : theta;
}

__device__ __forceinline__ double vonmises_draw(const RngKey &k, uint32_t site, double kappa, uint32_t sub0 = 0) {
  kappa = vm_clamp(kappa);
  const double R = vm_envelope(kappa);
  double f = 1.0;
  bool negative = false;
  for (uint32_t pair = 0; !vm_attempt_pair(k, site, pair, kappa, R, f, negative, sub0); ++pair) {
  }
  return vm_angle(f, negative);
}

// quenchedschwingeraction.cc:46-54 -> expcosdistribution.hh:51-65: the conditional of a link between staple angles
// x_p, x_m is exp(beta [cos(x - x_p) + cos(x - x_m)]) = exp(2 beta cos((x_m - x_p)/2) cos(x - (x_p + x_m)/2)): a von Mises
// law around the mean staple angle, shifted by pi when the cosine is negative.  The identity holds for any real
// x_p, x_m, so the staple sums need no mod_2pi of their own (the reference wraps them and tests |dx| > pi; same angle).
__device__ __forceinline__ void expcos_params(double beta, double x_p, double x_m, double &tau, double &centre) {
  const double ch = cos_half(x_m - x_p);
  tau = 2. * beta * fabs(ch);
  centre = fma(0.5, x_p + x_m, ch < 0.0 ? kPi : 0.0);
}

__device__ __forceinline__ double expcos_draw(const RngKey &k, uint32_t site, double beta, double x_p,
                                              double x_m, uint32_t sub0 = 0) {
  double tau, centre;
  expcos_params(beta, x_p, x_m, tau, centre);
  return mod_2pi_fast(vonmises_draw(k, site, tau, sub0) + centre);
}

// rotoraction.cc:20-37 -> expsin2distribution.hh:45-58
__device__ __forceinline__ double expsin2_draw(const RngKey &k, uint32_t site, double sigma) {
  return vonmises_draw(k, site, 0.5 * sigma);
}

// exp(-z) I0(z), z >= 0 (the normalisation of the rotor's conditioned fine action): power series in
// z^2/4 for z < 30 (all terms positive: no cancellation), Hankel asymptotic series beyond.  Relative
// accuracy ~1e-15.  The reference calls gsl_sf_bessel_I0_scaled (expsin2distribution.cc:7-17).
__device__ __forceinline__ double bessel_i0_scaled(double z) {
  if (z < 30.0) {
    const double q = 0.25 * z * z;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 120; ++k) {
      term *= q / ((double)k * (double)k);
      sum += term;
      if (term < 1e-17 * sum) break;
    }
    return exp(-z) * sum;
  }
  const double w = 1.0 / (8.0 * z);
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 30; ++k) {
    const double odd = 2.0 * k - 1.0;
    term *= odd * odd * w / (double)k;
    sum += term;
    if (term < 1e-17 * sum) break;
  }
  return sum / sqrt(kTwoPi * z);
}

// ExpSin2Distribution::fast_2pi_I0_scaled (expsin2distribution.cc:7-17), including the reference's
// three-term expansion for z > 100
__device__ __forceinline__ double two_pi_i0_scaled(double z) {
  if (z > 100.) {
    const double zi = 1. / z;
    return sqrt(2. * kPi * zi) * (1. + 0.125 * zi + 0.0703125 * zi * zi);
  }
  return 2. * kPi * bessel_i0_scaled(z);
}

// Heat-bath colour phase.  Each thread owns up to S cells of the region (linear index tid + NT m).  Their conditional
// parameters are set up first (no divergence) and every cell gets its first PAIR of attempts (one Philox call) in
// straight-line code; about 97 % of the cells are done then.  What is left is a geometric tail: a few cells per wave
// that need one more call, a few per workgroup that need two.  Retrying them where they sit makes every wave run the
// whole attempt code with one or two live lanes, several times over.  Instead the leftovers of the whole workgroup are
// pushed into a small LDS pool (HbPool: concentration, centre, Philox site, LDS offset), and after a barrier the first
// threads of the workgroup finish them, one entry each, and write the angles straight to their cells.  Entries that do not fit (the pool
// holds `cap` of them; expected ~40 per 1280 cells at beta = 1) are retried by their own lane on the spot.  Cells
// accepted at once are written back at once (cells of one colour phase are not in each other's stencils), so nothing
// but the loop state lives across cells.  Which random numbers a cell consumes is fixed by (site, attempt), so the
// result does not depend on any of this scheduling.
struct HbPool {
  double *base;            // kap[cap] | cen[cap] | site[cap] | off[cap] | count[2]
  uint32_t cap, use;       // capacity (0: no pool); number of uses so far (uniform over the workgroup)
  // pushes of use u go to count[u & 1]; the other counter is cleared meanwhile
  static __host__ __device__ constexpr size_t bytes(uint32_t cap) { return (size_t)cap * 24 + 8; }
  __device__ double *kap() const { return base; }
  __device__ double *cen() const { return base + cap; }
  __device__ uint32_t *site() const { return (uint32_t *)(base + 2 * cap); }
  __device__ uint32_t *off() const { return (uint32_t *)(base + 2 * cap) + cap; }
  __device__ uint32_t *count() const { return (uint32_t *)(base + 2 * cap) + 2 * cap; }
  __device__ static HbPool carve(double *lds, uint32_t cap) {  // call from every thread; thread 0 clears the counters
    HbPool p{lds, cap, 0u};
    if (cap && threadIdx.x == 0) p.count()[0] = p.count()[1] = 0;  // visible after the caller's next barrier
    return p;
  }
};

template <int NT, int S, bool LEAN = false, class Setup, class Commit>   // LEAN: round keys on the scalar side (16 VGPRs less)
__device__ __forceinline__ void heatbath_cells(uint32_t total, const RngKey &key, HbPool &pool, Setup setup, Commit commit) {
  PhiloxVKeys vk_;
  if (!LEAN) vk_ = philox_vkeys(key.k0, key.k1);
  const PhiloxVKeys *const vk = LEAN ? nullptr : &vk_;
  for (uint32_t b0 = 0; b0 < total; b0 += S * NT) {  // uniform trip count: the barriers below need every thread
    uint32_t *cnt = pool.count() + (pool.use & 1u);
    // The other counter (the one of the previous and of the next use) is cleared HERE.  Invariant: a WORKGROUP BARRIER
    // separates the drain of use u from the start of use u + 1 -- every thread takes part in a drain, so program order in
    // thread 0 alone would not do.  Within a call that barrier is the one at the end of the b0 loop below; between two
    // calls it is the caller's barrier between colour phases (every call site has one: the next phase reads what this
    // one wrote).  The clear therefore follows every read of this counter in the previous use's drain, and precedes the
    // barrier of this use, which every push of the next use follows.
    if (pool.cap && threadIdx.x == 0) pool.count()[(pool.use + 1u) & 1u] = 0;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const uint32_t idx = b0 + m * NT + threadIdx.x;
      if (idx < total) {
        double tau, cen, f = 1.0;
        uint32_t site, off;
        bool neg = false;
        setup(idx, tau, cen, site, off);
        const double kap = vm_clamp(tau), env = vm_envelope(kap);
        bool done = vm_attempt_pair(key, vk, site, 0, kap, env, f, neg);
        if (!done && pool.cap) {
          const uint32_t slot = atomicAdd(cnt, 1u);
          if (slot < pool.cap) {
            pool.kap()[slot] = kap; pool.cen()[slot] = cen; pool.site()[slot] = site; pool.off()[slot] = off;
            continue;  // finished after the barrier, by whichever thread takes the entry
          }
        }
        // no pool, or pool full: retry here
        for (uint32_t pair = 1; !done; ++pair) done = vm_attempt_pair(key, vk, site, pair, kap, env, f, neg);
        commit(off, mod_2pi_fast(vm_angle(f, neg) + cen));
      }
    }
    if (pool.cap) {
      __syncthreads();
      {  // the first threads finish the pooled cells, one each (r03: wave 0 alone, 64 at a time)
        const uint32_t filled = min(*cnt, pool.cap);
        for (uint32_t e = threadIdx.x; e < filled; e += NT) {
          const double k_ = pool.kap()[e], c_ = pool.cen()[e], r_ = vm_envelope(k_);
          const uint32_t s_ = pool.site()[e], o_ = pool.off()[e];
          double f = 1.0;
          bool ng = false;
          for (uint32_t pair = 1; !vm_attempt_pair(key, vk, s_, pair, k_, r_, f, ng); ++pair) {
          }
          commit(o_, mod_2pi_fast(vm_angle(f, ng) + c_));
        }
      }
      ++pool.use;
      // Another pass of this phase follows (more than S NT cells: tiles larger than the default): its pushes reuse the
      // entry arrays the drain above is still reading, so it waits.  (Between two phases the caller's own barrier does that.)
      if (b0 + S * NT < total) __syncthreads();
    }
  }
}

// ---- heat-bath angle draws, tabulated step envelope (launches whose concentrations stay below kVsKappaMax) ----------
// The wrapped-Cauchy sampler above pays, per draw, an envelope square root, a division and an fp64 cosine per attempt
// and an arccosine for the accepted one: ~300 fp64-class instructions of a ~520-instruction cell (388 lane
// instructions per link update, the heat bath = half the step).  For moderate concentrations -- the Schwinger model at
// beta <= 8, the rotor at 2 m0 / a <= 16 (kVsKappaMax): every BASELINE configuration -- the same von Mises law p(x) ~ exp(kappa cos x)
// is drawn here from a piecewise-constant envelope taken from a small table, so that an attempt costs no fp64
// arithmetic at all and the accepted angle is a linear function of random bits:
//     bins of |x|     edges (0, 1, 2, 3, 4, 6, 8, 12, 16) pi/16: eight bins, finer where the density is high;
//     bin k is proposed with probability q_k / 64 (six random bits through a 64-entry selector), |x| uniform inside it
//                     (35 random bits), sign from one more bit;
//     accepted with probability  exp(kappa (cos x - 1)) 2^lw[k],   2^lw[k] = (w_k / q_k) / max_j (H_j w_j / q_j),
// i.e. target / (proposal density x envelope constant), H_j = the target at the left edge of bin j (its maximum there).
// The q_k follow the shape of the target, which depends on kappa; so there are kVsClasses tables, for eight ranges of
// kappa, and a cell picks its table by an exact fp64 comparison: with t = |(x_m - x_p)/(4 pi)| reduced to [0, 1/2],
// kappa = scale |cos(2 pi t)| = scale sin(2 pi v), v = |t - 1/4| in [0, 1/4], class = floor(32 v): the table of class c
// is built for kappa_min = scale sin(2 pi c / 32) and is a valid envelope for every larger kappa (the normalised target
// only gets narrower).  Acceptance 0.70 ... 0.89 (0.79 on average; wrapped Cauchy: 0.82).  The tables are built by the
// host for the action's scale (runtime.hip, vs_build_tables; exported as mlmcpi_vs_table so that the oracle's own
// construction can be compared with it) and copied to LDS by the kernels: 512 selector bytes + 8 x 16 floats.
// The test is screened in fp32 (24 bits of |x|, one polynomial cosine, v_exp_f32) against the kVsU2Bits = 22 leading bits of
// u2 with a guard band that covers every fp32 rounding on the way; when the band does not decide (one attempt in ~3 10^4),
// the exact fp64 test with the full u2 takes over, as for the wrapped-Cauchy sampler, so the decision is always the one of
// the exact test -- which is what the oracle computes (oracle.cc, dev_vonmises_table).  (With 11 leading bits, as the
// wrapped-Cauchy sampler has them, one attempt in 2048 was open and 4-6 % of the wave iterations took the detour through
// the exact test -- a Philox call, an fp64 logarithm and cosine behind a function call: measured at 8 % of the fused
// sweep launch.)  Random numbers: the same Philox calls (word pair = one attempt), fields: bit 0 sign, bits 1..22 leading
// bits of u2, bits 23..57 position inside the bin (35 bits: 2 10^-11 rad), bits 58..63 selector.
// host rule (lattice2d.hip / path1d.hip; the oracle applies the same): this sampler where the action's largest concentration,
// 2 beta or 2 m0 / a, is <= kVsKappaMax; the wrapped-Cauchy sampler beyond.  r05: 16 (4 until then).  The eight classes split
// [0, scale] by sin(2 pi c / 32), so they get wider in kappa with the scale and a table, built for the smallest
// concentration of its class, fits the largest one less well: acceptance per attempt 0.78 on average at scale 2, 0.74 at 4,
// 0.67 at 8, 0.59 at 12, 0.58 at 16 (0.24 at worst), a pair fails for 5 / 7 / 11 / 18 / 20 % of the cells.  Even so a draw of
// the fused Schwinger launch (1024^2 x 32) takes 0.70 ms at beta = 4, 0.82 at 6 and 0.75 at 8 against 1.18 - 1.19 ms with the
// wrapped-Cauchy sampler; beyond 16 the eight bins (the finest pi / 16 wide) are too coarse for the target.  The error
// budget of the screening test below scales with kappa' = scale log2 e through the band, and |log2 a| <= 2 kappa' + |lw| <= 55
// is far from fp32's exponent range.
constexpr double kVsKappaMax = 16.0;
constexpr int kVsClasses = 8, kVsBins = 8, kVsSel = 64;
constexpr uint32_t kVsU2Bits = 22;  // leading bits of the acceptance uniform carried by the attempt itself
// The 64 bits (lo, hi) of an attempt (r04 layout: every field is one shift or one conversion away from its use):
//   lo[31..10]  b, the 22 leading bits of the acceptance uniform u2 = (b + tail) / 2^22 (tail: the refine call, rarely);
//               (float)lo is u2 2^32 to within 2^10, which is all the screening test needs
//   lo[9]       sign of the angle
//   lo[8..0]    the low 9 of the 35 position bits
//   hi[31..26]  bin selector
//   hi[25..0]   the high 26 position bits; (float)(hi << 6) is the position 2^32, rounded to 24 bits
// Device image of the tables (bytes; built by the host for the action's scale, runtime.hip vs_device_image):
//   code[class][64]   1 B   8 x (bin of the selector value): a byte offset into the next two tables
//   scr[bin]          8 B   float2 {pi/2 - edge, -width 2^-32}: y = pi/2 - |x| = fma(scr.y, (float)(hi << 6), scr.x), cos|x| = sin y
//   lw[class][bin]    8 B   float log2 of the bin's acceptance factor (+ 4 B of padding: lw and code share the class offset 64 cls)
//   fin[bin]          16 B  double2 {64 width, edge - 64 width} in units of pi/16: |x| = (pi/16) fma(fin.x, m, fin.y), m = 1 + position / 64
//   consts            16 B  float {s_acc, s_rej, 0, 0}: thresholds of the screening test for this scale (below)
constexpr uint32_t kVsCodeOff = 0, kVsScrOff = 512, kVsLwOff = 576, kVsFinOff = 1088, kVsConstOff = 1216, kVsTableBytes = 1232;
// host side of the same encoding: left edge and width of bin k in units of pi/16 (edges 0 1 2 3 4 6 8 12, widths 1 1 1 1 2 2 4 4)
__host__ __device__ constexpr uint32_t vs_edge16(uint32_t k) { return (0xC8643210u >> (4u * k)) & 15u; }
__host__ __device__ constexpr uint32_t vs_width16(uint32_t k) { return (0x44221111u >> (4u * k)) & 15u; }
// Screening test.  a = the acceptance probability in fp32 (error budget below), L = (float)lo.  The exact u2 2^32 lies within
// 2^10 (1 + 2^-22) of L, so
//     accepted for sure   L <= a 2^32 (1 - band) - kVsU2Slack,       rejected for sure   L >= a 2^32 (1 + band) + kVsU2Slack;
// neither: the exact fp64 test with the full u2 decides (one attempt in ~10^4).  Error budget of log2(a): |x| to 24 bits
// (1e-7 kappa'), the sine polynomial (1.2e-7 kappa'), kappa' itself (2e-7 kappa' from the fp32 cosine behind it), the fma (1e-6
// at |log2 a| <= 16): < 4e-7 (1 + kappa'), i.e. < 3e-7 (1 + kappa') relative on a, plus v_exp_f32's own ~2e-7 and 2^-23 for
// the conversion of lo and the threshold's own rounding.  The band is 1e-5 (1 + kappa'_max) for the whole launch (kappa'_max =
// scale log2 e): more than 20 times that, and two constants (s_acc = 2^32 (1 - band), s_rej = 2^32 (1 + band)) instead of
// five instructions per cell.
constexpr float kVsU2Slack = 1100.0f;
__host__ __device__ inline float vs_band_of_scale(double scale) { return (float)(1e-5 * (1.0 + scale * 1.4426950408889634)); }

struct VsTable {
  const uint8_t *base;   // LDS in the sweeps; global memory in the site-at-a-time kernels and the test hook
  float s_acc, s_rej;
  __device__ static VsTable at(const void *image, const uint32_t *__restrict__ d_table) {
    VsTable t{(const uint8_t *)image, 0.f, 0.f};
    if (d_table) {  // uniform address: scalar loads; then into vector registers, where a v_fma_f32 on them issues at full rate
      t.s_acc = __uint_as_float(to_vgpr(d_table[kVsConstOff / 4]));
      t.s_rej = __uint_as_float(to_vgpr(d_table[kVsConstOff / 4 + 1]));
    }
    return t;
  }
  // the workgroup copies the action's table from global memory (visible after the caller's next barrier)
  __device__ static VsTable stage(void *lds, const uint32_t *__restrict__ d_table) {
    if (d_table)
      for (uint32_t i = threadIdx.x; i < kVsTableBytes / 4; i += blockDim.x) ((uint32_t *)lds)[i] = d_table[i];
    return at(lds, d_table);
  }
  __device__ static VsTable in_global(const uint32_t *__restrict__ d_table) { return at(d_table, d_table); }
};

// sin(y), |y| <= pi/2, fp32: odd minimax polynomial of degree 9 (approximation error 4.6e-9; 1.2e-7 with the fp32 rounding,
// measured over 2 10^6 arguments -- the degree-11 Taylor polynomial this replaces: 1.6e-7)
__device__ __forceinline__ float sinf_half_pi(float y) {
  const float y2 = y * y;
  float p = 2.600053086e-06f;
  p = fmaf(p, y2, -1.980661437e-04f);
  p = fmaf(p, y2, 8.333017279e-03f);
  p = fmaf(p, y2, -1.666665710e-01f);
  return fmaf(y * y2, p, y);
}

struct VsCell {
  double centre;   // the draw is mod_2pi(centre +- |x|)
  float kp;        // kappa log2(e), fp32: screening only
  uint32_t cls;    // 64 x (which table): the byte offset of the class in code[][] and lw[][]
  uint32_t site;   // Philox counter word 0
};

// Cell set-up shared by the Schwinger links (x_p, x_m = the two staple sums, scale = 2 beta) and the rotor sites
// (x_p, x_m = the neighbours, scale = 2 m0 / a): conditional exp(scale/2 [cos(x - x_p) + cos(x - x_m)])
// = exp(kappa cos(x - centre)), kappa = scale |cos((x_m - x_p)/2)|, centre = (x_p + x_m)/2 (+ pi where the cosine is
// negative, i.e. for t > 1/4: an exact fp64 comparison); kappa itself is needed in fp32 only.
__device__ __forceinline__ void vs_cell(double scale, double x_p, double x_m, VsCell &c) {
  const double v = (x_m - x_p) * (0.25 / kPi);
  const double t = fabs(v - rint(v));
  c.centre = fma(0.5, x_p + x_m, t > 0.25 ? kPi : 0.0);
  const double w = fabs(t - 0.25);                                          // [0, 1/4]; NaN for a NaN state
  c.cls = min((uint32_t)(32.0 * w), (uint32_t)(kVsClasses - 1)) * kVsSel;  // (uint32_t)NaN = 0
  // |cos(2 pi t)| = |sin(pi/2 - 2 pi t)|, t in [0, 1/2]
  c.kp = fmaxf((float)scale * fabsf(sinf_half_pi(fmaf(-6.28318531f, (float)t, 1.57079633f))), 0.0f) * 1.44269504f;   // fmaxf: NaN -> 0
}
__device__ __forceinline__ double vs_kappa_exact(double scale, double x_p, double x_m) {
  return vm_clamp(scale * fabs(cos_half(x_m - x_p)));
}

// |x| of the attempt (lo, hi), fp64: (pi/16) (edge + width pos), pos = the 35 position bits / 2^35; code = 8 x the bin
__device__ __forceinline__ double vs_theta(uint32_t lo, uint32_t hi, uint32_t code, const VsTable &tab) {
  // mantissa = 000000 | 26 bits of hi | 9 bits of lo | 11 zeros: m = 1 + pos / 64
  const double m = __hiloint2double((int)(((hi >> 12) & 0x3FFFu) | 0x3FF00000u), (int)__builtin_amdgcn_alignbit(hi, lo << 23, 12));
  const double2 f = *(const double2 *)(tab.base + kVsFinOff + 2u * code);
  return (kPi / 16.0) * fma(f.x, m, f.y);   // edge + width pos is exact in fp64: one rounding, as in the oracle
}

// acceptance probability of one attempt in fp32; the bin's code comes back for the caller
__device__ __forceinline__ float vs_accept_prob(uint32_t hi, float kp, uint32_t cls, const VsTable &tab, uint32_t &code) {
  const uint8_t *row = tab.base + cls;   // the class's rows of code[][] and (kVsLwOff further on) of lw[][]
  code = row[hi >> 26];
  const float2 scr = *(const float2 *)(tab.base + kVsScrOff + code);
  const float sn = sinf_half_pi(fmaf(scr.y, (float)(hi << 6), scr.x));   // cos|x|
  return __builtin_amdgcn_exp2f(fmaf(kp, sn - 1.0f, *(const float *)(row + kVsLwOff + code)));
}

// the exact test: u2 = (b + tail) / 2^22 against exp(kappa (cos x - 1)) 2^lw, in logarithms
__device__ __forceinline__ int vs_exact(uint32_t lo, uint32_t hi, uint32_t code, double tail, double kappa, float lw, const VsTable &tab) {
  const double u2 = ((double)(lo >> (32 - kVsU2Bits)) + tail) * (1.0 / (double)(1u << kVsU2Bits));
  const double la = fma(kappa, cospi_unit(vs_theta(lo, hi, code, tab) * (1.0 / kPi)) - 1.0, 0.69314718055994531 * (double)lw);
  return (u2 <= 0.0 || log_unit(u2) <= la) ? 1 : 0;
}

// The open decisions of a pair of attempts, taken exactly (one attempt in ~10^4 gets here).  Not inlined: the fp64
// polynomial coefficients of this path would otherwise be hoisted into scalar registers for the whole kernel and push
// the hot loop's own constants out (the hot loop then reloads them with v_readlane every iteration).
__device__ __attribute__((noinline)) uint32_t vs_exact_pair(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t step, uint32_t site,
                                                            uint32_t w3, U4 q, uint32_t codes, float lwa, float lwb, double kappa,
                                                            int sa, int sb, const uint8_t *tab_base) {
  const VsTable tab{tab_base, 0.f, 0.f};
  const U4 e = philox4x32_10(site, chain, step, w3 | kVmRefine, k0, k1);
  if (sa < 0) sa = vs_exact(q.x, q.y, codes & 0xFFu, u01(e.x, e.y), kappa, lwa, tab);
  if (sa == 0 && sb < 0) sb = vs_exact(q.z, q.w, codes >> 8, u01(e.z, e.w), kappa, lwb, tab);
  return (uint32_t)(sa & 3) | ((uint32_t)(sb & 3) << 2);   // two's complement in two bits each: 3 = open (cannot remain), 1, 0
}

// attempts 2 pair and 2 pair + 1 of `site`; kappa_exact() is evaluated only when a screening decision is open.
// Returns true when one of them is accepted (or the attempt bound is hit); |x| and its sign come back either way.
template <class KappaExact>
__device__ __forceinline__ bool vs_attempt_pair(const RngKey &k, const PhiloxVKeys *vk, uint32_t site, uint32_t pair, float kp, uint32_t cls,
                                                const VsTable &tab, KappaExact kappa_exact, double &theta, bool &negative,
                                                uint32_t sub0 = 0) {
  const uint32_t w3 = (P_VONMISES << 24) | sub0 | pair;
  // vk == NULL (a compile-time fact at every call site): round keys on the scalar side -- 16 VGPRs less, for kernels that
  // live on occupancy
  const U4 q = vk ? philox4x32_10(site, k.chain, k.step, w3, k.k0, k.k1, *vk) : philox4x32_10(site, k.chain, k.step, w3, k.k0, k.k1);
  uint32_t ca, cb;
  const float pa = vs_accept_prob(q.y, kp, cls, tab, ca), pb = vs_accept_prob(q.w, kp, cls, tab, cb);
  const float la = (float)q.x, lb = (float)q.z;
  bool acc_a = la <= fmaf(pa, tab.s_acc, -kVsU2Slack), acc_b = lb <= fmaf(pb, tab.s_acc, -kVsU2Slack);  // lane masks
  if (!acc_a) {
    const bool rej_a = la >= fmaf(pa, tab.s_rej, kVsU2Slack), rej_b = lb >= fmaf(pb, tab.s_rej, kVsU2Slack);
    // open: the first attempt undecided, or rejected and the second undecided
    if (!rej_a || (!acc_b && !rej_b)) {
      const uint8_t *row = tab.base + cls + kVsLwOff;
      const uint32_t r = vs_exact_pair(k.k0, k.k1, k.chain, k.step, site, w3, q, ca | (cb << 8), *(const float *)(row + ca),
                                       *(const float *)(row + cb), kappa_exact(), rej_a ? 0 : -1, acc_b ? 1 : rej_b ? 0 : -1, tab.base);
      acc_a = (r & 3u) == 1u;
      acc_b = (r >> 2) == 1u;
    }
  }
  const bool first = acc_a;
  const uint32_t lo = first ? q.x : q.z, hi = first ? q.y : q.w;
  theta = vs_theta(lo, hi, first ? ca : cb, tab);
  negative = (lo & 0x200u) != 0;
  return acc_a || acc_b || pair + 1 >= kMaxVmPairs;
}

// Open cells of a step-envelope colour phase.  After the first pair of attempts ~5 % of the cells are still open.  Retrying
// them where they sit makes every wave run the whole attempt code again and again with two or three live lanes.  Instead
// they go on a list in LDS -- an entry is just the cell's LDS offset and the index of its next pair; stencil, centre and
// kappa are recomputed from the tile image, which the phase does not change under the cell -- and behind ONE barrier the
// first threads of the workgroup take one entry each and finish it where it is, however many pairs that takes.
// (r03 worked the list off in rounds -- one pair per entry, what is still open onto a second list, a barrier, and so on
// until a round fitted one wave: with ~110 entries per phase that was a barrier and a round more than this, and 7 % of
// the fused launch.  Measured on one box, r04: 0.8175 ms with rounds down to 64 entries, 0.7863 down to 128, 0.7626 with
// none, profiles/r04_ab_tail.txt.  Handing the cells of a phase out through a queue, so that retries ride along with
// fresh cells and no list is needed until the queue runs dry, was built and measured too: 3 % slower than the rounds --
// the returning atomic and the hand-out arithmetic per wave and iteration cost more than the rounds they replace.)
template <class E>  // uint16_t: offset in 12 bits, next pair in 4 (compile-time 64 x 32 tiles); uint32_t: 16 + 16
struct VsPool {
  static constexpr uint32_t kOffBits = sizeof(E) == 2 ? 12 : 16;
  static constexpr uint32_t kMaxPair = (1u << (8 * sizeof(E) - kOffBits)) - 1;
  E *buf;                // `cap` entries
  uint32_t *count;       // [2]: list lengths of even / odd uses
  uint32_t cap, use;     // capacity (0: no list, every cell is finished by its own lane); uses so far (uniform)
  VsTable tab;
  // LDS bytes: table | counters | list
  static __host__ __device__ constexpr size_t bytes(uint32_t cap) { return kVsTableBytes + 16 + ((size_t)cap * sizeof(E) + 7) / 8 * 8; }
  // call from every thread; visible after the caller's next barrier.  d_table == NULL: a kernel instance that never draws
  // from the step envelope -- nothing in LDS is touched.
  __device__ static VsPool carve(double *lds, uint32_t cap, const uint32_t *d_table) {
    VsPool p;
    p.tab = VsTable::stage(lds, d_table);
    p.count = (uint32_t *)((uint8_t *)lds + kVsTableBytes);
    p.buf = (E *)(p.count + 4);
    p.cap = cap;
    p.use = 0;
    if (d_table && threadIdx.x < 2) p.count[threadIdx.x] = 0;
    return p;
  }
};

// One colour phase.  off_of(idx) = LDS offset of cell idx of the phase; setup(off, cell) = centre, class, kappa', Philox
// site from the tile image; kappa_exact(off) = the fp64 concentration (rare); commit(off, angle).  The caller puts a
// barrier behind the call (every call site has one: the next phase reads what this one wrote).
// Pass 0 walks the cells of the phase (thread t: cells t, t + NT, ...), pass 1 the list (the rare exact test is a function
// call, vs_exact_pair, so that the two copies of the attempt code stay small).  S (cells per thread and pass of
// heatbath_cells, whose signature the call sites share) is not used here.
// LEAN (the 1-D rotor sweeps, whose 256-thread workgroups live on occupancy): one copy of the cell code with a run-time flag and
// the round keys on the scalar side -- 59 instead of 97 VGPRs, 7 instead of 4 waves per SIMD (the fast form cost the rotor
// sweeps 10 %; it gains the fused Schwinger launch, which LDS holds at 4 waves per SIMD anyway, 3 %).
template <int NT, int S, class E, int LEAN = 0, class OffOf, class Setup, class KappaExact, class Commit>   // LEAN: 1 = scalar round keys, 2 = + one copy of the cell code
__device__ __forceinline__ void heatbath_cells_step(uint32_t total, const RngKey &key, VsPool<E> &pool, OffOf off_of, Setup setup,
                                                    KappaExact kappa_exact, Commit commit) {
  using P = VsPool<E>;
  constexpr uint32_t kOffMask = (1u << P::kOffBits) - 1;
  uint32_t *const cnt = pool.count + (pool.use & 1u);
  // the counter of the NEXT use: last read in the previous use, which ended before the caller's barrier behind it
  if (pool.cap && threadIdx.x == 0) pool.count[(pool.use & 1u) ^ 1u] = 0;
  PhiloxVKeys vk_;
  if (!LEAN) vk_ = philox_vkeys(key.k0, key.k1);
  const PhiloxVKeys *const vk = LEAN ? nullptr : &vk_;
  // One cell: pairs of attempts until one is accepted -- or, with on_list, one pair and then onto the list.  Two copies of
  // this (pass 0 and pass 1) on purpose: with one copy and run-time flags the hot loop of pass 0 carries the list's
  // branches and loads (measured: +2.4 % on the fused launch).
  auto cell = [&](uint32_t off, uint32_t pair, auto on_list) {
    VsCell c;
    setup(off, c);
    double th = 0.0;
    bool neg = false;
    for (;;) {
      if (vs_attempt_pair(key, vk, c.site, pair, c.kp, c.cls, pool.tab, [&] { return kappa_exact(off); }, th, neg)) {
        commit(off, mod_2pi_fast(c.centre + (neg ? -th : th)));
        return;
      }
      ++pair;
      if ((bool)on_list && pair == 1 && pool.cap) {   // (kMaxPair >= 1: the entry can hold it)
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < pool.cap) {
          pool.buf[slot] = (E)(off | (pair << P::kOffBits));
          return;
        }
      }
    }
  };
  // Pass 0 takes whole rounds of NT cells only: the cells left over (130 of the 2178 of a phase of the fused kernel, NT =
  // 512) would cost the first waves a fifth iteration with the others waiting at the barrier; they join the list instead,
  // where threads are idle anyway.
  const uint32_t n_main = pool.cap ? total / NT * NT : total;
  if (LEAN >= 2) {   // one loop over both passes, one copy of the cell code
    uint32_t n = n_main, n_list = 0;
    for (int pass = 0; pass < 2; ++pass) {
      for (uint32_t i = threadIdx.x; i < n; i += NT) {
        uint32_t off, pair = 0;
        if (pass == 0) {
          off = off_of(i);
        } else if (i < n_list) {
          const uint32_t e = pool.buf[i];
          off = e & kOffMask;
          pair = e >> P::kOffBits;
        } else {
          off = off_of(n_main + (i - n_list));
        }
        cell(off, pair, pass == 0);
      }
      if (!pool.cap) break;
      if (pass == 0) {
        __syncthreads();
        n_list = min(*cnt, pool.cap);
        n = n_list + (total - n_main);
      }
    }
    ++pool.use;
    return;
  }
  for (uint32_t i = threadIdx.x; i < n_main; i += NT) cell(off_of(i), 0u, std::true_type{});
  if (pool.cap) {
    __syncthreads();
    const uint32_t n_list = min(*cnt, pool.cap);   // (the same value in every thread: nothing is pushed behind the barrier)
    const uint32_t n = n_list + (total - n_main);
    for (uint32_t i = threadIdx.x; i < n; i += NT) {
      uint32_t off, pair = 0;
      if (i < n_list) {
        const uint32_t e = pool.buf[i];
        off = e & kOffMask;
        pair = e >> P::kOffBits;
      } else {
        off = off_of(n_main + (i - n_list));
      }
      cell(off, pair, std::false_type{});
    }
  }
  ++pool.use;
}

// The same colour phase for callers that know a closed-form map thread -> cells (the fused Schwinger launch: a wave takes
// whole rows of the tile image).  heatbath_cells_step above hands out cells by their linear index, so every cell pays an
// integer division for its (row, column), a second one inside setup() for the Philox site, and the wraps of both
// coordinates: ~30 integer instructions of a ~200-instruction cell.  Here pass 0 is NIT rounds of NT cells whose LDS offset
// and Philox site the caller advances itself -- main_cell(off, site) is called NIT times by every thread, in order -- and
// only what the map leaves over (n_left cells: left_off(i), i < n_left) joins the list pass, whose cells get their site from
// site_of(off) as before.  Which cell a lane works on does not enter any result (a cell's random numbers are fixed by
// (site, attempt)): bit for bit the results of heatbath_cells_step.  stencil_load(off, v) issues the six reads of the cell's
// stencil (lds_read_f64: no wait), stencil_cell(v, cell) = centre, class, kappa' from them (not the site).
template <int NT, int NIT, class E, class Main, class Left, class SiteOf, class StencilLoad, class StencilCell, class KappaExact, class Commit>
__device__ __forceinline__ void heatbath_cells_step_mapped(uint32_t n_left, const RngKey &key, VsPool<E> &pool, Main main_cell, Left left_off,
                                                           SiteOf site_of, StencilLoad stencil_load, StencilCell stencil_cell,
                                                           KappaExact kappa_exact, Commit commit) {
  using P = VsPool<E>;
  constexpr uint32_t kOffMask = (1u << P::kOffBits) - 1;
  uint32_t *const cnt = pool.count + (pool.use & 1u);
  // the counter of the NEXT use (see heatbath_cells_step); this form needs the list: pool.cap > 0
  if (threadIdx.x == 0) pool.count[(pool.use & 1u) ^ 1u] = 0;
  const PhiloxVKeys vk_ = philox_vkeys(key.k0, key.k1);
  const PhiloxVKeys *const vk = &vk_;
  auto cell = [&](uint32_t off, uint32_t site, uint32_t pair, auto on_list) {   // two copies, as in heatbath_cells_step
    VsCell c;
    double sv[6];
    stencil_load(off, sv);
    lds_wait6(sv);
    stencil_cell(sv, c);
    double th = 0.0;
    bool neg = false;
    for (;;) {
      if (vs_attempt_pair(key, vk, site, pair, c.kp, c.cls, pool.tab, [&] { return kappa_exact(off); }, th, neg)) {
        commit(off, mod_2pi_fast(c.centre + (neg ? -th : th)));
        return;
      }
      ++pair;
      if ((bool)on_list && pair == 1) {
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < pool.cap) {
          pool.buf[slot] = (E)(off | (pair << P::kOffBits));
          return;
        }
      }
    }
  };
#ifndef MLMCPI_HB_CELLS1
  // Two cells per lane and round, stage by stage (stencils, Philox calls, table look-ups, screening tests of both, then the
  // rare exact test of either behind ONE branch): a cell is one dependency chain with four LDS round trips in it, and a SIMD
  // holds four waves -- two of them this stage's --, so the second chain fills what the first one waits for
  static_assert(NIT % 2 == 0, "rounds come in pairs");
  for (int k = 0; k < NIT; k += 2) {
    uint32_t off[2], site[2], ca[2], cb[2];
    VsCell c[2];
    U4 q[2];
    float pa[2], pb[2];
    bool acc_a[2], acc_b[2], rej_a[2], rej_b[2];
    double sv[2][6];
#pragma unroll
    for (int j = 0; j < 2; ++j) main_cell(off[j], site[j]);
    // the twelve stencil reads (single ds_read_b64: callers issue them through lds_read_f64) are in flight under the two
    // Philox calls, which need nothing but the sites
#pragma unroll
    for (int j = 0; j < 2; ++j) stencil_load(off[j], sv[j]);
    const uint32_t w3 = (P_VONMISES << 24);
#pragma unroll
    for (int j = 0; j < 2; ++j) q[j] = philox4x32_10(site[j], key.chain, key.step, w3, key.k0, key.k1, vk_);
    lds_wait12(sv[0], sv[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) stencil_cell(sv[j], c[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      pa[j] = vs_accept_prob(q[j].y, c[j].kp, c[j].cls, pool.tab, ca[j]);
      pb[j] = vs_accept_prob(q[j].w, c[j].kp, c[j].cls, pool.tab, cb[j]);
    }
    bool open = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float la = (float)q[j].x, lb = (float)q[j].z;
      acc_a[j] = la <= fmaf(pa[j], pool.tab.s_acc, -kVsU2Slack);
      acc_b[j] = lb <= fmaf(pb[j], pool.tab.s_acc, -kVsU2Slack);
      rej_a[j] = la >= fmaf(pa[j], pool.tab.s_rej, kVsU2Slack);
      rej_b[j] = lb >= fmaf(pb[j], pool.tab.s_rej, kVsU2Slack);
      // open: the first attempt undecided, or rejected and the second undecided
      open = open || (!acc_a[j] && (!rej_a[j] || (!acc_b[j] && !rej_b[j])));
    }
    if (open) {   // (one attempt in ~10^4: the exact test, as in vs_attempt_pair)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (!acc_a[j] && (!rej_a[j] || (!acc_b[j] && !rej_b[j]))) {
          const uint8_t *row = pool.tab.base + c[j].cls + kVsLwOff;
          const uint32_t r = vs_exact_pair(key.k0, key.k1, key.chain, key.step, site[j], w3, q[j], ca[j] | (cb[j] << 8), *(const float *)(row + ca[j]),
                                           *(const float *)(row + cb[j]), kappa_exact(off[j]), rej_a[j] ? 0 : -1, acc_b[j] ? 1 : rej_b[j] ? 0 : -1,
                                           pool.tab.base);
          acc_a[j] = (r & 3u) == 1u;
          acc_b[j] = (r >> 2) == 1u;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (acc_a[j] || acc_b[j]) {
        const bool first = acc_a[j];
        const uint32_t lo = first ? q[j].x : q[j].z, hi = first ? q[j].y : q[j].w;
        const double th = vs_theta(lo, hi, first ? ca[j] : cb[j], pool.tab);
        commit(off[j], mod_2pi_fast(c[j].centre + ((lo & 0x200u) ? -th : th)));
      } else {
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < pool.cap) pool.buf[slot] = (E)(off[j] | (1u << P::kOffBits));
        else cell(off[j], site[j], 1u, std::false_type{});   // list full: finished where it is
      }
    }
  }
#else
  for (int k = 0; k < NIT; ++k) {
    uint32_t off, site;
    main_cell(off, site);
    cell(off, site, 0u, std::true_type{});
  }
#endif
  __syncthreads();
  const uint32_t n_list = min(*cnt, pool.cap);
  const uint32_t n = n_list + n_left;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    uint32_t off, pair = 0;
    if (i < n_list) {
      const uint32_t e = pool.buf[i];
      off = e & kOffMask;
      pair = e >> P::kOffBits;
    } else {
      off = left_off(i - n_list);
    }
    cell(off, site_of(off), pair, std::false_type{});
  }
  ++pool.use;
}

// one draw x ~ exp(kappa cos(x - centre)) for the conditional between x_p and x_m (test hook; the sweeps go through
// heatbath_cells_step): returns mod_2pi(centre + x)
__device__ __forceinline__ double vs_draw(const RngKey &k, uint32_t site, double scale, double x_p, double x_m, const VsTable &tab) {
  VsCell c;
  vs_cell(scale, x_p, x_m, c);
  double th = 0.0;
  bool neg = false;
  for (uint32_t pair = 0; !vs_attempt_pair(k, (const PhiloxVKeys *)nullptr, site, pair, c.kp, c.cls, tab, [&] { return vs_kappa_exact(scale, x_p, x_m); }, th, neg); ++pair) {
  }
  return mod_2pi_fast(c.centre + (neg ? -th : th));
}

// -log of ExpCosDistribution::evaluate(x, x_p, x_m) (distribution/expcosdistribution.cc:7-21)
__device__ __forceinline__ double expcos_neg_log_pdf(double beta, double x, double x_p, double x_m) {
  double dx = x_p - x_m, z = x - x_m;
  double flip = (dx < 0.0) ? -1.0 : 1.0;
  dx *= flip;
  if (dx > kPi) {
    flip = -flip;
    dx = kTwoPi - dx;
  }
  z *= flip;
  const double sigma = 2. * beta * fabs(cos(0.5 * dx));
  return -sigma * (cos(z - 0.5 * dx) - 1.0) + log(kTwoPi * bessel_i0_scaled(sigma));
}

// ---- fill-in distributions of the Schwinger lattice coarsened in both directions ----------------------------
// distribution/besselproductdistribution.{hh,cc} (beta <= 8), approximatebesselproductdistribution.{hh,cc} (beyond)
struct BesselFill {
  double beta, I0_twobeta, sigma_beta;
  double alphaZ[17];   // besselproductdistribution.hh:55-71 (host-built)
  int approximate;     // beta > 8 (quenchedschwingerconditionedfineaction.hh:62-71)
};

__device__ __forceinline__ double bessel_i0(double z) {  // gsl_sf_bessel_I0
  const double az = fabs(z);
  return exp(az) * bessel_i0_scaled(az);
}

// BesselProductDistribution::Znorm_inv(phi, rescaled = true), besselproductdistribution.cc:15-25
__device__ __forceinline__ double bessel_znorm_inv_rescaled(const BesselFill &P, double phi) {
  double s = 1.0;
  for (int k = 1; k <= 16; ++k) s += P.alphaZ[k] * cos(k * phi);
  return 1.0 / s;
}

// BesselProductDistribution::draw (besselproductdistribution.hh:88-152).  The calls of `site` are numbered
// n = 0, 1, ...: an outer attempt takes one call (two uniforms), the truncated-normal loop one call per two
// normals; n is bounded, so every lane leaves the loop.
__device__ __forceinline__ double bessel_product_draw(const RngKey &k, uint32_t site, const BesselFill &P, double x_p,
                                                      double x_m) {
  double dx = x_m - x_p;
  const double flip = (dx < 0) ? -1. : +1.;
  dx *= flip;
  const double N_p = erf((kPi - 0.5 * dx) / P.sigma_beta);
  const double N_m = erf(0.5 * dx / P.sigma_beta) * pow(P.I0_twobeta, 2. * (dx / kPi - 1.));
  const double C_p = pow(P.I0_twobeta, 2. * (1. - dx * dx / (4. * kPi * kPi)));
  const double C_m = pow(P.I0_twobeta, 2. * (1. - (dx - 2. * kPi) * (dx - 2. * kPi) / (4. * kPi * kPi)));
  const double sigma = P.sigma_beta / sqrt(2.);
  uint32_t n = 0;
  double x = 0.0;
  while (n < 60000u) {
    double xi, xi2;
    rng_uniforms(k, site, P_BESSEL, n++, xi, xi2);
    double a_min, a_max, mu, C;
    if (xi >= N_m / (N_p + N_m)) {
      a_min = -kPi + dx; a_max = +kPi; mu = 0.5 * dx; C = C_p;
    } else {
      a_min = -kPi; a_max = -kPi + dx; mu = 0.5 * (dx - 2. * kPi); C = C_m;
    }
    bool inside = false;
    while (!inside && n < 60000u) {
      double g0, g1;
      rng_normals(k, site, P_BESSEL, n++, g0, g1);
      x = sigma * g0 + mu;
      inside = (x >= a_min) && (x < a_max);
      if (!inside) {
        x = sigma * g1 + mu;
        inside = (x >= a_min) && (x < a_max);
      }
    }
    const double I0 = bessel_i0(2. * P.beta * cos(0.5 * x));
    const double I0_dx = bessel_i0(2. * P.beta * cos(0.5 * (x - dx)));
    const double xs = (x - mu) / P.sigma_beta;
    if (xi2 <= I0 * I0_dx / C * exp(xs * xs)) break;
  }
  return mod_2pi(flip * x + x_p);
}

// approximatebesselproductdistribution.cc:43-54
__device__ __forceinline__ void approx_bessel_params(double beta, double x0, double &N_p, double &s2p_inv,
                                                     double &s2m_inv) {
  if (x0 < 0.125 * kPi) {
    s2p_inv = beta; s2m_inv = 0.0; N_p = 1.0;
  } else {
    s2p_inv = beta * cos(0.25 * x0);
    s2m_inv = beta * sin(0.25 * x0);
    const double rho = pow(s2p_inv / s2m_inv, 1.5) * exp(-4.0 * (s2p_inv - s2m_inv));
    N_p = 1.0 / (1.0 + rho);
  }
}
// approximatebesselproductdistribution.hh:82-107: call 0 = the uniform, call 1 = the normal
__device__ __forceinline__ double approx_bessel_draw(const RngKey &k, uint32_t site, double beta, double x_p,
                                                     double x_m) {
  double x0 = x_p - x_m;
  double flip = (x0 < 0) ? -1. : +1.;
  x0 *= flip;
  if (x0 > kPi) { x0 = kTwoPi - x0; flip = -flip; }
  double N_p, s2p, s2m;
  approx_bessel_params(beta, x0, N_p, s2p, s2m);
  double xi, unused, g0, g1;
  rng_uniforms(k, site, P_BESSEL, 0, xi, unused);
  rng_normals(k, site, P_BESSEL, 1, g0, g1);
  const double sigma = (xi <= N_p) ? 1. / sqrt(s2p) : 1. / sqrt(s2m);
  const double xshift = (xi <= N_p) ? 0.0 : kPi;
  const double x = sigma * g0 + 0.5 * x0 - xshift;
  return mod_2pi(flip * x + x_m);
}
// approximatebesselproductdistribution.cc:7-40
__device__ __forceinline__ double approx_bessel_pdf(double beta, double x, double x_p, double x_m) {
  double x0 = x_p - x_m, z = x - x_m;
  double flip = (x0 < 0) ? -1. : +1.;
  x0 *= flip;
  if (x0 > kPi) { x0 = kTwoPi - x0; flip = -flip; }
  z *= flip;
  double N_p, s2p, s2m;
  approx_bessel_params(beta, x0, N_p, s2p, s2m);
  const double N_m = 1. - N_p;
  double sp = 0.0, sm = 0.0;
  for (int kk = -4; kk <= 4; ++kk) {
    double zs = z - 0.5 * x0 + 2 * kk * kPi;
    sp += sqrt(s2p) * exp(-0.5 * s2p * zs * zs);
    zs += kPi;
    sm += sqrt(s2m) * exp(-0.5 * s2m * zs * zs);
  }
  return sqrt(0.5 / kPi) * (N_p * sp + N_m * sm);
}

// ---- GaussianFillinDistribution (distribution/gaussianfillindistribution.{hh,cc}): the four interior links of a 2 x 2
// block given the four perimeter sums phi_12 .. phi_41, as a two-peak Gaussian mixture in three non-trivial directions
// (eta_1, eta_2, eta_3) plus a uniform common shift omega.  Used by QuenchedSchwingerGaussianConditionedFineAction.
__device__ __forceinline__ double gaussfill_pc(double beta, double Phi) {  // gaussianfillindistribution.hh get_pc
  if (Phi < 0.125 * kPi) return 1.0;
  if (Phi > 0.375 * kPi) return 0.0;
  const double sp = beta * cos(Phi), sm = beta * sin(Phi);
  const double rho = pow(sp / sm, 1.5) * exp(-4.0 * (sp - sm));
  return 1. / (1. + rho);
}

// gaussianfillindistribution.hh draw (add_gaussian_noise = true): calls of `site` with purpose P_GAUSSFILL: 0 -> (xi, omega / 2 pi),
// 1 -> normals of eta_1, eta_2, 2 -> normal of eta_3
__device__ __forceinline__ void gaussfill_draw(const RngKey &k, uint32_t site, double beta, double phi_12, double phi_23,
                                               double phi_34, double phi_41, double (&theta)[4]) {
  const double Phi = 0.25 * (phi_12 + phi_23 + phi_34 + phi_41);
  double Phi_star = Phi;
  bool swap_eta = false, shift_eta = false;
  if (Phi_star < 0) { Phi_star = -Phi_star; swap_eta = true; }
  if (Phi_star > 0.5 * kPi) { Phi_star = kPi - Phi_star; swap_eta = !swap_eta; shift_eta = true; }
  const double p_c = gaussfill_pc(beta, Phi_star);
  double xi, om, n1, n2, n3, unused;
  rng_uniforms(k, site, P_GAUSSFILL, 0, xi, om);
  rng_normals(k, site, P_GAUSSFILL, 1, n1, n2);
  rng_normals(k, site, P_GAUSSFILL, 2, n3, unused);
  double eta_1, eta_2, eta_3, sigma;
  if (xi < p_c) {
    eta_1 = 0.0; eta_2 = 0.0; eta_3 = 0.0;
    sigma = 1. / sqrt(4. * beta * cos(Phi_star));
  } else {
    eta_1 = kPi; eta_2 = 0.0; eta_3 = 0.5 * kPi;
    sigma = 1. / sqrt(4. * beta * sin(Phi_star));
  }
  const double sqrt2 = 1.41421356237309504880;
  eta_1 += sqrt2 * sigma * n1;
  eta_2 += sqrt2 * sigma * n2;
  eta_3 += sigma * n3;
  if (swap_eta) { const double t = eta_1; eta_1 = eta_2; eta_2 = t; }
  if (shift_eta) { eta_1 += kPi; eta_2 += kPi; }
  const double omega = 2. * kPi * om;
  theta[0] = mod_2pi(0.5 * (+eta_1 + eta_2 + eta_3) + omega);
  theta[1] = mod_2pi(0.5 * (+eta_1 - eta_2 - eta_3) + omega + Phi - phi_12);
  theta[2] = mod_2pi(0.5 * (-eta_1 - eta_2 + eta_3) + omega + 2. * Phi - phi_12 - phi_23);
  theta[3] = mod_2pi(0.5 * (-eta_1 + eta_2 - eta_3) + omega + 3. * Phi - phi_12 - phi_23 - phi_34);
}

// gaussianfillindistribution.cc:7-67 (add_gaussian_noise = true).  The peak lattices of construct_peaks (:70-118), in
// units of pi/2: main peaks = {0 (mod 4)}^3 and {2 (mod 4)}^3, secondary peaks = (2 mod 4, 0 mod 4, 1 mod 4) and
// (0 mod 4, 2 mod 4, 3 mod 4), each coordinate within one period of the base cell (n_offsets = 1; 0 for beta > 72, which
// keeps only the base cell's 9 + 4 peaks).
__device__ __forceinline__ double gaussfill_pdf(double beta, double theta_1, double theta_2, double theta_3, double theta_4,
                                                double phi_12, double phi_23, double phi_34, double phi_41) {
  double eta_1 = mod_2pi(0.5 * (theta_1 + theta_2 - theta_3 - theta_4) + 0.5 * (phi_41 - phi_23));
  double eta_2 = mod_2pi(0.5 * (theta_1 - theta_2 - theta_3 + theta_4) + 0.5 * (phi_34 - phi_12));
  const double eta_3 = mod_2pi(0.5 * (theta_1 - theta_2 + theta_3 - theta_4) + 0.25 * (-phi_12 + phi_23 - phi_34 + phi_41));
  double Phi_star = 0.25 * (phi_12 + phi_23 + phi_34 + phi_41);
  bool swap_eta = false;
  if (Phi_star < 0.) { Phi_star = -Phi_star; swap_eta = true; }
  if (Phi_star > 0.5 * kPi) {
    Phi_star = kPi - Phi_star;
    swap_eta = !swap_eta;
    eta_1 = mod_2pi(eta_1 + kPi);
    eta_2 = mod_2pi(eta_2 + kPi);
  }
  if (swap_eta) { const double t = eta_1; eta_1 = eta_2; eta_2 = t; }
  const double p_c = gaussfill_pc(beta, Phi_star);
  const double s2c = 2. * beta * cos(Phi_star), s2s = 2. * beta * sin(Phi_star);
  const bool wide = !(beta > 72.0);  // n_offsets = 1
  const double h = 0.5 * kPi;
  auto gauss = [&](double s2, int px, int py, int pz) {
    const double d1 = eta_1 - h * px, d2 = eta_2 - h * py, d3 = eta_3 - h * pz;
    return exp(-0.5 * s2 * (d1 * d1 + d2 * d2 + 2. * d3 * d3));
  };
  double g_c = 0.0, g_s = 0.0;
  if (wide) {
    for (int a = -4; a <= 4; a += 4)
      for (int b = -4; b <= 4; b += 4)
        for (int c = -4; c <= 4; c += 4) g_c += gauss(s2c, a, b, c);
    for (int a = -6; a <= 6; a += 4)
      for (int b = -6; b <= 6; b += 4)
        for (int c = -6; c <= 6; c += 4) g_c += gauss(s2c, a, b, c);
    for (int a = -6; a <= 6; a += 4)
      for (int b = -4; b <= 4; b += 4)
        for (int c = -3; c <= 5; c += 4) g_s += gauss(s2s, a, b, c);
    for (int a = -4; a <= 4; a += 4)
      for (int b = -6; b <= 6; b += 4)
        for (int c = -5; c <= 3; c += 4) g_s += gauss(s2s, a, b, c);
  } else {
    g_c += gauss(s2c, 0, 0, 0);
    for (int a = -2; a <= 2; a += 4)
      for (int b = -2; b <= 2; b += 4)
        for (int c = -2; c <= 2; c += 4) g_c += gauss(s2c, a, b, c);
    g_s = gauss(s2s, 2, 0, 1) + gauss(s2s, -2, 0, 1) + gauss(s2s, 0, 2, -1) + gauss(s2s, 0, -2, -1);
  }
  return p_c * pow(s2c, 1.5) * g_c + (1. - p_c) * pow(s2s, 1.5) * g_s;
}

// ---- reductions -------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;  // valid in lane 0
}

// Sum over the workgroup in a fixed order (lane tree, then waves 0..n-1): bitwise reproducible.
// `scratch` needs blockDim.x/64 doubles per value.  Result valid in thread 0.
// Rotate a value through the 64 lanes of a wave by one lane (DPP wave_ror:1 / wave_rol:1, gfx9 family): no LDS, no
// barrier.  "up": lane l receives the value of lane l - 1 and lane 0 that of lane 63; "down": lane l receives lane l + 1.
__device__ __forceinline__ double wave_rotate_up(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_rotate_down(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

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
