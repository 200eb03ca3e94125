// oracle.cc -- CPU restatement of the inner MCMC sweep of eikehmueller/mlmcpathintegral.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mlmcpathintegral_amd/, include/) may
// link, import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg do, and there only as the checker / the CPU baseline.
//
// Pinning status: the reference's own tests contain no assertions or golden vectors (SURVEY.md
// section 4), and its action/sampler/QoI translation units need Eigen3 + GSL, which this image
// lacks, so they are unbuildable here (no stand-in headers are written).  The restatement is
// pinned by (i) the known answers recorded from the compiled reference in SURVEY.md 8(c)
// (tests/golden/survey_known_answers.json), (ii) the analytic expectation values the
// reference's drivers print, and (iii) oracle/_ref (the reference's Eigen/GSL-free lattice and
// statistics sources compiled where they lie) for index maps and the statistics estimators.
//
// Two families of functions live here:
//   * "reference order": sequential semantics of the reference, std::mt19937_64 with the
//     reference's seeds and libstdc++ distributions (chain-exact against the reference when
//     built with the same libstdc++);
//   * "device order": the multicolour / counter-based-RNG (Philox4x32-10) ordering that the HIP
//     kernels implement, restated sequentially on the CPU.  Both families share the same
//     per-site update formulas, which follow the reference files cited at each function.
//
// All paths cited are relative to /root/reference/src.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <numeric>
#include <memory>
#include <random>
#include <set>
#include <vector>

namespace {

const double kPi = 3.14159265358979323846;

// common/auxilliary.hh:42-44
inline double wrap_2pi(double x) { return x - 2. * kPi * std::floor(0.5 * (x + kPi) / kPi); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123 v1.x constants).  Not part of the reference:
// this is the device-order RNG.  Known-answer vectors: tests/golden/philox_kat.json.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// Purposes (upper byte of counter word 3) -- see DESIGN.md "RNG contract".
enum Purpose : uint32_t {
  P_MOMENTUM = 1,  // HMC momenta, one call per site, Box-Muller cosine branch
  P_ACCEPT = 2,    // HMC Metropolis uniform
  P_GFF_NORMAL = 3,  // GFF heat bath, one call per vertex pair (l>>1), branch l&1
  P_VONMISES = 4,    // heat-bath angle draws (Schwinger, rotor): one call per attempt, sub = attempt
  P_INIT = 6,        // initial states: one uniform per entry
  P_FILLIN = 7,      // two-level step: Gaussian fill-in of fine-only sites (site = fine index)
  P_ACCEPT2 = 8,     // two-level step: Metropolis uniform
  P_BESSEL = 9,      // two-level step, Schwinger coarsened in both directions: Bessel-product fill-in (sub = call counter)
  P_EXACT = 10,      // exact sampler of the harmonic oscillator: entries (2m, 2m+1) = Box-Muller pair of site m
};

inline double u01(uint32_t lo, uint32_t hi) {
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);  // [0,1), 53 bits
}

struct DevRng {
  uint32_t k0, k1, chain, step;
  Philox4 raw(uint32_t site, Purpose p, uint32_t sub) const {
    return philox4x32_10(site, chain, step, ((uint32_t)p << 24) | (sub & 0xFFFFFFu), k0, k1);
  }
  void uniforms(uint32_t site, Purpose p, uint32_t sub, double &a, double &b) const {
    Philox4 r = raw(site, p, sub);
    a = u01(r.v[0], r.v[1]);
    b = u01(r.v[2], r.v[3]);
  }
  // Box-Muller: radius from 1-u (in (0,1]), angle 2 pi v.
  void normals(uint32_t site, Purpose p, uint32_t sub, double &n0, double &n1) const {
    double u, v;
    uniforms(site, p, sub, u, v);
    double r = std::sqrt(-2.0 * std::log(1.0 - u));
    double phi = 2.0 * kPi * v;
    n0 = r * std::cos(phi);
    n1 = r * std::sin(phi);
  }
};

// ---------------------------------------------------------------------------------------------
// Random sources for the rejection samplers: one interface, two back ends.
// ---------------------------------------------------------------------------------------------
struct RefRng {  // reference order: engine + distribution objects that cache (A.4 of SURVEY.md)
  std::mt19937_64 engine;
  std::normal_distribution<double> normal{0.0, 1.0};
  std::uniform_real_distribution<double> uniform{0.0, 1.0};
  explicit RefRng(uint64_t seed) : engine(seed) {}
};

struct RefAttemptSource {
  RefRng &r;
  double next_normal() { return r.normal(r.engine); }
  double next_uniform() { return r.uniform(r.engine); }
};

// distribution/expsin2distribution.hh:45-58
template <class Src>
double expsin2_draw(Src &src, double sigma) {
  const double scale = kPi / std::sqrt(2. * sigma);
  for (;;) {
    double r = scale * src.next_normal();
    if (std::fabs(r) < kPi) {
      double s = std::sin(0.5 * r);
      double u = src.next_uniform();
      if (u < std::exp(-sigma * (s * s - r * r / (kPi * kPi)))) return r;
    }
  }
}

// distribution/expcosdistribution.hh:51-65
template <class Src>
double expcos_draw(Src &src, double beta, double x_p, double x_m) {
  const double dx = x_m - x_p;
  const double tau = 2. * beta * std::fabs(std::cos(0.5 * dx));
  const double sigma = kPi * std::sqrt(2. / tau);
  const double inv4pi2 = 1. / (4. * kPi * kPi);
  double x;
  for (;;) {
    x = sigma * src.next_normal();
    if (-kPi <= x && x < kPi) {
      double u = src.next_uniform();
      if (u <= std::exp(tau * (std::cos(x) - 1. + inv4pi2 * x * x))) break;
    }
  }
  return wrap_2pi(x + 0.5 * (x_p + x_m) + (std::fabs(dx) > kPi ? kPi : 0.0));
}

// Device-order angle sampler.  Both heat-bath conditionals of the reference are von Mises laws:
//   ExpCos   p(x) ~ exp(tau cos(x - centre)), tau = 2 beta |cos(dx/2)|   (expcosdistribution.cc:7-21)
//   ExpSin2  p(x) ~ exp(-sigma sin^2(x/2)) = exp((sigma/2)(cos x - 1))     (expsin2distribution.cc:20-24)
// The reference draws them by rejection from a Gaussian envelope whose acceptance rate is
// sqrt(tau/pi) I0(tau) e^-tau <= 0.27 and tends to 0 like sqrt(tau) for flat conditionals.  On a
// 64-wide wave the slowest lane sets the pace (and at 1024^2 x batch some link always has tau ~ 1e-8,
// i.e. ~1e4 attempts), so the device path samples the SAME distribution with the wrapped-Cauchy
// envelope of Best & Fisher (Appl. Statist. 28 (1979) 152-157), whose acceptance rate is >= 0.65
// for every concentration.  Equality in distribution with the reference's samplers is a test
// (tests/test_distributions.py), not an assumption.
//
// Arithmetic and random-number layout follow the device (mlmcpathintegral_amd/csrc/device_common.hpp, "heat-bath angle
// draws"): Best & Fisher's r = (1 + rho^2)/(2 rho) equals (1 + s)/(2 kappa), s = sqrt(1 + 4 kappa^2); with
// R = kappa r = (1 + s)/2:  z = cos(pi u1), f = cos(theta) = (kappa + R z)/(R + kappa z), c = R - kappa f, accept when
// u2 <= c exp(1 - c).  One Philox call (word 3 = P_VONMISES << 24 | sub0 | t) feeds attempts 2t (words 0, 1) and 2t + 1
// (words 2, 3); of an attempt's 64 bits v = hi:lo, u1 = (v >> 12) 2^-52, bit 0 is the sign of the angle, bits 1..11 are
// the leading bits b of u2 = (b + u2') / 2048, and the tail u2' (53 bits) comes from the call with word 3 | kVmRefine.
// The device consults the tail only when b does not decide; the decision is the same either way.
constexpr uint32_t kVmFillin = 1u << 23;  // sub-stream of the two-level fill-in draws
constexpr uint32_t kVmRefine = 1u << 22;  // the call holding the tails of a pair's acceptance uniforms
constexpr uint32_t kMaxVmPairs = 512u;
inline double dev_vonmises(const DevRng &rng, uint32_t site, double kappa, uint32_t sub0 = 0) {
  kappa = std::fmax(kappa, 1e-12);  // also maps NaN to a finite concentration: the loop always ends
  const double R = 0.5 + 0.5 * std::sqrt(1. + 4. * kappa * kappa);
  double f = 1.0;
  bool negative = false;
  for (uint32_t pair = 0; pair < kMaxVmPairs; ++pair) {
    const Philox4 w = rng.raw(site, P_VONMISES, sub0 | pair);
    const Philox4 e = rng.raw(site, P_VONMISES, sub0 | kVmRefine | pair);
    bool accepted = false;
    for (int h = 0; h < 2 && !accepted; ++h) {
      const uint32_t lo = w.v[2 * h], hi = w.v[2 * h + 1];
      const uint64_t v = ((uint64_t)hi << 32) | lo;
      const double u1 = (double)(v >> 12) * (1.0 / 4503599627370496.0);  // 52 bits
      negative = (lo & 1u) != 0;
      const double u2 = ((double)((lo >> 1) & 0x7FFu) + u01(e.v[2 * h], e.v[2 * h + 1])) * (1.0 / 2048.0);
      const double z = std::cos(kPi * u1);
      f = (kappa + R * z) / (R + kappa * z);
      const double c = R - kappa * f;
      accepted = (c * (2. - c) - u2 > 0.) || (std::log(c / u2) + 1. - c >= 0.);
    }
    if (accepted) break;
  }
  f = std::fmin(1.0, std::fmax(-1.0, f));
  const double theta = std::acos(f);
  return negative ? -theta : theta;
}

// Device-order angle sampler for moderate concentrations (sweeps of actions with kappa_max = scale <= kVsKappaMax;
// device: device_common.hpp, "tabulated step envelope").  Same von Mises law p(x) ~ exp(kappa cos x), piecewise-constant
// envelope: |x| falls into one of eight bins with edges (0, 1, 2, 3, 4, 6, 8, 12, 16) pi/16, bin k proposed with
// probability q_k / 64, |x| uniform inside the bin, accepted with probability
//     exp(kappa (cos x - 1)) (w_k / q_k) / max_j (H_j w_j / q_j),      H_j = exp(kappa_min (cos(edge_j) - 1)),
// = target / (proposal density x envelope constant) for every kappa >= kappa_min.  The q_k depend on the range the
// concentration lies in: with t = |(x_m - x_p) / (4 pi)| reduced to [0, 1/2], kappa = scale sin(2 pi v), v = |t - 1/4|,
// class c = floor(32 v) (at most 7), kappa_min(c) = scale sin(2 pi c / 32).  The integers q_k are chosen by a
// deterministic rule (below); tests compare them, and the acceptance factors, with the product's table
// (mlmcpi_vs_table).  Bits of an attempt v = hi:lo as for dev_vonmises (bit 0 sign, bits 1..11 leading bits of u2, tail
// of u2 from the refine call); of the 52 bits above them the top six select the bin (selector value s belongs to bin k
// when q_0 + ... + q_{k-1} <= s < q_0 + ... + q_k) and the other 46 are the position inside it.  The test is taken in
// logarithms, exactly as the device's exact path does.
constexpr double kVsKappaMax = 16.0;   // (round 5; 4 before: the device's rule, device_common.hpp)
constexpr int kVsClasses = 8, kVsBins = 8, kVsSel = 64;
struct VsTables {
  int q[kVsClasses][kVsBins];
  float lw[kVsClasses][kVsBins];  // log2 of the acceptance factor, as the float the device holds
  double scale = -1.0;
  static const int *edges16() {
    static const int e[kVsBins + 1] = {0, 1, 2, 3, 4, 6, 8, 12, 16};
    return e;
  }
  void build(double scale_) {
    scale = scale_;
    const int *e16 = edges16();
    for (int c = 0; c < kVsClasses; ++c) {
      const double kmin = scale * std::sin(2.0 * kPi * c / 32.0);
      double hw[kVsBins], w[kVsBins], total = 0.0;
      for (int k = 0; k < kVsBins; ++k) {
        w[k] = (e16[k + 1] - e16[k]) * (kPi / 16.0);
        hw[k] = std::exp(kmin * (std::cos(e16[k] * (kPi / 16.0)) - 1.0)) * w[k];
        total += hw[k];
      }
      // every bin at least one selector value; then the remaining values one at a time to the bin that currently
      // binds the envelope constant (largest H w / q); should the minimum of one have overshot, values are taken back
      // from the bin that stays lowest
      int sum = 0;
      for (int k = 0; k < kVsBins; ++k) {
        q[c][k] = std::max(1, (int)std::floor(hw[k] / total * kVsSel));
        sum += q[c][k];
      }
      for (; sum < kVsSel; ++sum) {
        int j = 0;
        for (int k = 1; k < kVsBins; ++k)
          if (hw[k] / q[c][k] > hw[j] / q[c][j]) j = k;
        ++q[c][j];
      }
      for (; sum > kVsSel; --sum) {
        int j = -1;
        for (int k = 0; k < kVsBins; ++k)
          if (q[c][k] > 1 && (j < 0 || hw[k] / (q[c][k] - 1) < hw[j] / (q[c][j] - 1))) j = k;
        --q[c][j];
      }
      double M = 0.0;
      for (int k = 0; k < kVsBins; ++k) M = std::fmax(M, hw[k] / q[c][k]);
      for (int k = 0; k < kVsBins; ++k) {
        const double x = std::log2(w[k] / q[c][k] / M) - 4e-6;  // a hair below: float rounding must not lift it above 1 / H_k
        float f = (float)x;
        if ((double)f > x) f = std::nextafterf(f, -INFINITY);
        lw[c][k] = f;
      }
    }
  }
};
inline const VsTables &vs_tables(double scale) {
  static thread_local VsTables t;
  if (t.scale != scale) t.build(scale);
  return t;
}

// one draw between the staple sums / neighbours x_p, x_m; returns the angle relative to the centre
inline double dev_vonmises_table(const DevRng &rng, uint32_t site, double scale, double x_p, double x_m, uint32_t sub0 = 0) {
  const VsTables &T = vs_tables(scale);
  const double v = (x_m - x_p) * (0.25 / kPi);
  const double t = std::fabs(v - std::rint(v));
  const int cls = std::min(kVsClasses - 1, (int)(32.0 * std::fabs(t - 0.25)));
  const double kappa = std::fmax(scale * std::fabs(std::cos(0.5 * (x_m - x_p))), 1e-12);
  const int *e16 = VsTables::edges16();
  double theta = 0.0;
  bool negative = false;
  for (uint32_t pair = 0; pair < kMaxVmPairs; ++pair) {
    const Philox4 w = rng.raw(site, P_VONMISES, sub0 | pair);
    const Philox4 e = rng.raw(site, P_VONMISES, sub0 | kVmRefine | pair);
    bool accepted = false;
    for (int h = 0; h < 2 && !accepted; ++h) {
      const uint32_t lo = w.v[2 * h], hi = w.v[2 * h + 1];
      int sel = (int)(hi >> 26), k = 0;
      while (sel >= T.q[cls][k]) sel -= T.q[cls][k++];
      // fields of an attempt (r04 layout): lo[31..10] the 22 leading bits of u2, lo[9] sign, lo[8..0] the low 9 position
      // bits; hi[31..26] selector, hi[25..0] the high 26 position bits
      const uint64_t pos35 = ((uint64_t)(hi & 0x3FFFFFFu) << 9) | (lo & 0x1FFu);
      const double pos = (double)pos35 * (1.0 / 34359738368.0);  // 35 bits
      theta = (kPi / 16.0) * ((double)e16[k] + (double)(e16[k + 1] - e16[k]) * pos);
      negative = (lo & 0x200u) != 0;
      const double u2 = ((double)(lo >> 10) + u01(e.v[2 * h], e.v[2 * h + 1])) * (1.0 / 4194304.0);
      accepted = u2 <= 0.0 || std::log(u2) <= kappa * (std::cos(theta) - 1.0) + 0.69314718055994531 * (double)T.lw[cls][k];
    }
    if (accepted) break;
  }
  return negative ? -theta : theta;
}

struct RefAngles {  // reference algorithms, reference engine
  RefRng &r;
  double expcos(uint32_t, double beta, double x_p, double x_m) {
    RefAttemptSource src{r};
    return expcos_draw(src, beta, x_p, x_m);
  }
  double expsin2(uint32_t, double sigma) {
    RefAttemptSource src{r};
    return expsin2_draw(src, sigma);
  }
};

struct DevAngles {  // device order: Philox + Best-Fisher, or the tabulated step envelope for actions of moderate concentration
  const DevRng &rng;
  double scale = 1e300;  // the largest concentration the action can produce (2 beta; 2 m0 / a): picks the sampler
  // angle relative to the centre of the conditional between x_p and x_m, concentration kappa = scale |cos((x_m - x_p)/2)|
  double between(uint32_t site, double x_p, double x_m, double kappa) const {
    return scale <= kVsKappaMax ? dev_vonmises_table(rng, site, scale, x_p, x_m) : dev_vonmises(rng, site, kappa);
  }
  double expcos(uint32_t site, double beta, double x_p, double x_m) {
    const double dx = x_m - x_p;
    const double tau = 2. * beta * std::fabs(std::cos(0.5 * dx));
    const double x = scale <= kVsKappaMax ? between(site, x_p, x_m, tau) : dev_vonmises(rng, site, tau);
    return wrap_2pi(x + 0.5 * (x_p + x_m) + (std::fabs(dx) > kPi ? kPi : 0.0));  // expcosdistribution.hh:64
  }
  double expsin2(uint32_t site, double sigma) { return dev_vonmises(rng, site, 0.5 * sigma); }
};

// ---------------------------------------------------------------------------------------------
// Lattice index maps.  lattice/lattice2d.hh:230-268,348-375; lattice/lattice2d.cc:137-155;
// lattice/lattice1d.cc:6-19.
// ---------------------------------------------------------------------------------------------
struct Grid2 {
  int Mt, Mx;
  bool rotated;
  unsigned vertex(int i, int j) const {
    if (rotated) {
      int ht = Mt / 2, hx = Mx / 2;
      int is = ((i + Mt) - (i & 1)) / 2;
      int js = ((j + Mx) - (j & 1)) / 2;
      int off = (Mt * Mx / 4) * (i & 1);
      return ht * (js % hx) + is % ht + off;
    }
    return Mt * ((j + Mx) % Mx) + ((i + Mt) % Mt);
  }
  void vertex_inv(unsigned l, int &i, int &j) const {
    if (rotated) {
      int ht = Mt / 2;
      int quarter = Mt * Mx / 4;
      int par = l / quarter;
      unsigned lh = l - quarter * par;
      int jh = lh / ht;
      j = 2 * jh + par;
      i = 2 * (lh - ht * jh) + par;
    } else {
      j = l / Mt;
      i = l - Mt * j;
    }
  }
  unsigned link(int i, int j, int mu) const {
    return 2 * Mt * ((j + Mx) % Mx) + 2 * ((i + Mt) % Mt) + mu;
  }
  void link_inv(unsigned l, int &i, int &j, int &mu) const {
    j = l / (2 * Mt);
    unsigned r = l - (2 * Mt) * j;
    i = r >> 1;
    mu = r & 1;
  }
  unsigned nvertices() const { return rotated ? Mt * Mx / 2 : Mt * Mx; }
  void neighbours(unsigned l, unsigned out[8]) const {
    static const int di_plain[8] = {+1, -1, 0, 0, +1, +1, -1, -1};
    static const int dj_plain[8] = {0, 0, +1, -1, +1, -1, +1, -1};
    static const int di_rot[8] = {+1, +1, -1, -1, +2, -2, 0, 0};
    static const int dj_rot[8] = {+1, -1, +1, -1, 0, 0, +2, -2};
    int i, j;
    vertex_inv(l, i, j);
    for (int k = 0; k < 8; ++k)
      out[k] = rotated ? vertex(i + di_rot[k], j + dj_rot[k]) : vertex(i + di_plain[k], j + dj_plain[k]);
  }
};

// ---------------------------------------------------------------------------------------------
// Actions.  One struct with a kind tag keeps the ctypes surface small.
// ---------------------------------------------------------------------------------------------
enum Kind { HARMONIC = 0, QUARTIC = 1, ROTOR = 2, GFF = 3, SCHWINGER = 4 };

struct ActionO {
  Kind kind;
  // 1-D
  unsigned M = 0;
  double T_final = 0, a = 0, m0 = 0, mu2 = 0, lambda = 0, x0 = 0;
  // 2-D
  Grid2 g{0, 0, false};
  double beta = 0, mass = 0, gff_mu2 = 0, gff_sigma = 0;
  RefRng rng;  // the action's own engine + distribution members
  std::uniform_real_distribution<double> init_uniform{-kPi, kPi};

  ActionO(Kind k, uint64_t seed) : kind(k), rng(seed) {}

  unsigned size() const {
    switch (kind) {
      case GFF: return g.Mt * g.Mx;
      case SCHWINGER: return 2 * g.Mt * g.Mx;
      default: return M;
    }
  }

  // ---- evaluate ----------------------------------------------------------------------------
  double evaluate(const double *x) const {
    switch (kind) {
      case HARMONIC: {  // action/qm/harmonicoscillatoraction.cc:8-18
        double inv_a2 = 1. / (a * a);
        double d = x[0] - x[M - 1];
        double S = inv_a2 * d * d + mu2 * x[0] * x[0];
        for (unsigned j = 1; j < M; ++j) {
          double dj = x[j] - x[j - 1];
          S += inv_a2 * dj * dj + mu2 * x[j] * x[j];
        }
        return 0.5 * a * m0 * S;
      }
      case QUARTIC: {  // action/qm/quarticoscillatoraction.cc:7-27
        double inv_a2 = 1. / (a * a);
        double S = 0;
        for (unsigned j = 0; j < M; ++j) {
          double xj = x[j];
          double d = xj - x[(j + M - 1) % M];
          double sh = xj - x0;
          double sh2 = sh * sh;
          double term = m0 * (inv_a2 * d * d + mu2 * (xj * xj)) + 0.5 * lambda * sh2 * sh2;
          S = (j == 0) ? term : S + term;
        }
        return 0.5 * a * S;
      }
      case ROTOR: {  // action/qm/rotoraction.cc:9-18
        double S = 1. - std::cos(x[0] - x[M - 1]);
        for (unsigned j = 1; j < M; ++j) S += 1. - std::cos(x[j] - x[j - 1]);
        return m0 / a * S;
      }
      case GFF: {  // action/qft/gffaction.cc:8-30 (n_gibbs_smooth == 0 branch)
        double kappa = 4. + gff_mu2, S = 0;
        unsigned nb[8];
        for (unsigned l = 0; l < g.nvertices(); ++l) {
          g.neighbours(l, nb);
          double loc = kappa * x[l];
          for (int k = 0; k < 4; ++k) loc -= x[nb[k]];
          S += x[l] * loc;
        }
        return 0.5 * S;
      }
      case SCHWINGER: {  // action/qft/quenchedschwingeraction.cc:7-22
        double S = 0;
        for (int i = 0; i < g.Mt; ++i)
          for (int j = 0; j < g.Mx; ++j) S += 1. - std::cos(plaquette(x, i, j));
        return beta * S;
      }
    }
    return 0;
  }

  double plaquette(const double *x, int i, int j) const {
    return x[g.link(i, j, 0)] + x[g.link(i + 1, j, 1)] - x[g.link(i, j + 1, 0)] - x[g.link(i, j, 1)];
  }

  // ---- force -------------------------------------------------------------------------------
  void force(const double *x, double *f) const {
    switch (kind) {
      case HARMONIC:  // action/qm/harmonicoscillatoraction.cc:21-35
      case QUARTIC: { // action/qm/quarticoscillatoraction.cc:30-53
        double c1 = m0 / a, c2 = 2. + a * a * mu2, c3 = a * lambda;
        for (unsigned j = 0; j < M; ++j) {
          double xm = x[(j + M - 1) % M], xp = x[(j + 1) % M];
          double v = c1 * (c2 * x[j] - xm - xp);
          if (kind == QUARTIC) {
            double sh = x[j] - x0;
            v += c3 * sh * sh * sh;
          }
          f[j] = v;
        }
        return;
      }
      case ROTOR: {  // action/qm/rotoraction.cc:59-79
        double c = m0 / a;
        for (unsigned j = 0; j < M; ++j) {
          double xm = x[(j + M - 1) % M], xp = x[(j + 1) % M];
          f[j] = c * (std::sin(x[j] - xm) + std::sin(x[j] - xp));
        }
        return;
      }
      case GFF: {  // action/qft/gffaction.cc:80-94
        double kappa = 4. + gff_mu2;
        unsigned nb[8];
        for (unsigned l = 0; l < g.nvertices(); ++l) {
          g.neighbours(l, nb);
          double v = kappa * x[l];
          for (int k = 0; k < 4; ++k) v -= x[nb[k]];
          f[l] = v;
        }
        return;
      }
      case SCHWINGER: {  // action/qft/quenchedschwingeraction.cc:68-89 (scatter form, same order)
        for (unsigned l = 0; l < size(); ++l) f[l] = 0.0;
        for (int i = 0; i < g.Mt; ++i)
          for (int j = 0; j < g.Mx; ++j) {
            double F = beta * std::sin(plaquette(x, i, j));
            f[g.link(i, j, 0)] += F;
            f[g.link(i + 1, j, 1)] += F;
            f[g.link(i, j + 1, 0)] -= F;
            f[g.link(i, j, 1)] -= F;
          }
        return;
      }
    }
  }

  // ---- conditioned single-site quantities ------------------------------------------------------
  // action/qm/rotoraction.hh:195-213, action/qm/quarticoscillatoraction.hh:160-194
  double w_minimum(double xm, double xp) const {
    if (kind == ROTOR) return std::atan2(std::sin(xp) + std::sin(xm), std::cos(xp) + std::cos(xm));
    if (kind == HARMONIC) return (0.5 / (1. + 0.5 * a * a * mu2)) * (xm + xp);  // harmonicoscillatoraction.hh:98,187-189
    if (kind == QUARTIC) {
      double xbar = 0.5 * (xm + xp), rho = 1. / (1. + 0.5 * a * a * mu2), x = xbar;
      for (int it = 0; it < 4; ++it) {
        double sh = x - x0;
        x = rho * (xbar - 0.5 * a * a * lambda / m0 * sh * sh * sh);
      }
      return x;
    }
    return 0.0;
  }
  double w_curvature(double xm, double xp) const {
    if (kind == ROTOR) return 2.0 * m0 / a * std::fabs(std::cos(0.5 * (xp - xm)));
    if (kind == HARMONIC) return (2. / a + a * mu2) * m0;  // harmonicoscillatoraction.hh:97,171-174
    if (kind == QUARTIC) {
      double x = 0.5 * (xm + xp);
      return (2. / a + a * mu2) * m0 + 3. * lambda * a * (x - x0) * (x - x0);
    }
    return 0.0;
  }

  // action/qft/quenchedschwingeraction.cc:25-43
  void staples(const double *x, int i, int j, int mu, double &tp, double &tm) const {
    if (mu == 0) {
      tp = wrap_2pi(x[g.link(i, j + 1, 0)] + x[g.link(i, j, 1)] - x[g.link(i + 1, j, 1)]);
      tm = wrap_2pi(x[g.link(i, j - 1, 0)] + x[g.link(i + 1, j - 1, 1)] - x[g.link(i, j - 1, 1)]);
    } else {
      tp = wrap_2pi(x[g.link(i, j, 0)] + x[g.link(i + 1, j, 1)] - x[g.link(i, j + 1, 0)]);
      tm = wrap_2pi(x[g.link(i - 1, j + 1, 0)] + x[g.link(i - 1, j, 1)] - x[g.link(i - 1, j, 0)]);
    }
  }

  double gff_delta(const double *x, unsigned l) const {
    unsigned nb[8];
    g.neighbours(l, nb);
    double D = 0.0;
    for (int k = 0; k < 4; ++k) D += x[nb[k]];
    return D;
  }

  // ---- local updates; Src supplies randomness ----------------------------------------------
  // rotoraction.cc:40-56, gffaction.cc:68-77, quenchedschwingeraction.cc:57-65
  bool overrelax(double *x, unsigned l) const {
    switch (kind) {
      case ROTOR: {
        double xm = x[(l + M - 1) % M], xp = x[(l + 1) % M];
        x[l] = wrap_2pi(2.0 * w_minimum(xm, xp) - x[l]);
        return true;
      }
      case GFF:
        x[l] = 2. * gff_delta(x, l) / (4. + gff_mu2) - x[l];
        return true;
      case SCHWINGER: {
        int i, j, mu;
        double tp, tm;
        g.link_inv(l, i, j, mu);
        staples(x, i, j, mu, tp, tm);
        x[l] = wrap_2pi((tp + tm) - x[l]);
        return true;
      }
      default: return false;  // action/action.hh:90-96: not implemented for HO / quartic
    }
  }

  // rotoraction.cc:20-37, gffaction.cc:33-42, quenchedschwingeraction.cc:46-54
  template <class Angles>
  bool heatbath(double *x, unsigned l, Angles &src, double gff_normal) const {
    switch (kind) {
      case ROTOR: {
        double xm = x[(l + M - 1) % M], xp = x[(l + 1) % M];
        double x_min = w_minimum(xm, xp);
        double sigma = 2. * w_curvature(xm, xp);
        x[l] = wrap_2pi(x_min + src.expsin2(l, sigma));
        return true;
      }
      case GFF:
        x[l] = gff_sigma * gff_normal + gff_delta(x, l) / (4. + gff_mu2);
        return true;
      case SCHWINGER: {
        int i, j, mu;
        double tp, tm;
        g.link_inv(l, i, j, mu);
        staples(x, i, j, mu, tp, tm);
        x[l] = src.expcos(l, beta, tp, tm);
        return true;
      }
      default: return false;
    }
  }

  bool heatbath_ref(double *x, unsigned l) {
    RefAngles src{rng};
    double n = (kind == GFF) ? rng.normal(rng.engine) : 0.0;
    return heatbath(x, l, src, n);
  }

  // rotoraction.cc:82-89, quenchedschwingeraction.cc:198-204, zeros for HO / quartic
  // (harmonicoscillatoraction.hh:155-158, quarticoscillatoraction.hh:143-146).  The reference's
  // GFF initial state is an exact sparse-Cholesky draw (gffaction.cc:121-123) which is out of
  // scope; zeros are used instead.
  void initialise_ref(double *x) {
    unsigned n = size();
    if (kind == ROTOR || kind == SCHWINGER)
      for (unsigned l = 0; l < n; ++l) x[l] = init_uniform(rng.engine);
    else
      for (unsigned l = 0; l < n; ++l) x[l] = 0.0;
  }

  // ---- device-order colouring -----------------------------------------------------------------
  int n_colours() const { return kind == SCHWINGER ? 4 : 2; }
  int colour_of(unsigned l) const {
    switch (kind) {
      case GFF: {
        int i, j;
        g.vertex_inv(l, i, j);
        return (i + j) & 1;
      }
      case SCHWINGER: {
        int i, j, mu;
        g.link_inv(l, i, j, mu);
        return mu == 0 ? (j & 1) : 2 + (i & 1);
      }
      default: return l & 1;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Device-order sweeps and HMC trajectory.
// ---------------------------------------------------------------------------------------------
// One full sweep: colours in ascending order, every entry updated exactly once.
// Rotor, device order.  With d = (x+ - x-)/2:  sin x+ + sin x- = 2 sin((x+ + x-)/2) cos d  and
// cos x+ + cos x- = 2 cos((x+ + x-)/2) cos d,  so getWminimum = atan2(...) (rotoraction.hh:206-213) is the mean angle
// (x+ + x-)/2, shifted by pi when cos d < 0.  The kernels use that closed form (no atan2, no sincos; better conditioned
// than the quotient of two cancelling sums when cos d -> 0): overrelaxation mod_2pi(2 x0 - x) = mod_2pi(x+ + x- - x),
// heat bath mod_2pi(x0 + ExpSin2(2 W'')) with the von Mises concentration kappa = W'' = (2 m0 / a) |cos d|.
// tests/test_oracle_golden.py checks the two forms against each other.
void rotor_dev_update(const ActionO &A, double *x, unsigned l, bool heat, const DevRng &rng) {
  const unsigned M = A.M;
  const double xm = x[(l + M - 1) % M], xp = x[(l + 1) % M];
  if (!heat) {
    x[l] = wrap_2pi(xm + xp - x[l]);
    return;
  }
  const double c = std::cos(0.5 * (xp - xm));
  const double kappa = 2.0 * A.m0 / A.a * std::fabs(c);
  const double centre = 0.5 * (xp + xm) + (c < 0.0 ? kPi : 0.0);
  const DevAngles src{rng, 2.0 * A.m0 / A.a};
  x[l] = wrap_2pi(centre + src.between(l, xp, xm, kappa));
}

// Action::heatbath_update / overrelaxation_update(state, l) in device order (Philox stream of site l): the unit the
// sweeps are made of, and what the device's site-at-a-time entry points compute
void dev_site_update(const ActionO &A, double *x, unsigned l, bool heat, const DevRng &rng) {
  if (A.kind == ROTOR) {
    rotor_dev_update(A, x, l, heat, rng);
    return;
  }
  if (!heat) {
    A.overrelax(x, l);
    return;
  }
  DevAngles src{rng, A.kind == SCHWINGER ? 2.0 * A.beta : 1e300};
  double gn = 0.0;
  if (A.kind == GFF) {
    double n0, n1;
    rng.normals(l >> 1, P_GFF_NORMAL, 0, n0, n1);
    gn = (l & 1) ? n1 : n0;
  }
  A.heatbath(x, l, src, gn);
}

void dev_sweep(const ActionO &A, double *x, bool heat, const DevRng &rng) {
  unsigned n = A.size();
  for (int c = 0; c < A.n_colours(); ++c)
    for (unsigned l = 0; l < n; ++l)
      if (A.colour_of(l) == c) dev_site_update(A, x, l, heat, rng);
}

// energies[0..3] = S(x_cur), T(p_0), S(x_trial), T(p_end); returns accept flag and leaves the
// accepted state in x.  Leapfrog scheme: sampler/hmcsampler.cc:22-69.
int dev_hmc_trajectory(const ActionO &A, double *x, unsigned nt, double dt, const DevRng &rng,
                       double *energies, double *dH_out) {
  unsigned n = A.size();
  std::vector<double> p(n), xt(x, x + n), f(n);
  for (unsigned l = 0; l < n; ++l) {
    double n0, n1;
    rng.normals(l, P_MOMENTUM, 0, n0, n1);
    p[l] = n0;
  }
  double T0 = 0;
  for (unsigned l = 0; l < n; ++l) T0 += p[l] * p[l];
  T0 *= 0.5;
  for (unsigned k = 0; k <= nt; ++k) {
    double dtp = (k == 0 || k == nt) ? 0.5 * dt : dt;
    double dtx = (k == nt) ? 0.0 : dt;
    A.force(xt.data(), f.data());
    for (unsigned l = 0; l < n; ++l) {
      p[l] -= dtp * f[l];
      xt[l] += dtx * p[l];
    }
  }
  double T1 = 0;
  for (unsigned l = 0; l < n; ++l) T1 += p[l] * p[l];
  T1 *= 0.5;
  double S0 = A.evaluate(x), S1 = A.evaluate(xt.data());
  double dH = (S1 - S0) + (T1 - T0);
  if (energies) {
    energies[0] = S0; energies[1] = T0; energies[2] = S1; energies[3] = T1;
  }
  if (dH_out) *dH_out = dH;
  bool acc;
  if (dH < 0.0) {
    acc = true;
  } else {
    double u, v;
    rng.uniforms(0, P_ACCEPT, 0, u, v);
    acc = u < std::exp(-dH);
  }
  if (acc) std::copy(xt.begin(), xt.end(), x);
  return acc ? 1 : 0;
}

// Two-level Metropolis step, device order.  montecarlo/twolevelmetropolisstep.cc:35-89 with
// action/qm/gaussianconditionedfineaction.cc:7-43 and action/qm/qmaction.cc:7-24.
// theta (fine, current state of the step) is updated in place on acceptance; terms = the three action
// differences (fine, coarse, trial).
// distribution/expsin2distribution.cc:7-24; std::cyl_bessel_i stands in for gsl_sf_bessel_I0_scaled (GSL is
// not in this image)
double two_pi_i0_scaled(double z) {
  if (z > 100.) {
    double zi = 1. / z;
    return std::sqrt(2. * kPi * zi) * (1. + 0.125 * zi + 0.0703125 * zi * zi);
  }
  return 2. * kPi * std::exp(-z) * std::cyl_bessel_i(0.0, z);
}
double expsin2_neg_log_pdf(double x, double sigma) {
  double sh = std::sin(0.5 * x);
  return -std::log(std::exp(-sigma * sh * sh) / two_pi_i0_scaled(0.5 * sigma));
}

// gaussianconditionedfineaction.cc:27-43 / rotorconditionedfineaction.cc:26-43, same summation order
double cfa_term(const ActionO &F, double x, double xm, double xp) {
  double d = x - F.w_minimum(xm, xp), c = F.w_curvature(xm, xp);
  if (F.kind == ROTOR) return expsin2_neg_log_pdf(d, 2.0 * c);
  return 0.5 * c * d * d - 0.5 * std::log(c);
}
double cfa_gaussian(const ActionO &F, const double *x) {
  unsigned M = F.M;
  double S = cfa_term(F, x[M - 1], x[M - 2], x[0]);
  for (unsigned j = 0; j < M / 2 - 1; ++j) S += cfa_term(F, x[2 * j + 1], x[2 * j], x[2 * j + 2]);
  return S;
}

int dev_twolevel_draw(const ActionO &F, const ActionO &Cc, const double *x_coarse, double *theta, const DevRng &rng,
                      double *terms) {
  unsigned M = F.M, Mc = M / 2;
  std::vector<double> tp(M), thetaC(Mc);
  for (unsigned j = 0; j < Mc; ++j) tp[2 * j] = x_coarse[j];  // copy_from_coarse
  for (unsigned j = 0; j < Mc; ++j) {                         // fill_fine_points
    double xm = tp[2 * j], xp = tp[(2 * j + 2) % M];
    double x0 = F.w_minimum(xm, xp);
    if (F.kind == ROTOR) {  // rotorconditionedfineaction.cc:7-24
      double sigma = 2. * F.w_curvature(xm, xp);
      tp[2 * j + 1] = wrap_2pi(x0 + dev_vonmises(rng, 2 * j + 1, 0.5 * sigma, kVmFillin));
      continue;
    }
    double sigma = 1. / std::sqrt(F.w_curvature(xm, xp));
    double n0, n1;
    rng.normals(2 * j + 1, P_FILLIN, 0, n0, n1);
    tp[2 * j + 1] = x0 + n0 * sigma;
  }
  double dS_fine = F.evaluate(tp.data()) - F.evaluate(theta);
  for (unsigned j = 0; j < Mc; ++j) thetaC[j] = theta[2 * j];  // copy_from_fine
  double dS_coarse = Cc.evaluate(thetaC.data()) - Cc.evaluate(x_coarse);
  double dS_trial = cfa_gaussian(F, theta) - cfa_gaussian(F, tp.data());
  double dS = dS_fine + dS_coarse + dS_trial;
  if (terms) { terms[0] = dS_fine; terms[1] = dS_coarse; terms[2] = dS_trial; }
  bool acc;
  if (dS < 0.0) {
    acc = true;
  } else {
    double u, v;
    rng.uniforms(0, P_ACCEPT2, 0, u, v);
    acc = u < std::exp(-dS);
  }
  if (acc) std::copy(tp.begin(), tp.end(), theta);
  return acc ? 1 : 0;
}

// common/fastbessel.cc:7-55: exp(-z) I0(z); beyond z = 100 the reference sums the Hankel series with the
// coefficients of fastbessel.hh:25-33 (4-7 terms), below it calls GSL (std::cyl_bessel_i stands in here).
double fast_bessel_i0_scaled(double z) {
  if (z > 100.) {
    const int terms = z > 1100. ? 4 : (z > 400. ? 5 : (z > 200. ? 6 : 7));
    double coeff[8];
    coeff[0] = 1.0;
    for (int n = 1; n <= terms; ++n) coeff[n] = 0.125 * (2.0 * n - 1.0) * (2.0 * n - 1.0) / n * coeff[n - 1];
    const double zi = 1. / z;
    double p = coeff[terms];
    for (int n = terms - 1; n >= 0; --n) p = zi * p + coeff[n];
    return p / std::sqrt(2. * kPi * z);
  }
  return std::exp(-z) * std::cyl_bessel_i(0.0, z);
}

// distribution/expcosdistribution.cc:7-21
double expcos_pdf(double beta, double x, double x_p, double x_m) {
  double dx = x_p - x_m, z = x - x_m;
  int flip = (dx < 0.0) ? -1 : +1;
  dx *= flip;
  if (dx > kPi) {
    flip *= -1;
    dx = 2. * kPi - dx;
  }
  z *= flip;
  const double sigma = 2. * beta * std::fabs(std::cos(0.5 * dx));
  const double Z = 2. * kPi * fast_bessel_i0_scaled(sigma);
  return 1. / Z * std::exp(sigma * (std::cos(z - 0.5 * dx) - 1.0));
}

// QuenchedSchwingerSemiConditionedFineAction::evaluate (quenchedschwingerconditionedfineaction.cc:332-379);
// rt == 2: the fine lattice is twice as long in the temporal direction as the coarse one, else in the spatial one
double schwinger_semi_cfa(const ActionO &F, int rt, const double *x) {
  const Grid2 &g = F.g;
  double S = 0.0;
  if (rt == 2) {
    for (int i = 0; i < g.Mt / 2; ++i)
      for (int j = 0; j < g.Mx; ++j) {
        double phi_p = wrap_2pi(-x[g.link(2 * i, j, 0)] + x[g.link(2 * i, j, 1)] + x[g.link(2 * i, j + 1, 0)]);
        double phi_m = wrap_2pi(+x[g.link(2 * i + 1, j, 0)] + x[g.link(2 * i + 2, j, 1)] - x[g.link(2 * i + 1, j + 1, 0)]);
        double theta = wrap_2pi(+x[g.link(2 * i + 1, j, 1)]);
        S -= std::log(expcos_pdf(F.beta, theta, phi_p, phi_m));
      }
  } else {
    for (int i = 0; i < g.Mt; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        double phi_p = wrap_2pi(-x[g.link(i, 2 * j, 1)] + x[g.link(i, 2 * j, 0)] + x[g.link(i + 1, 2 * j, 1)]);
        double phi_m = wrap_2pi(+x[g.link(i, 2 * j + 1, 1)] + x[g.link(i, 2 * j + 2, 0)] - x[g.link(i + 1, 2 * j + 1, 1)]);
        double theta = wrap_2pi(+x[g.link(i, 2 * j + 1, 0)]);
        S -= std::log(expcos_pdf(F.beta, theta, phi_p, phi_m));
      }
  }
  return S;
}

// quenchedschwingeraction.cc:147-195 (copy_from_fine: three coarsening cases); Mt, Mx = COARSE extents
void schwinger_copy_from_fine(int Mt, int Mx, int rt, int rx, const double *fine, double *coarse) {
  Grid2 gc{Mt, Mx, false}, gf{Mt * rt, Mx * rx, false};
  for (int i = 0; i < Mt; ++i)
    for (int j = 0; j < Mx; ++j) {
      double t0 = fine[gf.link(rt * i, rx * j, 0)];
      if (rt == 2) t0 += fine[gf.link(2 * i + 1, rx * j, 0)];
      double t1 = fine[gf.link(rt * i, rx * j, 1)];
      if (rx == 2) t1 += fine[gf.link(rt * i, 2 * j + 1, 1)];
      coarse[gc.link(i, j, 0)] = wrap_2pi(t0);
      coarse[gc.link(i, j, 1)] = wrap_2pi(t1);
    }
}
// quenchedschwingeraction.cc:92-144 (copy_from_coarse)
void schwinger_copy_from_coarse(int Mt, int Mx, int rt, int rx, const double *coarse, double *fine) {
  Grid2 gc{Mt, Mx, false}, gf{Mt * rt, Mx * rx, false};
  for (int i = 0; i < Mt; ++i)
    for (int j = 0; j < Mx; ++j) {
      double c0 = coarse[gc.link(i, j, 0)], c1 = coarse[gc.link(i, j, 1)];
      if (rt == 2) {
        fine[gf.link(2 * i, rx * j, 0)] = 0.5 * c0;
        fine[gf.link(2 * i + 1, rx * j, 0)] = 0.5 * c0;
      } else {
        fine[gf.link(i, rx * j, 0)] = c0;
      }
      if (rx == 2) {
        fine[gf.link(rt * i, 2 * j, 1)] = 0.5 * c1;
        fine[gf.link(rt * i, 2 * j + 1, 1)] = 0.5 * c1;
      } else {
        fine[gf.link(rt * i, j, 1)] = c1;
      }
    }
}

// ---- fill-in distributions of the Schwinger lattice coarsened in both directions --------------------------
// distribution/besselproductdistribution.{hh,cc} (beta <= 8) and approximatebesselproductdistribution.{hh,cc}
double bessel_i0(double z) { return std::cyl_bessel_i(0.0, std::fabs(z)); }  // gsl_sf_bessel_I0
double log_factorial(unsigned n) {  // auxilliary.cc:30-36
  double s = 0.0;
  for (unsigned k = 2; k <= n; ++k) s += std::log((double)k);
  return s;
}
double log_nCk(unsigned n, unsigned k) { return log_factorial(n) - log_factorial(k) - log_factorial(n - k); }

struct BesselProductO {
  double beta, I0_twobeta, sigma_beta;
  double alphaZ[17];
  explicit BesselProductO(double beta_) : beta(beta_) {  // besselproductdistribution.hh:44-72
    const unsigned kmax = 16, nmax = 32;
    I0_twobeta = bessel_i0(2 * beta);
    sigma_beta = kPi / std::sqrt(2 * std::log(I0_twobeta));
    double alpha0 = 1.0;
    for (unsigned k = 0; k <= kmax; ++k) {
      double sum = 0.0;
      for (unsigned n = k; n <= nmax; ++n)
        for (unsigned m = k; m <= nmax; ++m) {
          double log_comb = log_nCk(2 * n, n - k) + log_nCk(2 * m, m - k) - 2 * (log_factorial(n) + log_factorial(m));
          sum += std::pow(0.5 * beta, 2.0 * (n + m)) * std::exp(log_comb);
        }
      double alpha = ((k == 0) ? 2 : 4) * kPi * sum;
      if (k == 0) alpha0 = alpha; else alpha /= alpha0;
      alphaZ[k] = alpha;
    }
  }
  double Znorm_inv(double phi, bool rescaled) const {  // besselproductdistribution.cc:15-25
    double s = 1.0;
    for (unsigned k = 1; k <= 16; ++k) s += alphaZ[k] * std::cos(k * phi);
    if (!rescaled) s *= alphaZ[0];
    return 1.0 / s;
  }
  // besselproductdistribution.hh:88-152, device order: the calls of cell `site` are numbered n = 0, 1, ...;
  // an outer attempt takes one call (two uniforms), the truncated-normal loop one call per two normals.
  double draw(const DevRng &rng, uint32_t site, double x_p, double x_m) const {
    double dx = x_m - x_p;
    const double flip = (dx < 0) ? -1 : +1;
    dx *= flip;
    const double N_p = std::erf((kPi - 0.5 * dx) / sigma_beta);
    const double N_m = std::erf(0.5 * dx / sigma_beta) * std::pow(I0_twobeta, 2. * (dx / kPi - 1.));
    const double C_p = std::pow(I0_twobeta, 2. * (1. - dx * dx / (4. * kPi * kPi)));
    const double C_m = std::pow(I0_twobeta, 2. * (1. - (dx - 2. * kPi) * (dx - 2. * kPi) / (4. * kPi * kPi)));
    const double sigma = sigma_beta / std::sqrt(2.);
    uint32_t n = 0;
    double x = 0.0;
    while (n < 60000u) {
      double xi, xi2;
      rng.uniforms(site, P_BESSEL, n++, xi, xi2);
      double a_min, a_max, mu, C;
      if (xi >= N_m / (N_p + N_m)) {
        a_min = -kPi + dx; a_max = +kPi; mu = 0.5 * dx; C = C_p;
      } else {
        a_min = -kPi; a_max = -kPi + dx; mu = 0.5 * (dx - 2. * kPi); C = C_m;
      }
      bool inside = false;
      while (!inside && n < 60000u) {
        double g0, g1;
        rng.normals(site, P_BESSEL, n++, g0, g1);
        x = sigma * g0 + mu;
        inside = (x >= a_min) && (x < a_max);
        if (!inside) {
          x = sigma * g1 + mu;
          inside = (x >= a_min) && (x < a_max);
        }
      }
      const double I0 = bessel_i0(2. * beta * std::cos(0.5 * x));
      const double I0_dx = bessel_i0(2. * beta * std::cos(0.5 * (x - dx)));
      const double xs = (x - mu) / sigma_beta;
      if (xi2 <= I0 * I0_dx / C * std::exp(xs * xs)) break;
    }
    return wrap_2pi(flip * x + x_p);
  }
};

// approximatebesselproductdistribution.cc:43-54
void approx_bessel_params(double beta, double x0, double &N_p, double &s2p_inv, double &s2m_inv) {
  const double epsilon = 0.125 * kPi;
  if (x0 < epsilon) {
    s2p_inv = beta; s2m_inv = 0.0; N_p = 1.0;
  } else {
    s2p_inv = beta * std::cos(0.25 * x0);
    s2m_inv = beta * std::sin(0.25 * x0);
    const double rho = std::pow(s2p_inv / s2m_inv, 1.5) * std::exp(-4.0 * (s2p_inv - s2m_inv));
    N_p = 1.0 / (1.0 + rho);
  }
}
// approximatebesselproductdistribution.hh:82-107, device order: one call (uniform, unused) + one call (normal)
double approx_bessel_draw(const DevRng &rng, uint32_t site, double beta, double x_p, double x_m) {
  double x0 = x_p - x_m;
  double flip = (x0 < 0) ? -1 : +1;
  x0 *= flip;
  if (x0 > kPi) { x0 = 2. * kPi - x0; flip *= -1; }
  double N_p, s2p, s2m;
  approx_bessel_params(beta, x0, N_p, s2p, s2m);
  double xi, unused, g0, g1;
  rng.uniforms(site, P_BESSEL, 0, xi, unused);
  rng.normals(site, P_BESSEL, 1, g0, g1);
  const double sigma = (xi <= N_p) ? 1. / std::sqrt(s2p) : 1. / std::sqrt(s2m);
  const double xshift = (xi <= N_p) ? 0.0 : kPi;
  const double x = sigma * g0 + 0.5 * x0 - xshift;
  return wrap_2pi(flip * x + x_m);
}
// approximatebesselproductdistribution.cc:7-40
double approx_bessel_pdf(double beta, double x, double x_p, double x_m) {
  double x0 = x_p - x_m, z = x - x_m;
  double flip = (x0 < 0) ? -1 : +1;
  x0 *= flip;
  if (x0 > kPi) { x0 = 2. * kPi - x0; flip *= -1; }
  z *= flip;
  double N_p, s2p, s2m;
  approx_bessel_params(beta, x0, N_p, s2p, s2m);
  const double N_m = 1. - N_p;
  double sp = 0.0, sm = 0.0;
  for (int k = -4; k <= 4; ++k) {
    double zs = z - 0.5 * x0 + 2 * k * kPi;
    sp += std::sqrt(s2p) * std::exp(-0.5 * s2p * zs * zs);
    zs += kPi;
    sm += std::sqrt(s2m) * std::exp(-0.5 * s2m * zs * zs);
  }
  return std::sqrt(0.5 / kPi) * (N_p * sp + N_m * sm);
}

// QuenchedSchwingerConditionedFineAction::evaluate (quenchedschwingerconditionedfineaction.cc:207-289)
double schwinger_both_cfa(const ActionO &F, const BesselProductO *bp, const double *x) {
  const Grid2 &g = F.g;
  const double beta = F.beta;
  double S = 0.0;
  if (bp) {
    for (int i = 0; i < g.Mt / 2; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        double phi_12 = +x[g.link(2 * i, 2 * j + 1, 1)] + x[g.link(2 * i, 2 * j + 2, 0)];
        double phi_23 = +x[g.link(2 * i + 1, 2 * j + 2, 0)] - x[g.link(2 * i + 2, 2 * j + 1, 1)];
        double phi_34 = -x[g.link(2 * i + 1, 2 * j, 0)] - x[g.link(2 * i + 2, 2 * j, 1)];
        double phi_41 = -x[g.link(2 * i, 2 * j, 0)] + x[g.link(2 * i, 2 * j, 1)];
        double theta_1 = +x[g.link(2 * i, 2 * j + 1, 0)];
        double theta_2 = -x[g.link(2 * i + 1, 2 * j + 1, 1)];
        double theta_3 = -x[g.link(2 * i + 1, 2 * j + 1, 0)];
        double theta_4 = +x[g.link(2 * i + 1, 2 * j, 1)];
        double Phi = phi_12 + phi_23 + phi_34 + phi_41;
        S -= beta * (std::cos(theta_1 - theta_2 - phi_12) + std::cos(theta_2 - theta_3 - phi_23) +
                     std::cos(theta_3 - theta_4 - phi_34) + std::cos(theta_4 - theta_1 - phi_41));
        S -= std::log(bp->Znorm_inv(Phi, true));
      }
  } else {
    for (int i = 0; i < g.Mt / 2; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        double phi_p = wrap_2pi(+x[g.link(2 * i + 1, 2 * j, 0)] + x[g.link(2 * i + 2, 2 * j, 1)] +
                                x[g.link(2 * i + 2, 2 * j + 1, 1)] - x[g.link(2 * i + 1, 2 * j + 2, 0)]);
        double phi_m = wrap_2pi(-x[g.link(2 * i, 2 * j, 0)] + x[g.link(2 * i, 2 * j, 1)] + x[g.link(2 * i, 2 * j + 1, 1)] +
                                x[g.link(2 * i, 2 * j + 2, 0)]);
        double theta = wrap_2pi(+x[g.link(2 * i + 1, 2 * j, 1)] + x[g.link(2 * i + 1, 2 * j + 1, 1)]);
        S -= std::log(approx_bessel_pdf(beta, theta, phi_p, phi_m));
      }
    for (int i = 0; i < g.Mt; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        double phi_p = wrap_2pi(-x[g.link(i, 2 * j, 1)] + x[g.link(i, 2 * j, 0)] + x[g.link(i + 1, 2 * j, 1)]);
        double phi_m = wrap_2pi(+x[g.link(i, 2 * j + 1, 1)] + x[g.link(i, 2 * j + 2, 0)] - x[g.link(i + 1, 2 * j + 1, 1)]);
        double theta = wrap_2pi(+x[g.link(i, 2 * j + 1, 0)]);
        S -= std::log(expcos_pdf(beta, theta, phi_p, phi_m));
      }
  }
  return S;
}

// QuenchedSchwingerConditionedFineAction::fill_fine_points (quenchedschwingerconditionedfineaction.cc:7-78), device
// order: coarse cell c = j Mt_c + i: Philox(c, P_FILLIN, 0) -> (dtheta of the temporal pair, of the spatial pair),
// Philox(c, P_FILLIN, 1) -> dtheta of the interior pair, Bessel-product draw = stream (c, P_BESSEL), ExpCos draw of
// fine link l = von Mises stream (l, kVmFillin).
void schwinger_both_fill(const ActionO &F, const ActionO &Cc, const BesselProductO *bp, const DevRng &rng, double *tp) {
  const Grid2 &g = F.g, &gc = Cc.g;
  for (int i = 0; i < g.Mt / 2; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      double dt, dx;
      rng.uniforms((uint32_t)(j * gc.Mt + i), P_FILLIN, 0, dt, dx);
      dt = (2. * dt - 1.) * kPi;
      dx = (2. * dx - 1.) * kPi;
      tp[g.link(2 * i, 2 * j, 0)] = wrap_2pi(tp[g.link(2 * i, 2 * j, 0)] + dt);
      tp[g.link(2 * i + 1, 2 * j, 0)] = wrap_2pi(tp[g.link(2 * i + 1, 2 * j, 0)] - dt);
      tp[g.link(2 * i, 2 * j, 1)] = wrap_2pi(tp[g.link(2 * i, 2 * j, 1)] + dx);
      tp[g.link(2 * i, 2 * j + 1, 1)] = wrap_2pi(tp[g.link(2 * i, 2 * j + 1, 1)] - dx);
    }
  for (int i = 0; i < g.Mt / 2; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      const uint32_t c = (uint32_t)(j * gc.Mt + i);
      double theta_p = wrap_2pi(tp[g.link(2 * i + 1, 2 * j, 0)] + tp[g.link(2 * i + 2, 2 * j, 1)] +
                                tp[g.link(2 * i + 2, 2 * j + 1, 1)] - tp[g.link(2 * i + 1, 2 * j + 2, 0)]);
      double theta_m = wrap_2pi(tp[g.link(2 * i, 2 * j, 1)] + tp[g.link(2 * i, 2 * j + 1, 1)] + tp[g.link(2 * i, 2 * j + 2, 0)] -
                                tp[g.link(2 * i, 2 * j, 0)]);
      double theta_tilde = bp ? bp->draw(rng, c, theta_p, theta_m) : approx_bessel_draw(rng, c, F.beta, theta_p, theta_m);
      double d, unused;
      rng.uniforms(c, P_FILLIN, 1, d, unused);
      d = (2. * d - 1.) * kPi;
      tp[g.link(2 * i + 1, 2 * j, 1)] = wrap_2pi(0.5 * theta_tilde + d);
      tp[g.link(2 * i + 1, 2 * j + 1, 1)] = wrap_2pi(0.5 * theta_tilde - d);
    }
  for (int i = 0; i < g.Mt; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      double theta_p = wrap_2pi(tp[g.link(i, 2 * j, 0)] + tp[g.link(i + 1, 2 * j, 1)] - tp[g.link(i, 2 * j, 1)]);
      double theta_m = wrap_2pi(tp[g.link(i, 2 * j + 1, 1)] + tp[g.link(i, 2 * j + 2, 0)] - tp[g.link(i + 1, 2 * j + 1, 1)]);
      const unsigned l = g.link(i, 2 * j + 1, 0);
      const double dxx = theta_m - theta_p;
      const double tau = 2. * F.beta * std::fabs(std::cos(0.5 * dxx));
      const double x = dev_vonmises(rng, l, tau, kVmFillin);
      tp[l] = wrap_2pi(x + 0.5 * (theta_p + theta_m) + (std::fabs(dxx) > kPi ? kPi : 0.0));
    }
}

// Two-level step on the Schwinger lattice, semi-coarsening, device order: twolevelmetropolisstep.cc:35-89 with
// quenchedschwingeraction.cc:92-195 and QuenchedSchwingerSemiConditionedFineAction::fill_fine_points
// (quenchedschwingerconditionedfineaction.cc:130-204).  dtheta of coarse cell c comes from Philox(site c,
// P_FILLIN); the ExpCos draw of fine link l from the von Mises stream (site l, kVmFillin).
// twolevelmetropolisstep.cc:69-84
int twolevel_decide(double dS_fine, double dS_coarse, double dS_trial, const DevRng &rng, const std::vector<double> &tp,
                    double *theta, double *terms) {
  const double dS = dS_fine + dS_coarse + dS_trial;
  if (terms) { terms[0] = dS_fine; terms[1] = dS_coarse; terms[2] = dS_trial; }
  bool acc;
  if (dS < 0.0) {
    acc = true;
  } else {
    double u, v;
    rng.uniforms(0, P_ACCEPT2, 0, u, v);
    acc = u < std::exp(-dS);
  }
  if (acc) std::copy(tp.begin(), tp.end(), theta);
  return acc ? 1 : 0;
}

// ---- GaussianFillinDistribution (distribution/gaussianfillindistribution.{hh,cc}) and
// QuenchedSchwingerGaussianConditionedFineAction (quenchedschwingerconditionedfineaction.cc:81-134, 293-327) -------------
constexpr uint32_t P_GAUSSFILL = 13;  // coarse cell c: call 0 (xi, omega / 2 pi), calls 1, 2 the normals of eta_1, eta_2 / eta_3
struct GaussianFillinO {
  double beta;
  std::vector<std::array<double, 3>> main_peaks, secondary_peaks;
  explicit GaussianFillinO(double beta_) : beta(beta_) {  // gaussianfillindistribution.hh:33-46, .cc:70-118
    const int n_offsets = beta > 72.0 ? 0 : 1;
    std::set<std::array<int, 3>> main_idx, sec_idx;
    const int p_main[9][3] = {{0, 0, 0}, {2, 2, 2}, {-2, 2, 2}, {2, -2, 2}, {-2, -2, 2}, {2, 2, -2}, {-2, 2, -2}, {2, -2, -2}, {-2, -2, -2}};
    const int p_sec[4][3] = {{2, 0, 1}, {-2, 0, 1}, {0, 2, -1}, {0, -2, -1}};
    for (int kx = -n_offsets; kx <= n_offsets; ++kx)
      for (int ky = -n_offsets; ky <= n_offsets; ++ky)
        for (int kz = -n_offsets; kz <= n_offsets; ++kz) {
          for (auto &q : p_main) main_idx.insert({q[0] + 4 * kx, q[1] + 4 * ky, q[2] + 4 * kz});
          for (auto &q : p_sec) sec_idx.insert({q[0] + 4 * kx, q[1] + 4 * ky, q[2] + 4 * kz});
        }
    for (auto &x : main_idx) main_peaks.push_back({0.5 * kPi * x[0], 0.5 * kPi * x[1], 0.5 * kPi * x[2]});
    for (auto &x : sec_idx) secondary_peaks.push_back({0.5 * kPi * x[0], 0.5 * kPi * x[1], 0.5 * kPi * x[2]});
  }
  double get_pc(double Phi) const {
    if (Phi < 0.125 * kPi) return 1.0;
    if (Phi > 0.375 * kPi) return 0.0;
    const double sp = beta * std::cos(Phi), sm = beta * std::sin(Phi);
    return 1. / (1. + std::pow(sp / sm, 1.5) * std::exp(-4.0 * (sp - sm)));
  }
  void dev_draw(const DevRng &rng, uint32_t cell, double phi_12, double phi_23, double phi_34, double phi_41, double th[4]) const {
    const double Phi = 0.25 * (phi_12 + phi_23 + phi_34 + phi_41);
    double Phi_star = Phi;
    bool swap_eta = false, shift_eta = false;
    if (Phi_star < 0) { Phi_star *= -1.0; swap_eta = true; }
    if (Phi_star > 0.5 * kPi) { Phi_star = kPi - Phi_star; swap_eta = !swap_eta; shift_eta = true; }
    const double p_c = get_pc(Phi_star);
    double xi, om, n1, n2, n3, unused;
    rng.uniforms(cell, (Purpose)P_GAUSSFILL, 0, xi, om);
    rng.normals(cell, (Purpose)P_GAUSSFILL, 1, n1, n2);
    rng.normals(cell, (Purpose)P_GAUSSFILL, 2, n3, unused);
    double eta_1, eta_2, eta_3, sigma;
    if (xi < p_c) { eta_1 = eta_2 = eta_3 = 0.0; sigma = 1. / std::sqrt(4. * beta * std::cos(Phi_star)); }
    else { eta_1 = kPi; eta_2 = 0.0; eta_3 = 0.5 * kPi; sigma = 1. / std::sqrt(4. * beta * std::sin(Phi_star)); }
    eta_1 += std::sqrt(2.0) * sigma * n1;
    eta_2 += std::sqrt(2.0) * sigma * n2;
    eta_3 += sigma * n3;
    if (swap_eta) std::swap(eta_1, eta_2);
    if (shift_eta) { eta_1 += kPi; eta_2 += kPi; }
    const double omega = 2. * kPi * om;
    th[0] = wrap_2pi(0.5 * (+eta_1 + eta_2 + eta_3) + omega);
    th[1] = wrap_2pi(0.5 * (+eta_1 - eta_2 - eta_3) + omega + Phi - phi_12);
    th[2] = wrap_2pi(0.5 * (-eta_1 - eta_2 + eta_3) + omega + 2. * Phi - phi_12 - phi_23);
    th[3] = wrap_2pi(0.5 * (-eta_1 + eta_2 - eta_3) + omega + 3. * Phi - phi_12 - phi_23 - phi_34);
  }
  double evaluate(double theta_1, double theta_2, double theta_3, double theta_4, double phi_12, double phi_23, double phi_34,
                  double phi_41) const {
    double eta_1 = wrap_2pi(0.5 * (theta_1 + theta_2 - theta_3 - theta_4) + 0.5 * (phi_41 - phi_23));
    double eta_2 = wrap_2pi(0.5 * (theta_1 - theta_2 - theta_3 + theta_4) + 0.5 * (phi_34 - phi_12));
    double eta_3 = wrap_2pi(0.5 * (theta_1 - theta_2 + theta_3 - theta_4) + 0.25 * (-phi_12 + phi_23 - phi_34 + phi_41));
    double Phi_star = 0.25 * (phi_12 + phi_23 + phi_34 + phi_41);
    bool swap_eta = false;
    if (Phi_star < 0.) { Phi_star *= -1.0; swap_eta = true; }
    if (Phi_star > 0.5 * kPi) {
      Phi_star = kPi - Phi_star;
      swap_eta = !swap_eta;
      eta_1 = wrap_2pi(eta_1 + kPi);
      eta_2 = wrap_2pi(eta_2 + kPi);
    }
    if (swap_eta) std::swap(eta_1, eta_2);
    const double p_c = get_pc(Phi_star);
    const double s2c = 2. * beta * std::cos(Phi_star), s2s = 2. * beta * std::sin(Phi_star);
    double g_c = 0.0, g_s = 0.0;
    for (auto &p : main_peaks) {
      const double d1 = eta_1 - p[0], d2 = eta_2 - p[1], d3 = eta_3 - p[2];
      g_c += std::exp(-0.5 * s2c * (d1 * d1 + d2 * d2 + 2. * d3 * d3));
    }
    for (auto &p : secondary_peaks) {
      const double d1 = eta_1 - p[0], d2 = eta_2 - p[1], d3 = eta_3 - p[2];
      g_s += std::exp(-0.5 * s2s * (d1 * d1 + d2 * d2 + 2. * d3 * d3));
    }
    return p_c * std::pow(s2c, 1.5) * g_c + (1. - p_c) * std::pow(s2s, 1.5) * g_s;
  }
};

// quenchedschwingerconditionedfineaction.cc:81-134 on a state whose coarse links have been halved onto the fine links
void schwinger_gauss_fill(const ActionO &F, const ActionO &Cc, const GaussianFillinO &gd, const DevRng &rng, double *tp) {
  const Grid2 &g = F.g, &gc = Cc.g;
  for (int i = 0; i < g.Mt / 2; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      double dt, dx;
      rng.uniforms((uint32_t)(j * gc.Mt + i), P_FILLIN, 0, dt, dx);
      dt = (2. * dt - 1.) * kPi;
      dx = (2. * dx - 1.) * kPi;
      tp[g.link(2 * i, 2 * j, 0)] = wrap_2pi(tp[g.link(2 * i, 2 * j, 0)] + dt);
      tp[g.link(2 * i + 1, 2 * j, 0)] = wrap_2pi(tp[g.link(2 * i + 1, 2 * j, 0)] - dt);
      tp[g.link(2 * i, 2 * j, 1)] = wrap_2pi(tp[g.link(2 * i, 2 * j, 1)] + dx);
      tp[g.link(2 * i, 2 * j + 1, 1)] = wrap_2pi(tp[g.link(2 * i, 2 * j + 1, 1)] - dx);
    }
  for (int i = 0; i < g.Mt / 2; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      const double phi_12 = wrap_2pi(+tp[g.link(2 * i, 2 * j + 1, 1)] + tp[g.link(2 * i, 2 * j + 2, 0)]);
      const double phi_23 = wrap_2pi(+tp[g.link(2 * i + 1, 2 * j + 2, 0)] - tp[g.link(2 * i + 2, 2 * j + 1, 1)]);
      const double phi_34 = wrap_2pi(-tp[g.link(2 * i + 2, 2 * j, 1)] - tp[g.link(2 * i + 1, 2 * j, 0)]);
      const double phi_41 = wrap_2pi(-tp[g.link(2 * i, 2 * j, 0)] + tp[g.link(2 * i, 2 * j, 1)]);
      double th[4];
      gd.dev_draw(rng, (uint32_t)(j * gc.Mt + i), phi_12, phi_23, phi_34, phi_41, th);
      tp[g.link(2 * i, 2 * j + 1, 0)] = +th[0];
      tp[g.link(2 * i + 1, 2 * j + 1, 1)] = -th[1];
      tp[g.link(2 * i + 1, 2 * j + 1, 0)] = -th[2];
      tp[g.link(2 * i + 1, 2 * j, 1)] = +th[3];
    }
}
// quenchedschwingerconditionedfineaction.cc:293-327
double schwinger_gauss_cfa(const ActionO &F, const GaussianFillinO &gd, const double *x) {
  const Grid2 &g = F.g;
  double S = 0.0;
  for (int i = 0; i < g.Mt / 2; ++i)
    for (int j = 0; j < g.Mx / 2; ++j) {
      const double phi_12 = wrap_2pi(+x[g.link(2 * i, 2 * j + 1, 1)] + x[g.link(2 * i, 2 * j + 2, 0)]);
      const double phi_23 = wrap_2pi(+x[g.link(2 * i + 1, 2 * j + 2, 0)] - x[g.link(2 * i + 2, 2 * j + 1, 1)]);
      const double phi_34 = wrap_2pi(-x[g.link(2 * i + 2, 2 * j, 1)] - x[g.link(2 * i + 1, 2 * j, 0)]);
      const double phi_41 = wrap_2pi(-x[g.link(2 * i, 2 * j, 0)] + x[g.link(2 * i, 2 * j, 1)]);
      const double theta_1 = wrap_2pi(+x[g.link(2 * i, 2 * j + 1, 0)]), theta_2 = wrap_2pi(-x[g.link(2 * i + 1, 2 * j + 1, 1)]);
      const double theta_3 = wrap_2pi(-x[g.link(2 * i + 1, 2 * j + 1, 0)]), theta_4 = wrap_2pi(+x[g.link(2 * i + 1, 2 * j, 1)]);
      S -= std::log(gd.evaluate(theta_1, theta_2, theta_3, theta_4, phi_12, phi_23, phi_34, phi_41));
    }
  return S;
}

int dev_schwinger_twolevel_draw(const ActionO &F, const ActionO &Cc, const double *phi_coarse, double *theta,
                                const DevRng &rng, double *terms, int cfa_kind = 0) {
  const Grid2 &g = F.g, &gc = Cc.g;
  const int rt = g.Mt / gc.Mt, rx = g.Mx / gc.Mx;
  const unsigned nf = 2u * g.Mt * g.Mx, nc = 2u * gc.Mt * gc.Mx;
  std::vector<double> tp(nf, 0.0), thetaC(nc);
  schwinger_copy_from_coarse(gc.Mt, gc.Mx, rt, rx, phi_coarse, tp.data());
  auto dtheta = [&](int i, int j) {  // coarse cell (i, j)
    double u, v;
    rng.uniforms((uint32_t)(j * gc.Mt + i), P_FILLIN, 0, u, v);
    return (2. * u - 1.) * kPi;
  };
  auto draw = [&](unsigned l, double x_p, double x_m) {  // expcosdistribution.hh:51-65 with the device sampler
    const double dx = x_m - x_p;
    const double tau = 2. * F.beta * std::fabs(std::cos(0.5 * dx));
    const double x = dev_vonmises(rng, l, tau, kVmFillin);
    return wrap_2pi(x + 0.5 * (x_p + x_m) + (std::fabs(dx) > kPi ? kPi : 0.0));
  };
  if (rt == 2 && rx == 1) {
    for (int i = 0; i < g.Mt / 2; ++i)
      for (int j = 0; j < g.Mx; ++j) {
        const double d = dtheta(i, j);
        tp[g.link(2 * i, j, 0)] = wrap_2pi(tp[g.link(2 * i, j, 0)] + d);
        tp[g.link(2 * i + 1, j, 0)] = wrap_2pi(tp[g.link(2 * i + 1, j, 0)] - d);
      }
    for (int i = 0; i < g.Mt / 2; ++i)
      for (int j = 0; j < g.Mx; ++j) {
        double theta_p = wrap_2pi(tp[g.link(2 * i, j, 1)] + tp[g.link(2 * i, j + 1, 0)] - tp[g.link(2 * i, j, 0)]);
        double theta_m = wrap_2pi(tp[g.link(2 * i + 1, j, 0)] + tp[g.link(2 * i + 2, j, 1)] - tp[g.link(2 * i + 1, j + 1, 0)]);
        const unsigned l = g.link(2 * i + 1, j, 1);
        tp[l] = draw(l, theta_p, theta_m);
      }
  } else if (rt == 1 && rx == 2) {
    for (int i = 0; i < g.Mt; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        const double d = dtheta(i, j);
        tp[g.link(i, 2 * j, 1)] = wrap_2pi(tp[g.link(i, 2 * j, 1)] + d);
        tp[g.link(i, 2 * j + 1, 1)] = wrap_2pi(tp[g.link(i, 2 * j + 1, 1)] - d);
      }
    for (int i = 0; i < g.Mt; ++i)
      for (int j = 0; j < g.Mx / 2; ++j) {
        double theta_p = wrap_2pi(tp[g.link(i, 2 * j, 0)] + tp[g.link(i + 1, 2 * j, 1)] - tp[g.link(i, 2 * j, 1)]);
        double theta_m = wrap_2pi(tp[g.link(i, 2 * j + 1, 1)] + tp[g.link(i, 2 * j + 2, 0)] - tp[g.link(i + 1, 2 * j + 1, 1)]);
        const unsigned l = g.link(i, 2 * j + 1, 0);
        tp[l] = draw(l, theta_p, theta_m);
      }
  } else if (rt == 2 && rx == 2 && cfa_kind == 1) {  // QuenchedSchwingerGaussianConditionedFineAction
    const GaussianFillinO gd(F.beta);
    schwinger_gauss_fill(F, Cc, gd, rng, tp.data());
    double dS_fine = F.evaluate(tp.data()) - F.evaluate(theta);
    schwinger_copy_from_fine(gc.Mt, gc.Mx, rt, rx, theta, thetaC.data());
    double dS_coarse = Cc.evaluate(thetaC.data()) - Cc.evaluate(phi_coarse);
    double dS_trial = schwinger_gauss_cfa(F, gd, theta) - schwinger_gauss_cfa(F, gd, tp.data());
    return twolevel_decide(dS_fine, dS_coarse, dS_trial, rng, tp, theta, terms);
  } else if (rt == 2 && rx == 2) {
    // quenchedschwingerconditionedfineaction.hh:62-71: true distribution up to beta = 8, approximation beyond
    static std::unique_ptr<BesselProductO> cached;  // the coefficient table costs ~10 ms to build
    const BesselProductO *bp_ptr = nullptr;
    if (!(F.beta > 8.0)) {
      if (!cached || cached->beta != F.beta) cached.reset(new BesselProductO(F.beta));
      bp_ptr = cached.get();
    }
    struct { const BesselProductO *p; const BesselProductO *get() const { return p; } } bp{bp_ptr};
    schwinger_both_fill(F, Cc, bp.get(), rng, tp.data());
    double dS_fine = F.evaluate(tp.data()) - F.evaluate(theta);
    schwinger_copy_from_fine(gc.Mt, gc.Mx, rt, rx, theta, thetaC.data());
    double dS_coarse = Cc.evaluate(thetaC.data()) - Cc.evaluate(phi_coarse);
    double dS_trial = schwinger_both_cfa(F, bp.get(), theta) - schwinger_both_cfa(F, bp.get(), tp.data());
    return twolevel_decide(dS_fine, dS_coarse, dS_trial, rng, tp, theta, terms);
  } else {
    return -1;  // "invalid coarsening for fill-in"
  }
  double dS_fine = F.evaluate(tp.data()) - F.evaluate(theta);
  schwinger_copy_from_fine(gc.Mt, gc.Mx, rt, rx, theta, thetaC.data());
  double dS_coarse = Cc.evaluate(thetaC.data()) - Cc.evaluate(phi_coarse);
  double dS_trial = schwinger_semi_cfa(F, rt, theta) - schwinger_semi_cfa(F, rt, tp.data());
  return twolevel_decide(dS_fine, dS_coarse, dS_trial, rng, tp, theta, terms);
}

// Exact draw of the Gaussian free field by spectral synthesis, device order (include/mlmcpi_hip.h): direct O(N^2)
// evaluation of phi(x) = Re sum_k w_k e^{+i k x} / sqrt(N lambda(k)); only for small lattices.  sub = 0: draws,
// sub = 1: initialise_state (gffaction.cc:121-123: the initial state is an exact draw).
void dev_gff_exact(const ActionO &A, double *phi, uint64_t seed, uint32_t chain, uint32_t step, uint32_t sub) {
  const int Mt = A.g.Mt, Mx = A.g.Mx, N = Mt * Mx;
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  std::vector<double> wr(N), wi(N);
  for (int l = 0; l < N; ++l) {
    const int kx = l / Mt, kt = l - kx * Mt;
    const double lambda = 4.0 + A.gff_mu2 - 2.0 * std::cos(2. * kPi * kt / Mt) - 2.0 * std::cos(2. * kPi * kx / Mx);
    double n0, n1;
    r.normals((uint32_t)l, P_EXACT, sub, n0, n1);
    const double s = std::sqrt(1.0 / ((double)N * lambda));
    wr[l] = s * n0;
    wi[l] = s * n1;
  }
  for (int j = 0; j < Mx; ++j)
    for (int i = 0; i < Mt; ++i) {
      double acc = 0.0;
      for (int l = 0; l < N; ++l) {
        const int kx = l / Mt, kt = l - kx * Mt;
        const double ph = 2. * kPi * ((double)((kt * i) % Mt) / Mt + (double)((kx * j) % Mx) / Mx);
        acc += wr[l] * std::cos(ph) - wi[l] * std::sin(ph);
      }
      phi[j * Mt + i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Reference-order samplers.
// ---------------------------------------------------------------------------------------------
// montecarlo/mcmcstep.hh:21-72
struct StepCounters {
  unsigned n_total = 0, n_accepted = 0;
  bool accept = false;
  double p_accept() const { return n_accepted / (1. * n_total); }
  void reset() { n_total = n_accepted = 0; }
};

// sampler/hmcsampler.hh:84-109, sampler/hmcsampler.cc:8-113
struct HmcO : StepCounters {
  ActionO *A;
  unsigned nt, n_rep;
  double dt;
  std::mt19937_64 engine{8923759};
  std::normal_distribution<double> normal{0.0, 1.0};
  std::uniform_real_distribution<double> uniform{0.0, 1.0};
  std::vector<double> cur, p, trial, dp;
  int tuned = -1;  // -1 not run, 0 failed, 1 converged

  HmcO(ActionO *A_, unsigned nt_, double dt_, unsigned n_rep_, unsigned n_burnin, int autotune,
       unsigned tune_iters, unsigned tune_samples)
      : A(A_), nt(nt_), n_rep(n_rep_), dt(dt_) {
    unsigned n = A->size();
    cur.assign(n, 0.0); p.assign(n, 0.0); trial.assign(n, 0.0); dp.assign(n, 0.0);
    A->initialise_ref(cur.data());
    std::vector<double> tmp(n, 0.0);
    for (unsigned i = 0; i < n_burnin; ++i) draw(tmp.data());
    if (autotune) tune(0.8, tune_iters, tune_samples);
    reset();
  }

  bool single_step() {
    unsigned n = A->size();
    for (unsigned l = 0; l < n; ++l) p[l] = normal(engine);
    double T0 = 0;
    for (unsigned l = 0; l < n; ++l) T0 += p[l] * p[l];
    T0 *= 0.5;
    trial = cur;
    for (unsigned k = 0; k <= nt; ++k) {
      double dtp = (k == 0 || k == nt) ? 0.5 * dt : dt;
      double dtx = (k == nt) ? 0.0 : dt;
      A->force(trial.data(), dp.data());
      for (unsigned l = 0; l < n; ++l) p[l] -= dtp * dp[l];
      for (unsigned l = 0; l < n; ++l) trial[l] += dtx * p[l];
    }
    double T1 = 0;
    for (unsigned l = 0; l < n; ++l) T1 += p[l] * p[l];
    T1 *= 0.5;
    double dH = (A->evaluate(trial.data()) - A->evaluate(cur.data())) + (T1 - T0);
    bool ok = dH < 0.0 ? true : (uniform(engine) < std::exp(-dH));
    if (ok) cur = trial;
    return ok;
  }

  void draw(double *out) {
    accept = false;
    for (unsigned r = 0; r < n_rep; ++r) accept = accept || single_step();  // short-circuit (F11)
    ++n_total;
    n_accepted += accept ? 1 : 0;
    if (accept) std::copy(cur.begin(), cur.end(), out);  // copy_if_rejected == false
  }

  // hmcsampler.cc:77-113 (100 bisection iterations x 1000 steps in the reference)
  void tune(double target, unsigned iters, unsigned samples) {
    double dt0 = dt, lo = 0.5 * dt, hi = 2. * dt;
    bool converged = false;
    for (unsigned k = 0; k < iters; ++k) {
      reset();
      dt = 0.5 * (lo + hi);
      for (unsigned j = 0; j < samples; ++j) {
        n_accepted += single_step() ? 1 : 0;
        ++n_total;
      }
      if (p_accept() > target) lo = dt; else hi = dt;
      if (std::fabs(p_accept() - target) < 1.E-2) converged = true;
    }
    if (!converged) dt = dt0;
    tuned = converged ? 1 : 0;
    reset();
  }
};

// sampler/overrelaxedheatbathsampler.hh:102-129, sampler/overrelaxedheatbathsampler.cc:8-37
struct HeatBathO : StepCounters {
  ActionO *A;
  unsigned n_hb, n_or;
  bool random_order;
  std::mt19937_64 engine{871417};
  std::vector<unsigned> order;
  std::vector<double> cur;

  HeatBathO(ActionO *A_, unsigned n_hb_, unsigned n_or_, unsigned n_burnin, bool random_)
      : A(A_), n_hb(n_hb_), n_or(n_or_), random_order(random_) {
    order.resize(A->size());
    std::iota(order.begin(), order.end(), 0u);
    cur.assign(A->size(), 0.0);
    A->initialise_ref(cur.data());
    std::vector<double> tmp(A->size());
    for (unsigned i = 0; i < n_burnin; ++i) draw(tmp.data());
    reset();
  }

  void draw(double *out) {
    for (unsigned s = 0; s < n_or; ++s) {
      if (random_order) std::shuffle(order.begin(), order.end(), engine);
      for (unsigned l : order) A->overrelax(cur.data(), l);
    }
    for (unsigned s = 0; s < n_hb; ++s) {
      if (random_order) std::shuffle(order.begin(), order.end(), engine);
      for (unsigned l : order) A->heatbath_ref(cur.data(), l);
    }
    accept = true;
    ++n_total;
    ++n_accepted;
    std::copy(cur.begin(), cur.end(), out);
  }
};

// ---------------------------------------------------------------------------------------------
// Statistics (single rank).  common/statistics.hh:104-211, common/statistics.cc:4-95.
// ---------------------------------------------------------------------------------------------
struct StatsO {
  unsigned k_max, n = 0, n_long = 0;
  std::deque<double> window;
  std::vector<double> S;
  double avg = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  explicit StatsO(unsigned k) : k_max(k), S(k, 0.0) {}
  void reset() { n = 0; avg = 0; }
  void hard_reset() {
    reset();
    window.clear();
    S.assign(k_max, 0.0);
    a1 = a2 = a3 = a4 = 0;
    n_long = 0;
  }
  void record(double Q) {
    ++n; ++n_long;
    window.push_front(Q);
    if (window.size() > k_max) window.pop_back();
    avg = ((n - 1.0) * avg + Q) / (1.0 * n);
    a1 = ((n_long - 1.0) * a1 + Q) / (1.0 * n_long);
    a2 = ((n_long - 1.0) * a2 + Q * Q) / (1.0 * n_long);
    a3 = ((n_long - 1.0) * a3 + Q * Q * Q) / (1.0 * n_long);
    a4 = ((n_long - 1.0) * a4 + Q * Q * Q * Q) / (1.0 * n_long);
    for (unsigned k = 0; k < window.size(); ++k) {
      unsigned Nk = n_long - k;
      S[k] = ((Nk - 1.0) * S[k] + window[0] * window[k]) / (1.0 * Nk);
    }
  }
  double variance() const { return 1.0 * n_long / (n_long - 1.0) * (S[0] - a1 * a1); }
  double variance_error() const {
    return std::sqrt(1.0 / n_long * (a4 - 4 * a1 * a3 + 8 * a1 * a1 * a2 - a2 * a2 - 4 * a1 * a1 * a1 * a1));
  }
  double tau_int() const {
    double acc = 0.0, c0 = S[0] - a1 * a1;
    for (unsigned k = 1; k < S.size(); ++k) acc += (1. - k / (1.0 * n_long)) * (S[k] - a1 * a1);
    return std::fmax(1.0, 1.0 + 2.0 * acc / c0);
  }
  double error() const { return std::sqrt(tau_int() * variance() / (1.0 * n)); }
};

// ---------------------------------------------------------------------------------------------
// Multilevel callers in REFERENCE ORDER (1-D actions): sequential, std::mt19937_64 engines with the reference's seeds
// (+ a replica offset, as the reference's MPI ranks get distinct seeds), one object per class of the reference.
// Purpose: to see what the reference's scheme itself does at a low hierarchical acceptance -- the
// delayed-acceptance two-level step fed with coarse samples taken ceil(2 tau_int) draws apart, tau_int re-read on
// every coarse draw.  Not timed, not shipped.
// ---------------------------------------------------------------------------------------------
void *orc_action_1d_impl(int kind, unsigned M, double T_final, double m0, double mu2, double lambda, double x0) {
  uint64_t seed = kind == ROTOR ? 21172817ull : 124129017ull;
  ActionO *A = new ActionO((Kind)kind, seed);
  A->M = M; A->T_final = T_final; A->a = T_final / M; A->m0 = m0; A->mu2 = mu2; A->lambda = lambda; A->x0 = x0;
  return A;
}
struct SamplerO {  // sampler/sampler.hh:18-44 + montecarlo/mcmcstep.hh:21-72
  virtual ~SamplerO() {}
  virtual void draw(double *out) = 0;
  virtual void set_state(const double *x) = 0;
  virtual bool accepted() const = 0;
};
struct HmcSamplerO : SamplerO {  // HMCSampler behind the Sampler interface
  HmcO h;
  HmcSamplerO(ActionO *A, unsigned nt, double dt, unsigned n_burnin, uint64_t seed_offset) : h(A, nt, dt, 1, 0, 0, 0, 0) {
    h.engine.seed(8923759ull + seed_offset);
    std::vector<double> tmp(A->size(), 0.0);
    for (unsigned i = 0; i < n_burnin; ++i) h.draw(tmp.data());
    h.reset();
  }
  void draw(double *out) override { h.draw(out); }
  void set_state(const double *x) override { std::copy(x, x + h.cur.size(), h.cur.begin()); }  // hmcsampler.cc:72-74
  bool accepted() const override { return h.accept; }
};
// montecarlo/twolevelmetropolisstep.{hh,cc} with GaussianConditionedFineAction (action/qm/gaussianconditionedfineaction.cc:7-43)
// and QMAction::copy_from_coarse / copy_from_fine (action/qm/qmaction.cc:7-24).  (The reference's constructor also times
// 10000 draws from a zero coarse state; that only advances the engines and is left out.)
struct TwoLevelStepO : StepCounters {
  ActionO *F, *C;
  std::mt19937_64 engine, cfa_engine;
  std::uniform_real_distribution<double> uniform{0.0, 1.0};
  std::normal_distribution<double> cfa_normal{0.0, 1.0};
  std::vector<double> theta, thetaC, prime;
  double S_theta = 0, cfa_theta = 0;
  TwoLevelStepO(ActionO *F_, ActionO *C_, uint64_t seed_offset)
      : F(F_), C(C_), engine(89216491ull + seed_offset), cfa_engine(11897197ull + seed_offset), theta(F_->M, 0.0), thetaC(C_->M, 0.0),
        prime(F_->M, 0.0) {
    S_theta = F->evaluate(theta.data());
    cfa_theta = cfa_gaussian(*F, theta.data());
  }
  void set_state(const double *x) {  // twolevelmetropolisstep.cc:92-97
    std::copy(x, x + theta.size(), theta.begin());
    S_theta = F->evaluate(theta.data());
    cfa_theta = cfa_gaussian(*F, theta.data());
  }
  void draw(const double *coarse, double *out) {  // twolevelmetropolisstep.cc:35-89
    const unsigned M = F->M, Mc = M / 2;
    for (unsigned j = 0; j < Mc; ++j) prime[2 * j] = coarse[j];
    for (unsigned j = 0; j < Mc - 1; ++j) {  // interior points, then the one that wraps around
      const double xm = prime[2 * j], xp = prime[2 * (j + 1)];
      prime[2 * j + 1] = F->w_minimum(xm, xp) + cfa_normal(cfa_engine) * (1. / std::sqrt(F->w_curvature(xm, xp)));
    }
    {
      const double xm = prime[M - 2], xp = prime[0];
      prime[M - 1] = F->w_minimum(xm, xp) + cfa_normal(cfa_engine) * (1. / std::sqrt(F->w_curvature(xm, xp)));
    }
    const double S_prime = F->evaluate(prime.data());
    const double dS_fine = S_prime - S_theta;
    for (unsigned j = 0; j < Mc; ++j) thetaC[j] = theta[2 * j];
    const double dS_coarse = C->evaluate(thetaC.data()) - C->evaluate(coarse);
    const double cfa_prime = cfa_gaussian(*F, prime.data());
    const double dS = dS_fine + dS_coarse + (cfa_theta - cfa_prime);
    accept = dS < 0.0 ? true : (uniform(engine) < std::exp(-dS));
    if (accept) {
      theta = prime;
      S_theta = S_prime;
      cfa_theta = cfa_prime;
    }
    ++n_total;
    n_accepted += accept ? 1 : 0;
    if (accept) std::copy(theta.begin(), theta.end(), out);  // copy_if_rejected == false
  }
};
// sampler/hierarchicalsampler.cc:8-81 on the actions acts[top .. L): restrict, one draw of the coarsest-level sampler, two-level
// steps up, break at the first rejection
struct HierarchicalO : SamplerO, StepCounters {
  std::vector<ActionO *> act;                        // act[0] = the sampler's own (finest) level
  std::vector<std::unique_ptr<TwoLevelStepO>> step;  // step[l]: act[l + 1] -> act[l]
  std::unique_ptr<HmcSamplerO> coarse;
  std::vector<std::vector<double>> state;
  HierarchicalO(const std::vector<ActionO *> &acts, unsigned top, unsigned nt, double dt, uint64_t seed_offset) {
    for (unsigned l = top; l < acts.size(); ++l) act.push_back(acts[l]);
    for (unsigned l = 0; l + 1 < act.size(); ++l) step.emplace_back(new TwoLevelStepO(act[l], act[l + 1], seed_offset + 1000003ull * (top + 1) + 7ull * l));
    for (ActionO *A : act) state.emplace_back(A->M, 0.0);
    coarse.reset(new HmcSamplerO(act.back(), nt, dt, 0, seed_offset + 1000003ull * (top + 1)));
  }
  void draw(double *out) override {
    const int n = (int)act.size();
    accept = true;
    for (int l = 1; l < n; ++l)
      for (unsigned j = 0; j < act[l]->M; ++j) state[l][j] = state[l - 1][2 * j];
    for (int l = n - 1; l >= 0; --l) {
      if (l == n - 1) {
        coarse->set_state(state[l].data());
        coarse->draw(state[l].data());
        accept = accept && coarse->accepted();
      } else {
        step[l]->set_state(state[l].data());
        step[l]->draw(state[l + 1].data(), state[l].data());
        accept = accept && step[l]->accept;
      }
      if (!accept) break;
    }
    ++n_total;
    n_accepted += accept ? 1 : 0;
    if (accept) std::copy(state[0].begin(), state[0].end(), out);
  }
  void set_state(const double *x) override { std::copy(x, x + state[0].size(), state[0].begin()); }
  bool accepted() const override { return accept; }
};
// montecarlo/montecarlomultilevel.cc:71-204 with sampler = 'hierarchical': burn-in, then n_samples Y samples per level (the
// reference adapts the numbers to a tolerance; fixed here), coarse samples through draw_coarse_sample (:170-190).
// sub_mode 0: the reference -- ceil(2 tau_int) of the coarse sampler's QoI, re-read on every coarse draw (window k_max);
// sub_mode n > 0: n draws apart, fixed (experiment).  QoI = QoIXsquared.
struct MlmcRefO {
  std::vector<std::unique_ptr<ActionO>> owned;
  std::vector<ActionO *> act;
  std::vector<std::unique_ptr<TwoLevelStepO>> step;
  std::vector<std::unique_ptr<HierarchicalO>> sampler;  // sampler[l]: for act[l + 1]
  std::vector<StatsO> stats_Y, stats_sampler, stats_fine, stats_coarse;   // Y = fine - coarse part of the QoI, and the parts
  std::vector<std::vector<double>> phi, phi_coarse;
  std::vector<double> t_indep;
  std::vector<unsigned> n_indep, t_sampler;
  unsigned L, sub_mode;
  static double xsq(const std::vector<double> &x) {  // qoi/qm/qoixsquared.cc:7-20
    double s = 0;
    for (double v : x) s += v * v;
    return s / x.size();
  }
  MlmcRefO(int kind, unsigned M, double T_final, double m0, double mu2, double lambda, double x0, unsigned n_level, unsigned nt, double dt,
           unsigned window, unsigned sub_mode_, uint64_t seed_offset)
      : L(n_level), sub_mode(sub_mode_) {
    for (unsigned l = 0; l < L; ++l) {
      ActionO *A = (ActionO *)orc_action_1d_impl(kind, M >> l, T_final, m0, mu2, lambda, x0);
      owned.emplace_back(A);
      act.push_back(A);
    }
    for (unsigned l = 0; l + 1 < L; ++l) {
      step.emplace_back(new TwoLevelStepO(act[l], act[l + 1], seed_offset + 500009ull * (l + 1)));
      sampler.emplace_back(new HierarchicalO(act, l + 1, nt, dt, seed_offset));
      stats_sampler.emplace_back(window);
    }
    for (unsigned l = 0; l < L; ++l) {
      stats_Y.emplace_back(window);
      stats_fine.emplace_back(window);
      stats_coarse.emplace_back(window);
      phi.emplace_back(act[l]->M, 0.0);
      phi_coarse.emplace_back(act[l]->M, 0.0);
    }
    t_indep.assign(L, 0.0);
    n_indep.assign(L, 0);
    t_sampler.assign(L, 0);
  }
  void draw_coarse_sample(unsigned level, std::vector<double> &x) {  // montecarlomultilevel.cc:170-190
    StatsO &st = stats_sampler[level - 1];
    for (;;) {
      const double need = sub_mode ? (double)sub_mode : std::ceil(2. * st.tau_int());   // re-read every time round, as the while loop does
      if (!(t_sampler[level - 1] < need)) break;
      sampler[level - 1]->draw(x.data());
      st.record(xsq(x));
      ++t_sampler[level - 1];
    }
    t_indep[level - 1] = (n_indep[level - 1] * t_indep[level - 1] + t_sampler[level - 1]) / (1.0 + n_indep[level - 1]);
    ++n_indep[level - 1];
    t_sampler[level - 1] = 0;
  }
  double sample_Y(int level, bool sub_sample) {
    if (level == (int)L - 1) {
      if (sub_sample) draw_coarse_sample(level, phi[level]); else sampler[level - 1]->draw(phi[level].data());
      return xsq(phi[level]);
    }
    if (sub_sample) draw_coarse_sample(level + 1, phi_coarse[level + 1]); else sampler[level]->draw(phi_coarse[level + 1].data());
    step[level]->draw(phi_coarse[level + 1].data(), phi[level].data());
    const double qf = xsq(phi[level]), qc = xsq(phi_coarse[level + 1]);
    if (sub_sample) {
      stats_fine[level].record(qf);
      stats_coarse[level].record(qc);
    }
    return qf - qc;
  }
  // only_level >= 0: that level alone (the levels are independent estimators)
  void run(unsigned n_burnin, unsigned n_samples, int only_level = -1) {
    for (unsigned l = 0; l < L; ++l) stats_Y[l].hard_reset();
    for (int level = (int)L - 1; level >= 0; --level)   // :83-101 (burn-in draws are not sub-sampled)
      for (unsigned j = 0; j < n_burnin && (only_level < 0 || only_level == level); ++j) stats_Y[level].record(sample_Y(level, false));
    for (unsigned l = 0; l < L; ++l) stats_Y[l].reset();
    for (unsigned l = 0; l + 1 < L; ++l) stats_sampler[l].reset();
    for (int level = (int)L - 1; level >= 0; --level)
      for (unsigned j = 0; j < n_samples && (only_level < 0 || only_level == level); ++j) stats_Y[level].record(sample_Y(level, true));
  }
};

}  // namespace

// =============================================================================================
// C surface for ctypes (tests, smoke, cpu_baseline).
// =============================================================================================

// =================================================================================================================
// Gaussian free field on the levels of a coarsening hierarchy: GFFAction with Gibbs smoothing
// (action/qft/gffaction.{hh,cc}), GFFConditionedFineAction (action/qft/gffconditionedfineaction.cc:7-49), the vertex
// lists of Lattice2D (lattice/lattice2d.cc:82-134) and the two-level step between two GFF levels
// (montecarlo/twolevelmetropolisstep.cc:35-89).  Dense matrices as in the reference (gffaction.cc:126-173), with
// Gauss-Jordan inverses -- deliberately not the Cholesky route of the library under test.
// =================================================================================================================
constexpr uint32_t P_GFF_GIBBS = 11, P_GFF_EXACT = 12;

typedef std::vector<double> Mat;  // row major n x n

static Mat mat_inverse(Mat A, unsigned n) {  // Gauss-Jordan with partial pivoting
  Mat I((size_t)n * n, 0.0);
  for (unsigned i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
  for (unsigned c = 0; c < n; ++c) {
    unsigned piv = c;
    for (unsigned r = c + 1; r < n; ++r)
      if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
    if (piv != c)
      for (unsigned k = 0; k < n; ++k) {
        std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]);
        std::swap(I[(size_t)c * n + k], I[(size_t)piv * n + k]);
      }
    const double inv = 1.0 / A[(size_t)c * n + c];
    for (unsigned k = 0; k < n; ++k) { A[(size_t)c * n + k] *= inv; I[(size_t)c * n + k] *= inv; }
    for (unsigned r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = A[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (unsigned k = 0; k < n; ++k) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; I[(size_t)r * n + k] -= f * I[(size_t)c * n + k]; }
    }
  }
  return I;
}
static Mat mat_mul(const Mat &A, const Mat &B, unsigned n) {
  Mat C((size_t)n * n, 0.0);
  for (unsigned i = 0; i < n; ++i)
    for (unsigned k = 0; k < n; ++k) {
      const double a = A[(size_t)i * n + k];
      if (a != 0.0)
        for (unsigned j = 0; j < n; ++j) C[(size_t)i * n + j] += a * B[(size_t)k * n + j];
    }
  return C;
}

struct GffLevelO {
  Grid2 g;
  int ctype, level, n_gibbs;
  double mass, mu2, omega;
  unsigned N;
  std::vector<unsigned> nb, fineonly, pairs;  // neighbour table [8 N]; lattice2d.cc:82-134
  Grid2 gc{0, 0, false};
  Mat Qhat, Lt_inv_T;  // smoothed precision matrix; L^-1 of the Cholesky factor Q = L L^T (so that phi = L^-T psi)

  GffLevelO(unsigned Mt, unsigned Mx, int ctype_, int level_, double mass_, int n_gibbs_, double omega_)
      : g{(int)Mt, (int)Mx, ctype_ == 4 && (level_ % 2)}, ctype(ctype_), level(level_), n_gibbs(n_gibbs_), mass(mass_), omega(omega_) {
    N = g.nvertices();
    const double a_lat = g.rotated ? std::sqrt(2.) / Mt : 1. / Mt;  // gffaction.hh:174-181
    mu2 = a_lat * a_lat * mass * mass;
    nb.resize(8 * (size_t)N);
    for (unsigned l = 0; l < N; ++l) g.neighbours(l, &nb[8 * (size_t)l]);
    // lattice2d.cc:20-134
    int rt = 1, rx = 1;
    bool ok = true;
    switch (ctype) {
      case 0: rt = rx = 2; break;
      case 1: rt = 2; break;
      case 2: rx = 2; break;
      case 3: (level % 2 == 0 ? rt : rx) = 2; break;
      case 4: if (g.rotated) { rt = rx = 2; ok = !((Mt % 2) || (Mx % 2)); } break;
      default: ok = false;
    }
    unsigned mt = Mt, mx = Mx;
    if (rt > 1) { if (Mt % rt) ok = false; mt = Mt / rt; }
    if (rx > 1) { if (Mx % rx) ok = false; mx = Mx / rx; }
    if (ok && mt > 1 && mx > 1) {
      gc = Grid2{(int)mt, (int)mx, ctype == 4 && ((level + 1) % 2)};
      for (int i = 0; i < (int)Mt; ++i)
        for (int j = 0; j < (int)Mx; ++j) {
          bool coarse;
          if (ctype == 4) {
            if (g.rotated) { if ((i + j) % 2) continue; coarse = (i % 2 == 0) && (j % 2 == 0); }
            else coarse = (i + j) % 2 == 0;
          } else {
            coarse = (i % rt == 0) && (j % rx == 0);
          }
          (coarse ? pairs : fineonly).push_back(g.vertex(i, j));
          if (coarse) pairs.push_back(gc.vertex(i / rt, j / rx));
        }
      // the reference sorts both lists (lattice2d.cc:121-122) and its map iterates in key order
      std::sort(fineonly.begin(), fineonly.end());
      std::vector<std::pair<unsigned, unsigned>> pv;
      for (size_t k = 0; k < pairs.size(); k += 2) pv.push_back({pairs[k], pairs[k + 1]});
      std::sort(pv.begin(), pv.end());
      for (size_t k = 0; k < pv.size(); ++k) { pairs[2 * k] = pv[k].first; pairs[2 * k + 1] = pv[k].second; }
    }
  }

  Mat precision(const double *stencil, int shells) const {  // gffaction.cc:176-197
    Mat Q((size_t)N * N, 0.0);
    for (unsigned l = 0; l < N; ++l) {
      Q[(size_t)l * N + l] += stencil[0];
      for (int s = 0; s < shells; ++s)
        for (int k = 0; k < 4; ++k) Q[(size_t)l * N + nb[8 * (size_t)l + 4 * s + k]] += stencil[s + 1];
    }
    return Q;
  }
  void build() {  // gffaction.cc:126-173
    if (!Qhat.empty()) return;
    const double st[2] = {4. + mu2, -1.};
    const Mat Q = precision(st, 1);
    const double h = 4. + 0.5 * mu2;
    const double st_eff[3] = {h - 4. / h, -2. / h, -1. / h};
    const Mat Qeff = precision(st_eff, 2);
    const Mat Sigma = mat_inverse(Q, N), Sigma_eff = mat_inverse(Qeff, N);
    Mat M((size_t)N * N, 0.0);
    for (unsigned i = 0; i < N; ++i)
      for (unsigned j = 0; j <= i; ++j) M[(size_t)i * N + j] = Qeff[(size_t)i * N + j];
    if (std::fabs(omega - 1.0) > 1e-14)
      for (unsigned i = 0; i < N; ++i) M[(size_t)i * N + i] += (1. / omega - 1.) * Qeff[(size_t)i * N + i];
    Mat G((size_t)N * N, 0.0);
    for (unsigned i = 0; i < N; ++i) G[(size_t)i * N + i] = 1.0;
    if (n_gibbs > 0) {
      Mat Gt = mat_mul(mat_inverse(M, N), Qeff, N);
      for (double &v : Gt) v = -v;
      for (unsigned i = 0; i < N; ++i) Gt[(size_t)i * N + i] += 1.0;
      for (int k = 0; k < n_gibbs; ++k) G = mat_mul(G, Gt, N);
    }
    Mat D(Sigma);
    for (size_t k = 0; k < D.size(); ++k) D[k] -= Sigma_eff[k];
    Mat GT((size_t)N * N);
    for (unsigned i = 0; i < N; ++i)
      for (unsigned j = 0; j < N; ++j) GT[(size_t)j * N + i] = G[(size_t)i * N + j];
    Mat S = mat_mul(mat_mul(G, D, N), GT, N);
    for (size_t k = 0; k < S.size(); ++k) S[k] += Sigma_eff[k];
    Qhat = mat_inverse(S, N);
    // Cholesky factor of Q (textbook recursion), then its inverse
    Mat L((size_t)N * N, 0.0);
    for (unsigned j = 0; j < N; ++j) {
      double d = Q[(size_t)j * N + j];
      for (unsigned k = 0; k < j; ++k) d -= L[(size_t)j * N + k] * L[(size_t)j * N + k];
      L[(size_t)j * N + j] = std::sqrt(d);
      for (unsigned i = j + 1; i < N; ++i) {
        double v = Q[(size_t)i * N + j];
        for (unsigned k = 0; k < j; ++k) v -= L[(size_t)i * N + k] * L[(size_t)j * N + k];
        L[(size_t)i * N + j] = v / L[(size_t)j * N + j];
      }
    }
    Lt_inv_T = mat_inverse(L, N);
  }
  double evaluate(const double *phi) {  // gffaction.cc:8-30
    double S = 0.0;
    if (n_gibbs == 0) {
      for (unsigned l = 0; l < N; ++l) {
        double loc = (4. + mu2) * phi[l];
        for (int k = 0; k < 4; ++k) loc -= phi[nb[8 * (size_t)l + k]];
        S += phi[l] * loc;
      }
    } else {
      build();
      for (unsigned i = 0; i < N; ++i) {
        double y = 0.0;
        for (unsigned j = 0; j < N; ++j) y += Qhat[(size_t)i * N + j] * phi[j];
        S += phi[i] * y;
      }
    }
    return 0.5 * S;
  }
  // gffaction.cc:200-213 with the device's random numbers
  void dev_draw(double *phi, const DevRng &rng) {
    build();
    std::vector<double> psi(N);
    auto normals = [&](Purpose p, uint32_t sub, std::vector<double> &out) {
      for (unsigned q = 0; 2 * q < N; ++q) {
        double n0, n1;
        rng.normals(q, p, sub, n0, n1);
        out[2 * q] = n0;
        if (2 * q + 1 < N) out[2 * q + 1] = n1;
      }
    };
    normals((Purpose)P_GFF_EXACT, 0, psi);
    for (unsigned i = 0; i < N; ++i) {  // solve L^T phi = psi
      double y = 0.0;
      for (unsigned j = i; j < N; ++j) y += Lt_inv_T[(size_t)j * N + i] * psi[j];
      phi[i] = y;
    }
    const double h = 4. + 0.5 * mu2, d0 = h - 4. / h;
    const double sigma_eff = 1. / std::sqrt(d0), kappa = omega / h, gamma = std::sqrt(omega * (2. - omega));
    for (int k = 0; k < n_gibbs; ++k) {  // gffaction.cc:45-66
      normals((Purpose)P_GFF_GIBBS, (uint32_t)k, psi);
      for (unsigned l = 0; l < N; ++l) {
        double Delta = (1. - omega) * d0 * phi[l];
        for (int m = 0; m < 4; ++m) Delta += 2. * kappa * phi[nb[8 * (size_t)l + m]];
        for (int m = 4; m < 8; ++m) Delta += kappa * phi[nb[8 * (size_t)l + m]];
        phi[l] = sigma_eff * (gamma * psi[l] + sigma_eff * Delta);
      }
    }
  }
  // gffconditionedfineaction.cc:7-49
  double cfa(double *phi, bool fill, const DevRng *rng) const {
    const double sigma2 = 1. / (4. + mu2), sigma = std::sqrt(sigma2), sigma2_inv = 1. / sigma2;
    double S = 0.0;
    for (unsigned l : fineonly) {
      double Delta = 0.0;
      for (int k = 0; k < 4; ++k) Delta += phi[nb[8 * (size_t)l + k]];
      if (fill) {
        double n0, n1;
        rng->normals(l, P_FILLIN, 0, n0, n1);
        phi[l] = sigma * (n0 + sigma * Delta);
      }
      const double dphi = phi[l] - sigma2 * Delta;
      S += 0.5 * sigma2_inv * dphi * dphi;
    }
    return S;
  }
};

extern "C" {

void orc_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  Philox4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  std::memcpy(out, r.v, sizeof(r.v));
}

// purpose/sub as in the RNG contract; out[0..1] uniforms, out[2..3] Box-Muller normals
void orc_dev_random(uint64_t seed, uint32_t chain, uint32_t step, uint32_t site, uint32_t purpose,
                    uint32_t sub, double *out) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  r.uniforms(site, (Purpose)purpose, sub, out[0], out[1]);
  r.normals(site, (Purpose)purpose, sub, out[2], out[3]);
}

double orc_mod_2pi(double x) { return wrap_2pi(x); }

// ---- lattice -----------------------------------------------------------------------------------
unsigned orc_vertex_cart2lin(int Mt, int Mx, int rotated, int i, int j) {
  return Grid2{Mt, Mx, rotated != 0}.vertex(i, j);
}
void orc_vertex_lin2cart(int Mt, int Mx, int rotated, unsigned l, int *i, int *j) {
  Grid2{Mt, Mx, rotated != 0}.vertex_inv(l, *i, *j);
}
unsigned orc_link_cart2lin(int Mt, int Mx, int i, int j, int mu) { return Grid2{Mt, Mx, false}.link(i, j, mu); }
void orc_link_lin2cart(int Mt, int Mx, unsigned l, int *i, int *j, int *mu) {
  Grid2{Mt, Mx, false}.link_inv(l, *i, *j, *mu);
}
void orc_neighbours2d(int Mt, int Mx, int rotated, unsigned *out) {
  Grid2 g{Mt, Mx, rotated != 0};
  for (unsigned l = 0; l < g.nvertices(); ++l) g.neighbours(l, out + 8 * l);
}
void orc_neighbours1d(unsigned M, unsigned *out) {  // lattice/lattice1d.cc:12-17
  for (unsigned l = 0; l < M; ++l) {
    out[2 * l] = (l - 1 + M) % M;
    out[2 * l + 1] = (l + 1 + M) % M;
  }
}

// ---- actions -----------------------------------------------------------------------------------
// Engine seeds: rotoraction.hh:106, quenchedschwingeraction.hh:116, gffaction.hh:182,
// harmonicoscillatoraction.hh:100 (HO engine only feeds the exact sampler, unused here).
void *orc_action_1d(int kind, unsigned M, double T_final, double m0, double mu2, double lambda, double x0) {
  return orc_action_1d_impl(kind, M, T_final, m0, mu2, lambda, x0);
}
void *orc_action_gff(int Mt, int Mx, double mass) {  // gffaction.hh:164-185
  ActionO *A = new ActionO(GFF, 2481317ull);
  A->g = Grid2{Mt, Mx, false};
  A->mass = mass;
  double a_lat = 1. / Mt;
  A->gff_mu2 = a_lat * a_lat * mass * mass;
  A->gff_sigma = 1. / std::sqrt(4. + A->gff_mu2);
  return A;
}
void *orc_action_schwinger(int Mt, int Mx, double beta) {
  ActionO *A = new ActionO(SCHWINGER, 2481317ull);
  A->g = Grid2{Mt, Mx, false};
  A->beta = beta;
  return A;
}
void orc_action_free(void *h) { delete (ActionO *)h; }
unsigned orc_action_size(void *h) { return ((ActionO *)h)->size(); }
double orc_action_evaluate(void *h, const double *x) { return ((ActionO *)h)->evaluate(x); }
void orc_action_force(void *h, const double *x, double *f) { ((ActionO *)h)->force(x, f); }
int orc_action_overrelaxation_update(void *h, double *x, unsigned l) { return ((ActionO *)h)->overrelax(x, l) ? 0 : -1; }
int orc_action_heatbath_update(void *h, double *x, unsigned l) { return ((ActionO *)h)->heatbath_ref(x, l) ? 0 : -1; }
void orc_action_initialise_state(void *h, double *x) { ((ActionO *)h)->initialise_ref(x); }
double orc_action_wminimum(void *h, double xm, double xp) { return ((ActionO *)h)->w_minimum(xm, xp); }
double orc_action_wcurvature(void *h, double xm, double xp) { return ((ActionO *)h)->w_curvature(xm, xp); }
double orc_action_gff_mu2(void *h) { return ((ActionO *)h)->gff_mu2; }
void orc_action_staples(void *h, const double *x, unsigned l, double *tp, double *tm) {
  ActionO *A = (ActionO *)h;
  int i, j, mu;
  A->g.link_inv(l, i, j, mu);
  A->staples(x, i, j, mu, *tp, *tm);
}

// ---- reference-order multilevel run (MlmcRefO) and the single-level HMC chain it must agree with -------------------------
// out[level][8] = mean(Y), variance, tau_int, samples, mean draws between coarse samples (t_indep), acceptance of the level's
// two-level step, mean of the fine part of Y, mean of its coarse part; acc[level] = acceptance of the hierarchical sampler feeding level `level` (the sampler of act[level + 1];
// last entry: of the coarsest level's own sampler)
void orc_mlmc_ref_run(int kind, unsigned M, double T_final, double m0, double mu2, double lambda, double x0, unsigned n_level, unsigned nt,
                      double dt, unsigned window, unsigned sub_mode, unsigned n_burnin, unsigned n_samples, int only_level, uint64_t seed_offset,
                      double *out, double *acc) {
  MlmcRefO R(kind, M, T_final, m0, mu2, lambda, x0, n_level, nt, dt, window, sub_mode, seed_offset);
  R.run(n_burnin, n_samples, only_level);
  for (unsigned l = 0; l < n_level; ++l) {
    if (only_level >= 0 && (int)l != only_level) {
      for (int q = 0; q < 8; ++q) out[8 * l + q] = 0.0;
      acc[l] = 0.0;
      continue;
    }
    out[8 * l + 0] = R.stats_Y[l].avg;
    out[8 * l + 1] = R.stats_Y[l].variance();
    out[8 * l + 2] = R.stats_Y[l].tau_int();
    out[8 * l + 3] = R.stats_Y[l].n;
    out[8 * l + 4] = l + 1 < n_level ? R.t_indep[l] : R.t_indep[n_level - 2];
    out[8 * l + 5] = l + 1 < n_level ? R.step[l]->p_accept() : 1.0;
    out[8 * l + 6] = l + 1 < n_level ? R.stats_fine[l].avg : R.stats_Y[l].avg;
    out[8 * l + 7] = l + 1 < n_level ? R.stats_coarse[l].avg : 0.0;
    acc[l] = R.sampler[l + 1 < n_level ? l : n_level - 2]->p_accept();
  }
}
// single-level chains of the same action for comparison: HMC (hmcsampler.cc) or the hierarchical sampler ON ITS OWN
// (hierarchicalsampler.cc: exact by construction -- delayed acceptance from the restricted current state).
// out = mean, variance, tau_int, samples, acceptance
void orc_single_level_ref_run(int kind, unsigned M, double T_final, double m0, double mu2, double lambda, double x0, unsigned n_level,
                              unsigned nt, double dt, unsigned window, unsigned n_burnin, unsigned n_samples, uint64_t seed_offset, double *out) {
  std::vector<std::unique_ptr<ActionO>> owned;
  std::vector<ActionO *> act;
  for (unsigned l = 0; l < n_level; ++l) {
    owned.emplace_back((ActionO *)orc_action_1d_impl(kind, M >> l, T_final, m0, mu2, lambda, x0));
    act.push_back(owned.back().get());
  }
  std::unique_ptr<SamplerO> S;
  StepCounters *cnt;
  if (n_level == 1) {
    auto *h = new HmcSamplerO(act[0], nt, dt, 0, seed_offset);
    S.reset(h);
    cnt = &h->h;
  } else {
    auto *h = new HierarchicalO(act, 0, nt, dt, seed_offset);
    S.reset(h);
    cnt = h;
  }
  StatsO st(window);
  std::vector<double> x(M, 0.0);
  for (unsigned j = 0; j < n_burnin; ++j) S->draw(x.data());
  cnt->reset();
  for (unsigned j = 0; j < n_samples; ++j) {
    S->draw(x.data());
    st.record(MlmcRefO::xsq(x));
  }
  out[0] = st.avg; out[1] = st.variance(); out[2] = st.tau_int(); out[3] = st.n; out[4] = cnt->p_accept();
}

// rejection samplers with the reference engine of a fresh RefRng (seeded) -- for distribution tests
void orc_expcos_draws(uint64_t seed, double beta, double x_p, double x_m, unsigned n, double *out) {
  RefRng r(seed);
  RefAttemptSource s{r};
  for (unsigned k = 0; k < n; ++k) out[k] = expcos_draw(s, beta, x_p, x_m);
}
void orc_expsin2_draws(uint64_t seed, double sigma, unsigned n, double *out) {
  RefRng r(seed);
  RefAttemptSource s{r};
  for (unsigned k = 0; k < n; ++k) out[k] = expsin2_draw(s, sigma);
}
// device-order single draws (site/chain/step select the Philox stream)
double orc_dev_expcos_draw(uint64_t seed, uint32_t chain, uint32_t step, uint32_t site, double beta,
                           double x_p, double x_m) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  DevAngles s{r};
  return s.expcos(site, beta, x_p, x_m);
}
// the tabulated step-envelope sampler: heat-bath draw between x_p and x_m for an action of the given scale (<= 4),
// returned as the device's test hook returns it: wrap_2pi(centre + x)
double orc_dev_vs_draw(uint64_t seed, uint32_t chain, uint32_t step, uint32_t site, double scale, double x_p, double x_m) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  const double c = std::cos(0.5 * (x_m - x_p));
  return wrap_2pi(0.5 * (x_p + x_m) + (c < 0.0 ? kPi : 0.0) + dev_vonmises_table(r, site, scale, x_p, x_m));
}
// the oracle's own construction of the sampler's tables: q[8][8] selector counts, lw[8][8] log2 acceptance factors
void orc_vs_tables(double scale, int *q, float *lw) {
  const VsTables &T = vs_tables(scale);
  for (int c = 0; c < kVsClasses; ++c)
    for (int k = 0; k < kVsBins; ++k) {
      q[c * kVsBins + k] = T.q[c][k];
      lw[c * kVsBins + k] = T.lw[c][k];
    }
}
double orc_dev_expsin2_draw(uint64_t seed, uint32_t chain, uint32_t step, uint32_t site, double sigma) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  DevAngles s{r};
  return s.expsin2(site, sigma);
}

// ---- QoIs --------------------------------------------------------------------------------------
double orc_qoi_xsquared(const double *x, unsigned M) {  // qoi/qm/qoixsquared.cc:7-20
  double s = 0.0;
  for (unsigned i = 0; i < M; ++i) s += x[i] * x[i];
  return s / M;
}
double orc_qoi_susceptibility(const double *x, unsigned M, double T_final) {  // qoi/qm/qoisusceptibility.cc:8-23
  double Q = wrap_2pi(x[0] - x[M - 1]);
  for (unsigned i = 1; i < M; ++i) Q += wrap_2pi(x[i] - x[i - 1]);
  return 1. / (4. * kPi * kPi) * (Q * Q) / T_final;
}
double orc_qoi_2d_susceptibility(const double *x, int Mt, int Mx) {  // qoi/qft/qoi2dsusceptibility.cc:8-27
  ActionO A(SCHWINGER, 1);
  A.g = Grid2{Mt, Mx, false};
  double Q = 0.0;
  for (int i = 0; i < Mt; ++i)
    for (int j = 0; j < Mx; ++j) Q += wrap_2pi(A.plaquette(x, i, j));
  return 1. / (4. * kPi * kPi) * Q * Q;
}
double orc_qoi_avg_plaquette(const double *x, int Mt, int Mx) {  // qoi/qft/qoiavgplaquette.cc:8-27
  ActionO A(SCHWINGER, 1);
  A.g = Grid2{Mt, Mx, false};
  double s = 0.0;
  for (int i = 0; i < Mt; ++i)
    for (int j = 0; j < Mx; ++j) s += std::cos(A.plaquette(x, i, j));
  return s / (Mx * Mt);
}
double orc_qoi_2d_phi_squared(const double *x, unsigned N) {  // qoi/qft/qoi2dphisquared.cc:8-15
  double s = 0.0;
  for (unsigned l = 0; l < N; ++l) s += x[l] * x[l];
  return s / N;
}

// ---- reference-order samplers ---------------------------------------------------------------------
void *orc_hmc_new(void *action, unsigned nt, double dt, unsigned n_rep, unsigned n_burnin, int autotune,
                  unsigned tune_iters, unsigned tune_samples) {
  return new HmcO((ActionO *)action, nt, dt, n_rep, n_burnin, autotune, tune_iters, tune_samples);
}
void orc_hmc_free(void *h) { delete (HmcO *)h; }
int orc_hmc_draw(void *h, double *out) { ((HmcO *)h)->draw(out); return ((HmcO *)h)->accept ? 1 : 0; }
void orc_hmc_set_state(void *h, const double *x) { HmcO *s = (HmcO *)h; std::copy(x, x + s->cur.size(), s->cur.begin()); }
void orc_hmc_get_state(void *h, double *x) { HmcO *s = (HmcO *)h; std::copy(s->cur.begin(), s->cur.end(), x); }
double orc_hmc_dt(void *h) { return ((HmcO *)h)->dt; }
int orc_hmc_tuned(void *h) { return ((HmcO *)h)->tuned; }
double orc_hmc_p_accept(void *h) { return ((HmcO *)h)->p_accept(); }
void orc_hmc_reset_stats(void *h) { ((HmcO *)h)->reset(); }

void *orc_heatbath_new(void *action, unsigned n_sweep_heatbath, unsigned n_sweep_overrelax, unsigned n_burnin,
                       int random_order) {
  return new HeatBathO((ActionO *)action, n_sweep_heatbath, n_sweep_overrelax, n_burnin, random_order != 0);
}
void orc_heatbath_free(void *h) { delete (HeatBathO *)h; }
void orc_heatbath_draw(void *h, double *out) { ((HeatBathO *)h)->draw(out); }
void orc_heatbath_set_state(void *h, const double *x) { HeatBathO *s = (HeatBathO *)h; std::copy(x, x + s->cur.size(), s->cur.begin()); }

// ---- device-order --------------------------------------------------------------------------------
void orc_dev_sweep(void *action, double *x, int heatbath, uint64_t seed, uint32_t chain, uint32_t step) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  dev_sweep(*(ActionO *)action, x, heatbath != 0, r);
}
void orc_dev_site_update(void *action, double *x, unsigned l, int heatbath, uint64_t seed, uint32_t chain, uint32_t step) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  dev_site_update(*(ActionO *)action, x, l, heatbath != 0, r);
}
int orc_dev_hmc_trajectory(void *action, double *x, unsigned nt, double dt, uint64_t seed, uint32_t chain,
                           uint32_t step, double *energies, double *dH) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  return dev_hmc_trajectory(*(ActionO *)action, x, nt, dt, r, energies, dH);
}
// initial state in device order: U(-pi,pi) per entry for rotor / Schwinger (purpose P_INIT),
// N(0,1) for GFF, zeros for HO / quartic.
void orc_dev_initialise(void *action, double *x, uint64_t seed, uint32_t chain) {
  ActionO *A = (ActionO *)action;
  if (A->kind == GFF) {
    dev_gff_exact(*A, x, seed, chain, 0, 1);
    return;
  }
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, 0};
  for (unsigned l = 0; l < A->size(); ++l) {
    if (A->kind == ROTOR || A->kind == SCHWINGER) {
      double u, v;
      r.uniforms(l, P_INIT, 0, u, v);
      x[l] = -kPi + 2.0 * kPi * u;
    } else if (A->kind == GFF) {
      double n0, n1;
      r.normals(l, P_INIT, 0, n0, n1);
      x[l] = n0;
    } else {
      x[l] = 0.0;
    }
  }
}

int orc_dev_twolevel_draw(void *fine, void *coarse, const double *x_coarse, double *theta, uint64_t seed,
                          uint32_t chain, uint32_t step, double *terms) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  return dev_twolevel_draw(*(ActionO *)fine, *(ActionO *)coarse, x_coarse, theta, r, terms);
}

int orc_dev_lattice_twolevel_draw(void *fine, void *coarse, const double *phi_coarse, double *theta, uint64_t seed,
                                  uint32_t chain, uint32_t step, double *terms) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  return dev_schwinger_twolevel_draw(*(ActionO *)fine, *(ActionO *)coarse, phi_coarse, theta, r, terms);
}
int orc_dev_lattice_twolevel_draw_cfa(void *fine, void *coarse, int cfa_kind, const double *phi_coarse, double *theta, uint64_t seed,
                                      uint32_t chain, uint32_t step, double *terms) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  return dev_schwinger_twolevel_draw(*(ActionO *)fine, *(ActionO *)coarse, phi_coarse, theta, r, terms, cfa_kind);
}
// density of GaussianFillinDistribution (for normalisation / consistency checks)
double orc_gaussfill_pdf(double beta, const double *theta4, const double *phi4) {
  return GaussianFillinO(beta).evaluate(theta4[0], theta4[1], theta4[2], theta4[3], phi4[0], phi4[1], phi4[2], phi4[3]);
}
void orc_gaussfill_dev_draw(double beta, const double *phi4, uint64_t seed, uint32_t chain, uint32_t step, uint32_t cell, double *theta4) {
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  GaussianFillinO(beta).dev_draw(r, cell, phi4[0], phi4[1], phi4[2], phi4[3], theta4);
}
double orc_expcos_pdf(double beta, double x, double x_p, double x_m) { return expcos_pdf(beta, x, x_p, x_m); }
double orc_i0_scaled(double z) { return fast_bessel_i0_scaled(z); }

// ---- densities and reference-order draws for the reference-held pins (tests/golden/schwinger_ref_python.json) ------
// distribution/expsin2distribution.cc:7-24 (evaluate; fast_2pi_I0_scaled with its own large-argument branch)
double orc_expsin2_pdf(double x, double sigma) {
  const double z = 0.5 * sigma;
  double norm;
  if (z > 100.) {
    const double zi = 1. / z;
    norm = std::sqrt(2. * kPi * zi) * (1. + 0.125 * zi + 0.0703125 * zi * zi);
  } else {
    norm = 2. * kPi * std::exp(-z) * std::cyl_bessel_i(0.0, z);
  }
  const double s = std::sin(0.5 * x);
  return std::exp(-sigma * s * s) / norm;
}
// distribution/besselproductdistribution.cc:7-12 (evaluate)
double orc_bessel_product_pdf(double beta, double x, double x_p, double x_m) {
  BesselProductO bp(beta);
  return bp.Znorm_inv(x_p - x_m, false) * bessel_i0(2 * beta * std::cos(0.5 * (x - x_p))) *
         bessel_i0(2 * beta * std::cos(0.5 * (x - x_m)));
}
double orc_approx_bessel_pdf(double beta, double x, double x_p, double x_m) { return approx_bessel_pdf(beta, x, x_p, x_m); }
// approximatebesselproductdistribution.cc:43-54; x0 already folded into [0, pi]
void orc_approx_bessel_params(double beta, double x0, double *out3) { approx_bessel_params(beta, x0, out3[0], out3[1], out3[2]); }
// quenchedschwingeraction.cc:13-17: the raw plaquette angle of every vertex, out[Mt * j + i]
void orc_schwinger_plaquettes(void *h, const double *x, double *out) {
  ActionO *A = (ActionO *)h;
  for (int j = 0; j < A->g.Mx; ++j)
    for (int i = 0; i < A->g.Mt; ++i) out[A->g.Mt * j + i] = A->plaquette(x, i, j);
}
// BesselProductDistribution::draw in reference order (besselproductdistribution.hh:88-152): the distribution objects
// live in the distribution (mutable members, hh:155-160), the engine is the caller's.  `preroll` draws are taken first
// with an engine of the same seed, as src/test_distribution.cc does (time_sample draws 10^6 before save_distribution
// reseeds a fresh engine: the normal distribution's cached second variate survives the reseed).
void orc_bessel_product_ref_draws(uint64_t seed, double beta, double x_p, double x_m, unsigned preroll, unsigned n, double *out) {
  BesselProductO bp(beta);
  std::normal_distribution<double> normal(0.0, 1.0);
  std::uniform_real_distribution<double> uniform(0.0, 1.0);
  auto draw = [&](std::mt19937_64 &engine) {
    double dx = x_m - x_p;
    const double flip = (dx < 0) ? -1 : +1;
    dx *= flip;
    const double N_p = std::erf((kPi - 0.5 * dx) / bp.sigma_beta);
    const double N_m = std::erf(0.5 * dx / bp.sigma_beta) * std::pow(bp.I0_twobeta, 2. * (dx / kPi - 1.));
    const double C_p = std::pow(bp.I0_twobeta, 2. * (1. - dx * dx / (4. * kPi * kPi)));
    const double C_m = std::pow(bp.I0_twobeta, 2. * (1. - (dx - 2. * kPi) * (dx - 2. * kPi) / (4. * kPi * kPi)));
    const double sigma = bp.sigma_beta / std::sqrt(2.);
    bool accepted = false;
    double x = 0.0;
    while (!accepted) {
      double xi = uniform(engine);
      double a_min, a_max, mu, C;
      if (xi >= N_m / (N_p + N_m)) {
        a_min = -kPi + dx; a_max = +kPi; mu = 0.5 * dx; C = C_p;
      } else {
        a_min = -kPi; a_max = -kPi + dx; mu = 0.5 * (dx - 2. * kPi); C = C_m;
      }
      while (!accepted) {
        x = sigma * normal(engine) + mu;
        accepted = (x >= a_min) && (x < a_max);
      }
      const double I0 = bessel_i0(2. * beta * std::cos(0.5 * x));
      const double I0_dx = bessel_i0(2. * beta * std::cos(0.5 * (x - dx)));
      const double xs = (x - mu) / bp.sigma_beta;
      xi = uniform(engine);
      accepted = xi <= I0 * I0_dx / C * std::exp(xs * xs);
    }
    return wrap_2pi(flip * x + x_p);
  };
  if (preroll) {
    std::mt19937_64 engine(seed);
    for (unsigned k = 0; k < preroll; ++k) (void)draw(engine);
  }
  std::mt19937_64 engine(seed);
  for (unsigned k = 0; k < n; ++k) out[k] = draw(engine);
}

// ---- exact sampler of the harmonic oscillator (harmonicoscillatoraction.cc:38-66) ---------------------------------
// build_covariance: precision matrix Sigma(i,i) = a m0 mu2 + 2 m0/a, Sigma(i,i+-1) = -m0/a (periodic), L = chol(Sigma^-1).
// The inverse is taken by Gauss-Jordan elimination here (Eigen's .inverse() in the reference), the factor by the
// textbook Cholesky recursion.  L row-major [M][M].
int orc_ho_cholesky(void *action, double *L) {
  const ActionO &A = *(ActionO *)action;
  if (A.kind != HARMONIC) return -1;
  const unsigned M = A.M;
  const double d = A.a * A.m0 * A.mu2 + 2.0 * A.m0 / A.a, c = -A.m0 / A.a;
  std::vector<double> Q((size_t)M * M, 0.0), Cinv((size_t)M * M, 0.0);
  for (unsigned i = 0; i < M; ++i) {
    Q[(size_t)i * M + i] = d;
    Q[(size_t)i * M + (i + 1) % M] += c;
    Q[(size_t)i * M + (i + M - 1) % M] += c;
    Cinv[(size_t)i * M + i] = 1.0;
  }
  for (unsigned col = 0; col < M; ++col) {  // Gauss-Jordan (Q is symmetric positive definite: no pivoting needed)
    const double piv = Q[(size_t)col * M + col];
    for (unsigned k = 0; k < M; ++k) { Q[(size_t)col * M + k] /= piv; Cinv[(size_t)col * M + k] /= piv; }
    for (unsigned r = 0; r < M; ++r) {
      if (r == col) continue;
      const double f = Q[(size_t)r * M + col];
      if (f == 0.0) continue;
      for (unsigned k = 0; k < M; ++k) { Q[(size_t)r * M + k] -= f * Q[(size_t)col * M + k]; Cinv[(size_t)r * M + k] -= f * Cinv[(size_t)col * M + k]; }
    }
  }
  std::fill(L, L + (size_t)M * M, 0.0);
  for (unsigned i = 0; i < M; ++i)
    for (unsigned j = 0; j <= i; ++j) {
      double s = Cinv[(size_t)i * M + j];
      for (unsigned k = 0; k < j; ++k) s -= L[(size_t)i * M + k] * L[(size_t)j * M + k];
      L[(size_t)i * M + j] = (i == j) ? std::sqrt(s) : s / L[(size_t)j * M + j];
    }
  return 0;
}
// draw, device order: y_k from Philox (site k >> 1, P_EXACT, branch k & 1), x = L y
void orc_dev_exact_draw(void *action, const double *L, double *x, uint64_t seed, uint32_t chain, uint32_t step) {
  const ActionO &A = *(ActionO *)action;
  const unsigned M = A.M;
  DevRng r{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  std::vector<double> y(M + 1);
  for (unsigned m = 0; 2 * m < M; ++m) r.normals(m, P_EXACT, 0, y[2 * m], y[2 * m + 1]);
  for (unsigned i = 0; i < M; ++i) {
    double s = 0.0;
    for (unsigned k = 0; k <= i; ++k) s += L[(size_t)i * M + k] * y[k];
    x[i] = s;
  }
}

void orc_dev_gff_exact_draw(void *action, double *phi, uint64_t seed, uint32_t chain, uint32_t step) {
  dev_gff_exact(*(ActionO *)action, phi, seed, chain, step, 0);
}

// ---- transfers between lattice levels ------------------------------------------------------------------
void orc_schwinger_copy_from_fine(int Mt, int Mx, int rt, int rx, const double *fine, double *coarse) {
  schwinger_copy_from_fine(Mt, Mx, rt, rx, fine, coarse);
}
void orc_schwinger_copy_from_coarse(int Mt, int Mx, int rt, int rx, const double *coarse, double *fine) {
  schwinger_copy_from_coarse(Mt, Mx, rt, rx, coarse, fine);
}
// gffaction.cc:97-118 with the fine2coarse_map of lattice2d.cc:126-134 (unrotated): vertex (rt i, rx j) <-> (i, j)
void orc_gff_transfer(int Mt, int Mx, int rt, int rx, double *fine, double *coarse, int to_coarse) {
  Grid2 gc{Mt, Mx, false}, gf{Mt * rt, Mx * rx, false};
  for (int i = 0; i < Mt; ++i)
    for (int j = 0; j < Mx; ++j) {
      if (to_coarse) coarse[gc.vertex(i, j)] = fine[gf.vertex(rt * i, rx * j)];
      else fine[gf.vertex(rt * i, rx * j)] = coarse[gc.vertex(i, j)];
    }
}

// ---- GFF levels -------------------------------------------------------------------------------------------
void *orc_gff_level_new(unsigned Mt, unsigned Mx, int ctype, int level, double mass, int n_gibbs, double omega) {
  return new GffLevelO(Mt, Mx, ctype, level, mass, n_gibbs, omega);
}
void orc_gff_level_free(void *h) { delete (GffLevelO *)h; }
unsigned orc_gff_level_size(void *h) { return ((GffLevelO *)h)->N; }
unsigned orc_gff_level_n_coarse(void *h) { return (unsigned)((GffLevelO *)h)->pairs.size() / 2; }
double orc_gff_level_mu2(void *h) { return ((GffLevelO *)h)->mu2; }
void orc_gff_level_tables(void *h, unsigned *pairs, unsigned *fineonly) {
  GffLevelO *L = (GffLevelO *)h;
  std::copy(L->pairs.begin(), L->pairs.end(), pairs);
  std::copy(L->fineonly.begin(), L->fineonly.end(), fineonly);
}
void orc_gff_level_matrix(void *h, int which, double *out) {
  GffLevelO *L = (GffLevelO *)h;
  L->build();
  const Mat &M = which == 0 ? L->Qhat : L->Lt_inv_T;
  std::copy(M.begin(), M.end(), out);
}
double orc_gff_level_evaluate(void *h, const double *phi) { return ((GffLevelO *)h)->evaluate(phi); }
void orc_gff_level_dev_draw(void *h, double *phi, uint64_t seed, uint32_t chain, uint32_t step) {
  DevRng rng{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  ((GffLevelO *)h)->dev_draw(phi, rng);
}
double orc_gff_cfa_evaluate(void *h, const double *phi) { return ((GffLevelO *)h)->cfa(const_cast<double *>(phi), false, nullptr); }
double orc_gff_cfa_dev_fill(void *h, double *phi, uint64_t seed, uint32_t chain, uint32_t step) {
  DevRng rng{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  return ((GffLevelO *)h)->cfa(phi, true, &rng);
}
void orc_gff_copy(void *fine_h, double *fine, double *coarse, int to_coarse) {  // gffaction.cc:97-118
  GffLevelO *F = (GffLevelO *)fine_h;
  for (size_t k = 0; k < F->pairs.size(); k += 2) {
    if (to_coarse) coarse[F->pairs[k + 1]] = fine[F->pairs[k]]; else fine[F->pairs[k]] = coarse[F->pairs[k + 1]];
  }
}
// twolevelmetropolisstep.cc:35-89 in device order; theta updated in place on acceptance; terms = (dS_fine, dS_coarse, dS_trial)
int orc_gff_dev_twolevel_draw(void *fine_h, void *coarse_h, const double *phi_coarse, double *theta, uint64_t seed, uint32_t chain,
                              uint32_t step, double *terms) {
  GffLevelO *F = (GffLevelO *)fine_h, *Cc = (GffLevelO *)coarse_h;
  DevRng rng{(uint32_t)seed, (uint32_t)(seed >> 32), chain, step};
  std::vector<double> prime(F->N, 0.0), theta_C(Cc->N, 0.0);
  orc_gff_copy(F, prime.data(), const_cast<double *>(phi_coarse), 0);
  const double cfa_prime = F->cfa(prime.data(), true, &rng);
  const double cfa_theta = F->cfa(theta, false, nullptr);
  const double dS_fine = F->evaluate(prime.data()) - F->evaluate(theta);
  orc_gff_copy(F, theta, theta_C.data(), 1);
  const double dS_coarse = Cc->evaluate(theta_C.data()) - Cc->evaluate(phi_coarse);
  const double dS_trial = cfa_theta - cfa_prime;
  if (terms) { terms[0] = dS_fine; terms[1] = dS_coarse; terms[2] = dS_trial; }
  const double dS = dS_fine + dS_coarse + dS_trial;
  bool accept = dS < 0.0;
  if (!accept) {
    double u, v;
    rng.uniforms(0, P_ACCEPT2, 0, u, v);
    accept = u < std::exp(-dS);
  }
  if (accept) std::copy(prime.begin(), prime.end(), theta);
  return accept ? 1 : 0;
}

// ---- statistics ------------------------------------------------------------------------------------
void *orc_stats_new(unsigned k_max) { return new StatsO(k_max); }
void orc_stats_free(void *h) { delete (StatsO *)h; }
void orc_stats_record(void *h, const double *q, unsigned n) { for (unsigned i = 0; i < n; ++i) ((StatsO *)h)->record(q[i]); }
void orc_stats_reset(void *h, int hard) { if (hard) ((StatsO *)h)->hard_reset(); else ((StatsO *)h)->reset(); }
// out: average, variance, variance_error, tau_int, error, samples
void orc_stats_get(void *h, double *out) {
  StatsO *s = (StatsO *)h;
  out[0] = s->avg; out[1] = s->variance(); out[2] = s->variance_error(); out[3] = s->tau_int();
  out[4] = s->error(); out[5] = (double)s->n;
}

// mpi/mpi_random.cc:5-29: seeds of the ranks' engines = the sorted set {seed} + outputs of minstd_rand(seed) until it
// holds `world` distinct values (std::set iterates in ascending order; rank r takes the r-th smallest).
void orc_rank_seeds(unsigned seed, unsigned world, unsigned *out) {
  std::linear_congruential_engine<unsigned int, 48271, 0, 2147483647> seed_engine;
  seed_engine.seed(seed);
  std::set<unsigned int> seeds;
  seeds.insert(seed);
  while (seeds.size() < world) seeds.insert(seed_engine());
  unsigned r = 0;
  for (unsigned v : seeds) out[r++] = v;
}
// the engine of rank `rank`: std::mt19937_64 seeded with that rank's seed (parallel_mt19937_64::seed)
void orc_rank_engine_outputs(unsigned seed, unsigned rank, unsigned world, unsigned n, unsigned long long *out) {
  std::vector<unsigned> seeds(world);
  orc_rank_seeds(seed, world, seeds.data());
  std::mt19937_64 engine(seeds[rank]);
  for (unsigned i = 0; i < n; ++i) out[i] = engine();
}

// ---- analytic expectation values ---------------------------------------------------------------------
// action/qm/harmonicoscillatoraction.cc:69-74
double orc_ho_xsquared_analytical(unsigned M, double T_final, double m0, double mu2) {
  double a = T_final / M;
  double R = 1. + 0.5 * a * a * mu2 - a * std::sqrt(mu2) * std::sqrt(1. + 0.25 * a * a * mu2);
  return 1. / (2. * m0 * std::sqrt(mu2) * std::sqrt(1 + 0.25 * a * a * mu2)) * (1. + std::pow(R, (double)M)) /
         (1. - std::pow(R, (double)M));
}
// common/auxilliary.cc:197-209
double orc_gff_phi_squared_analytical(double mass, int Mt, int Mx) {
  double mu2 = mass * mass / (1.0 * Mt * Mx), s = 0.0;
  for (int k1 = 0; k1 < Mt; ++k1)
    for (int k2 = 0; k2 < Mx; ++k2) {
      double s1 = std::sin(kPi * k1 / Mt), s2 = std::sin(kPi * k2 / Mx);
      s += 1. / (4. * (s1 * s1 + s2 * s2) + mu2);
    }
  return s / (1.0 * Mt * Mx);
}

}  // extern "C"
