// runtime.hip -- plumbing entry points of the C ABI: errors, device selection, memory, host-side
// index maps (lattice/lattice2d.hh:230-268,348-375; lattice/lattice1d.cc:11-18;
// lattice/lattice2d.cc:137-155), RNG test hooks.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "internal.hpp"

namespace mlmcpi {

static thread_local char g_err[512] = "";

int fail(int status, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

int fail_hip(hipError_t e, const char *what) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return MLMCPI_ERR_HIP;
}

// Scratch for reduction partials: one buffer per (host thread, device, stream), grown on demand (the first call of a
// given size allocates; steady state does not).  A launch sequence that uses it (reduce -> finish) is issued by one
// thread on one stream, so neither another stream nor another host thread driving the same device can overwrite the
// partials in between.  The caller passes the stream it is about to launch on.
namespace {
struct ScratchSlot { int dev; hipStream_t stream; void *ptr; size_t bytes; };
struct ScratchPool {
  std::vector<ScratchSlot> slots;
  ~ScratchPool() {
    for (auto &s : slots)
      if (s.ptr) (void)hipFree(s.ptr);  // may fail at process teardown (runtime already gone): nothing to do about it
  }
};
thread_local ScratchPool t_scratch;
}  // namespace

int scratch(size_t bytes, void **d_ptr, hipStream_t stream) {
  int dev = 0;
  MLMCPI_HIP_TRY(hipGetDevice(&dev));
  ScratchSlot *slot = nullptr;
  for (auto &s : t_scratch.slots)
    if (s.dev == dev && s.stream == stream) slot = &s;
  if (!slot) {
    t_scratch.slots.push_back(ScratchSlot{dev, stream, nullptr, 0});
    slot = &t_scratch.slots.back();
  }
  if (slot->bytes < bytes) {
    if (slot->ptr) {
      MLMCPI_HIP_TRY(hipStreamSynchronize(stream));  // the only work that can still read it is on this stream
      MLMCPI_HIP_TRY(hipFree(slot->ptr));
      slot->ptr = nullptr;
      slot->bytes = 0;
    }
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    MLMCPI_HIP_TRY(hipMalloc(&slot->ptr, want));
    slot->bytes = want;
  }
  *d_ptr = slot->ptr;
  return MLMCPI_OK;
}

// ---- tuning knobs -----------------------------------------------------------------------------------
namespace {
std::mutex g_tuning_mutex;
Tuning g_tuning;
bool g_tuning_loaded = false;

bool apply_option(Tuning &t, const char *name, const char *value) {
  const std::string n = name ? name : "", v = value ? value : "";
  if (n == "MLMCPI_SWEEP_TILE") {
    unsigned a = 0, b = 0, c = 0;
    t.tile_w = t.tile_h = t.tile_nt = 0;
    if (v.empty()) return true;
    if (sscanf(v.c_str(), "%ux%ux%u", &a, &b, &c) == 3 && a >= 2 && b >= 2 && a % 2 == 0 && b % 2 == 0 && (c == 256 || c == 512 || c == 1024)) {
      t.tile_w = a; t.tile_h = b; t.tile_nt = c;
      return true;
    }
    return false;
  }
  if (n == "MLMCPI_OR_KERNEL") {
    t.or_lds = v == "lds";
    t.or_patch = v == "patch";
    t.or_block = v == "block";
    return v.empty() || v == "lds" || v == "patch" || v == "block" || v == "perm";
  }
  if (n == "MLMCPI_OR_HEAT") {
    t.or_heat_split = v == "split";
    t.or_heat_wide = v == "wide" ? 1 : v == "narrow" ? -1 : 0;
    return v.empty() || v == "split" || v == "fused" || v == "wide" || v == "narrow";
  }
  if (n == "MLMCPI_OR_THREADS") {
    const unsigned x = (unsigned)atoi(v.c_str());
    t.or_threads = (x == 256 || x == 512 || x == 1024) ? x : 0;
    return v.empty() || t.or_threads != 0;
  }
  return false;
}
void load_tuning_locked() {
  if (g_tuning_loaded) return;
  for (const char *name : {"MLMCPI_SWEEP_TILE", "MLMCPI_OR_KERNEL", "MLMCPI_OR_THREADS", "MLMCPI_OR_HEAT"})
    if (const char *e = getenv(name)) apply_option(g_tuning, name, e);
  g_tuning_loaded = true;
}
}  // namespace

Tuning tuning() {
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  load_tuning_locked();
  return g_tuning;
}

// ---- tables of the step-envelope sampler (device_common.hpp, "tabulated step envelope") -----------------------------------
// For the action's scale (kappa = scale |cos(.)| <= scale <= kVsKappaMax) and each of the kVsClasses ranges of kappa: the
// proposal probabilities q_k / 64 of the eight bins, as a 64-entry selector, and log2 of the acceptance factors.
// q[c][k] selector values for bin k of class c, lw[c][k] = log2 of its acceptance factor
void vs_build_tables(double scale, int *q_out, float *lw) {
  for (int c = 0; c < kVsClasses; ++c) {
    const double kmin = scale * sin(2.0 * kPi * c / 32.0);  // the smallest concentration of the class
    double hw[kVsBins], w[kVsBins], total = 0.0;
    for (int k = 0; k < kVsBins; ++k) {
      w[k] = vs_width16(k) * (kPi / 16.0);
      hw[k] = exp(kmin * (cos(vs_edge16(k) * (kPi / 16.0)) - 1.0)) * w[k];  // target at the bin's left edge (its maximum) x width
      total += hw[k];
    }
    int *q = q_out + c * kVsBins, sum = 0;
    for (int k = 0; k < kVsBins; ++k) {
      q[k] = (int)floor(hw[k] / total * kVsSel);
      if (q[k] < 1) q[k] = 1;
      sum += q[k];
    }
    while (sum < kVsSel) {  // the bin that binds the envelope constant gets the next selector value
      int j = 0;
      for (int k = 1; k < kVsBins; ++k)
        if (hw[k] / q[k] > hw[j] / q[j]) j = k;
      ++q[j];
      ++sum;
    }
    while (sum > kVsSel) {  // (many bins at the minimum of one value): take from the bin that stays lowest
      int j = -1;
      for (int k = 0; k < kVsBins; ++k)
        if (q[k] > 1 && (j < 0 || hw[k] / (q[k] - 1) < hw[j] / (q[j] - 1))) j = k;
      --q[j];
      --sum;
    }
    double M = 0.0;
    for (int k = 0; k < kVsBins; ++k) M = fmax(M, hw[k] / q[k]);
    for (int k = 0; k < kVsBins; ++k) {
      // acceptance factor (w_k / q_k) / M <= 1 / H_k, lowered by 4e-6 in log2 so that float rounding never lifts it above
      const double x = log2(w[k] / q[k] / M) - 4e-6;
      float f = (float)x;
      if ((double)f > x) f = nextafterf(f, -INFINITY);
      lw[c * kVsBins + k] = f;
    }
  }
}

// the device image (device_common.hpp, "Device image of the tables"): code | scr | lw | fin | consts
static void vs_device_image(double scale, uint32_t *image) {
  int q[kVsClasses * kVsBins];
  float lw[kVsClasses * kVsBins];
  vs_build_tables(scale, q, lw);
  memset(image, 0, kVsTableBytes);
  uint8_t *bytes = (uint8_t *)image;
  for (int c = 0; c < kVsClasses; ++c) {
    int pos = 0;
    for (int k = 0; k < kVsBins; ++k) {
      for (int i = 0; i < q[c * kVsBins + k]; ++i) bytes[kVsCodeOff + c * kVsSel + pos++] = (uint8_t)(8 * k);
      memcpy(bytes + kVsLwOff + c * kVsSel + 8 * k, &lw[c * kVsBins + k], 4);
    }
  }
  for (int k = 0; k < kVsBins; ++k) {
    const double e = vs_edge16(k), w = vs_width16(k);
    const float scr[2] = {(float)(0.5 * kPi - (kPi / 16.0) * e), (float)(-(kPi / 16.0) * w / 4294967296.0)};
    const double fin[2] = {64.0 * w, e - 64.0 * w};
    memcpy(bytes + kVsScrOff + 8 * k, scr, 8);
    memcpy(bytes + kVsFinOff + 16 * k, fin, 16);
  }
  const float band = vs_band_of_scale(scale);
  const float consts[4] = {4294967296.0f * (1.0f - band), 4294967296.0f * (1.0f + band), 0.f, 0.f};
  memcpy(bytes + kVsConstOff, consts, 16);
}

namespace {
struct VsTableEntry { int device; double scale; uint32_t *d_table; };
std::mutex g_vs_mutex;
std::vector<VsTableEntry> g_vs_tables;
}  // namespace

int vs_table_device(double scale, const uint32_t **d_table) {
  int dev = 0;
  MLMCPI_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_vs_mutex);
  for (const VsTableEntry &e : g_vs_tables)
    if (e.device == dev && e.scale == scale) {
      *d_table = e.d_table;
      return MLMCPI_OK;
    }
  uint32_t host[kVsTableBytes / 4];
  vs_device_image(scale, host);
  uint32_t *d = nullptr;
  MLMCPI_HIP_TRY(hipMalloc((void **)&d, kVsTableBytes));
  MLMCPI_HIP_TRY(hipMemcpy(d, host, kVsTableBytes, hipMemcpyHostToDevice));  // first use of this scale on this device only
  g_vs_tables.push_back(VsTableEntry{dev, scale, d});
  *d_table = d;
  return MLMCPI_OK;
}

// ---- analytic topological susceptibility of the quenched Schwinger model and the coupling matched to it (host) -----------
// common/auxilliary.cc:30-33,44-79 (Phi_chit), :98-193 (compute_In) and action/qft/quenchedschwingerrenormalisation.cc:7-64.
// The reference integrates with GSL (QAWO / QAG) and finds the root with gsl_root_fsolver_bisection; here: composite
// 16-point Gauss-Legendre panels over [-pi, pi] (the integrands are analytic; panels narrow enough for the peak of width
// 1/sqrt(x) at the origin) and a plain bisection with the reference's interval, tolerance and fall-back.
namespace {
// nodes / weights of the 16-point Gauss-Legendre rule on [-1, 1] (positive half)
const double kGLx[8] = {0.0950125098376374401853193, 0.2816035507792589132304605, 0.4580167776572273863424194, 0.6178762444026437484466718,
                        0.7554044083550030338951012, 0.8656312023878317438804679, 0.9445750230732325760779884, 0.9894009349916499325961542};
const double kGLw[8] = {0.1894506104550684962853967, 0.1826034150449235888667637, 0.1691565193950025381893121, 0.1495959888165767320815017,
                        0.1246289712555338720524763, 0.0951585116824927848099251, 0.0622535239386478928628438, 0.0271524594117540948517806};

template <class F>
double integrate_mpi_pi(F f, double x) {
  int panels = 64;
  const int want = (int)ceil(8.0 * sqrt(fabs(x) + 1.0));
  if (want > panels) panels = want;
  const double h = 2.0 * kPi / panels;
  double sum = 0.0;
  for (int p = 0; p < panels; ++p) {
    const double mid = -kPi + (p + 0.5) * h;
    for (int i = 0; i < 8; ++i) sum += kGLw[i] * (f(mid + 0.5 * h * kGLx[i]) + f(mid - 0.5 * h * kGLx[i]));
  }
  return 0.5 * h * sum;
}

// exp(-x) I_n(x), n < nmax (the reference: gsl_sf_bessel_In_scaled, auxilliary.cc:159): Miller's downward recurrence
// f_{k-1} = f_{k+1} + (2k / x) f_k from far above the orders wanted, normalised by exp(x) = I_0 + 2 sum_k I_k.  Every term is
// positive, so the relative accuracy is that of the recurrence (~1e-15) also where I_n is 1e-23 of I_0 -- a quadrature
// with absolute error 1e-16 returns noise of either sign there.
void bessel_In_scaled(double x, int nmax, double *In) {
  const int N = nmax + 64 + (int)(12.0 * sqrt(fabs(x)));
  double f_up = 0.0, f = 1e-280, sum = 0.0;
  for (int n = 0; n < nmax; ++n) In[n] = 0.0;
  for (int k = N; k >= 1; --k) {  // f = f_k, f_up = f_{k+1}
    const double f_dn = f_up + (2.0 * k / x) * f;
    sum += 2.0 * f;
    if (k < nmax) In[k] = f;
    f_up = f;
    f = f_dn;
    if (f > 1e250) {  // rescale everything computed so far
      const double s = 1e-250;
      f *= s; f_up *= s; sum *= s;
      for (int n = 0; n < nmax; ++n) In[n] *= s;
    }
  }
  In[0] = f;
  sum += f;
  for (int n = 0; n < nmax; ++n) In[n] /= sum;
}

// In = exp(-x) I_n(x), and the two integrals the reference calls I'_n and I''_n (auxilliary.cc:98-131)
void schwinger_In(double x, int nmax, double *In, double *dIn, double *ddIn) {
  bessel_In_scaled(x, nmax, In);
  for (int n = 0; n < nmax; ++n) {
    dIn[n] = integrate_mpi_pi([&](double phi) { return -1. / (4. * kPi * kPi) * phi * exp(x * (cos(phi) - 1.0)) * sin(n * phi); }, x);
    ddIn[n] = integrate_mpi_pi([&](double phi) { return 1. / (8. * kPi * kPi * kPi) * phi * phi * exp(x * (cos(phi) - 1.0)) * cos(n * phi); }, x);
  }
}

// auxilliary.cc:44-79
double phi_chit(double beta, unsigned int n_plaq) {
  const int nmax = 20;
  double In[nmax], dIn[nmax], ddIn[nmax], weight[nmax], weight_sum = 0.0;
  schwinger_In(beta, nmax, In, dIn, ddIn);
  for (int n = 0; n < nmax; ++n) {
    weight[n] = (1 + (n > 0)) * pow(In[n] / In[0], (double)n_plaq);
    weight_sum += weight[n];
  }
  double r = 0.0;
  for (int n = 0; n < nmax; ++n)
    if (weight[n] > 0.0)  // (terms whose weight underflowed: In[n] may be 0 too)
      r += beta * weight[n] / weight_sum * (ddIn[n] / In[n] - (n_plaq - 1.0) * (dIn[n] * dIn[n]) / (In[n] * In[n]));
  return r;
}
double schwinger_chit(double beta, unsigned int n_plaq) { return n_plaq / beta * phi_chit(beta, n_plaq); }  // auxilliary.cc:30-33
}  // namespace

// ---- test kernels ------------------------------------------------------------------------------
__global__ void test_random_kernel(RngKey key, uint32_t purpose, uint32_t sub, uint32_t n, double *out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double u0, u1, n0, n1;
  rng_uniforms(key, k, purpose, sub, u0, u1);
  rng_normals(key, k, purpose, sub, n0, n1);
  out[4 * k + 0] = u0;
  out[4 * k + 1] = u1;
  out[4 * k + 2] = n0;
  out[4 * k + 3] = n1;
}

__global__ void test_philox_kernel(U4 ctr, uint32_t k0, uint32_t k1, uint32_t *out) {
  U4 r = philox4x32_10(ctr.x, ctr.y, ctr.z, ctr.w, k0, k1);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

__global__ void test_expcos_kernel(RngKey key, double beta, const double *xp, const double *xm, uint32_t n,
                                   double *out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = expcos_draw(key, k, beta, xp[k], xm[k]);
}

__global__ void __launch_bounds__(256)
    test_vs_draw_kernel(RngKey key, double scale, const double *xp, const double *xm, uint32_t n, const uint32_t *d_table, double *out) {
  __shared__ uint32_t tab_lds[kVsTableBytes / 4];
  const VsTable tab = VsTable::stage(tab_lds, d_table);
  __syncthreads();
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = vs_draw(key, k, scale, xp[k], xm[k], tab);
}

__global__ void test_expsin2_kernel(RngKey key, const double *sigma, uint32_t n, double *out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = expsin2_draw(key, k, sigma[k]);
}

}  // namespace mlmcpi

using namespace mlmcpi;

extern "C" {

int mlmcpi_abi_version(void) { return MLMCPI_ABI_VERSION; }
const char *mlmcpi_last_error(void) { return g_err; }

int mlmcpi_device_count(int *count) {
  MLMCPI_REQUIRE(count, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    fail_hip(e, "hipGetDeviceCount");
    return MLMCPI_ERR_NO_DEVICE;
  }
  return MLMCPI_OK;
}

int mlmcpi_set_device(int device) {
  MLMCPI_HIP_TRY(hipSetDevice(device));
  return MLMCPI_OK;
}

int mlmcpi_device_name(char *buf, size_t len) {
  MLMCPI_REQUIRE(buf && len > 0, "buf is NULL");
  int dev = 0;
  MLMCPI_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MLMCPI_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, len, "%s (%s)", prop.name, prop.gcnArchName);
  return MLMCPI_OK;
}

int mlmcpi_malloc(void **d_ptr, size_t bytes) {
  MLMCPI_REQUIRE(d_ptr, "d_ptr is NULL");
  MLMCPI_HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
  return MLMCPI_OK;
}
int mlmcpi_free(void *d_ptr) {
  if (d_ptr) MLMCPI_HIP_TRY(hipFree(d_ptr));
  return MLMCPI_OK;
}
int mlmcpi_memset(void *d_ptr, int value, size_t bytes, void *stream) {
  MLMCPI_HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, as_stream(stream)));
  return MLMCPI_OK;
}
int mlmcpi_copy_h2d(void *d_dst, const void *src, size_t bytes, void *stream) {
  MLMCPI_HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  MLMCPI_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return MLMCPI_OK;
}
int mlmcpi_copy_d2h(void *dst, const void *d_src, size_t bytes, void *stream) {
  MLMCPI_HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  MLMCPI_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return MLMCPI_OK;
}
int mlmcpi_copy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream) {
  MLMCPI_HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return MLMCPI_OK;
}
int mlmcpi_stream_synchronize(void *stream) {
  MLMCPI_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return MLMCPI_OK;
}

// ---- index maps ----------------------------------------------------------------------------------
uint32_t mlmcpi_vertex_cart2lin(uint32_t Mt, uint32_t Mx, int rotated, int i, int j) {
  const int mt = (int)Mt, mx = (int)Mx;
  if (rotated) {
    const int half_t = mt / 2, half_x = mx / 2;
    const int ii = ((i + mt) - (i & 1)) / 2, jj = ((j + mx) - (j & 1)) / 2;
    return (uint32_t)(half_t * (jj % half_x) + ii % half_t + (mt * mx / 4) * (i & 1));
  }
  return (uint32_t)(mt * ((j + mx) % mx) + (i + mt) % mt);
}

void mlmcpi_vertex_lin2cart(uint32_t Mt, uint32_t Mx, int rotated, uint32_t ell, int *i, int *j) {
  const int mt = (int)Mt, mx = (int)Mx;
  if (rotated) {
    const int quarter = mt * mx / 4, half_t = mt / 2;
    const int parity = (int)ell / quarter;
    const int rest = (int)ell - quarter * parity;
    const int jh = rest / half_t;
    *j = 2 * jh + parity;
    *i = 2 * (rest - half_t * jh) + parity;
  } else {
    *j = (int)(ell / Mt);
    *i = (int)(ell - Mt * (uint32_t)*j);
  }
}

uint32_t mlmcpi_link_cart2lin(uint32_t Mt, uint32_t Mx, int i, int j, int mu) {
  const int mt = (int)Mt, mx = (int)Mx;
  return (uint32_t)(2 * mt * ((j + mx) % mx) + 2 * ((i + mt) % mt) + mu);
}

void mlmcpi_link_lin2cart(uint32_t Mt, uint32_t Mx, uint32_t ell, int *i, int *j, int *mu) {
  (void)Mx;
  const uint32_t row = ell / (2 * Mt), rest = ell - 2 * Mt * row;
  *j = (int)row;
  *i = (int)(rest >> 1);
  *mu = (int)(rest & 1u);
}

int mlmcpi_neighbours_1d(uint32_t M, uint32_t *out) {
  MLMCPI_REQUIRE(out && M > 0, "bad arguments");
  for (uint32_t ell = 0; ell < M; ++ell) {
    out[2 * ell] = (ell + M - 1) % M;
    out[2 * ell + 1] = (ell + 1) % M;
  }
  return MLMCPI_OK;
}

int mlmcpi_neighbours_2d(uint32_t Mt, uint32_t Mx, int rotated, uint32_t *out) {
  MLMCPI_REQUIRE(out && Mt > 0 && Mx > 0, "bad arguments");
  MLMCPI_REQUIRE(!rotated || (Mt % 2 == 0 && Mx % 2 == 0), "rotated lattices need even Mt, Mx");
  // nearest neighbours first, then diagonals (next-nearest on the rotated lattice)
  static const int step_plain[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
  static const int step_rot[8][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}, {2, 0}, {-2, 0}, {0, 2}, {0, -2}};
  const uint32_t nv = rotated ? Mt * Mx / 2 : Mt * Mx;
  for (uint32_t ell = 0; ell < nv; ++ell) {
    int i, j;
    mlmcpi_vertex_lin2cart(Mt, Mx, rotated, ell, &i, &j);
    for (int k = 0; k < 8; ++k) {
      const int *s = rotated ? step_rot[k] : step_plain[k];
      out[8 * ell + k] = mlmcpi_vertex_cart2lin(Mt, Mx, rotated, i + s[0], j + s[1]);
    }
  }
  return MLMCPI_OK;
}

int mlmcpi_set_option(const char *name, const char *value) {
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  load_tuning_locked();
  if (!apply_option(g_tuning, name, value)) return fail(MLMCPI_ERR_INVALID, "unknown option or value: %s=%s", name ? name : "(null)", value ? value : "(null)");
  return MLMCPI_OK;
}

// ---- test hooks ------------------------------------------------------------------------------------
int mlmcpi_test_philox(const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4) {
  MLMCPI_REQUIRE(ctr4 && key2 && out4, "NULL argument");
  uint32_t *d = nullptr;
  MLMCPI_HIP_TRY(hipMalloc(&d, 16));
  hipLaunchKernelGGL(test_philox_kernel, dim3(1), dim3(1), 0, 0, U4{ctr4[0], ctr4[1], ctr4[2], ctr4[3]}, key2[0],
                     key2[1], d);
  MLMCPI_LAUNCH_CHECK("test_philox_kernel");
  MLMCPI_HIP_TRY(hipMemcpy(out4, d, 16, hipMemcpyDeviceToHost));
  MLMCPI_HIP_TRY(hipFree(d));
  return MLMCPI_OK;
}

int mlmcpi_test_random(uint64_t seed, uint32_t chain, uint32_t step, uint32_t purpose, uint32_t sub, uint32_t n,
                       double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_out && n > 0, "bad arguments");
  hipLaunchKernelGGL(test_random_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                     make_key(seed, chain, step), purpose, sub, n, d_out);
  MLMCPI_LAUNCH_CHECK("test_random_kernel");
  return MLMCPI_OK;
}

int mlmcpi_test_expcos(uint64_t seed, uint32_t chain, uint32_t step, double beta, const double *d_xp,
                       const double *d_xm, uint32_t n, double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_xp && d_xm && d_out && n > 0, "bad arguments");
  hipLaunchKernelGGL(test_expcos_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                     make_key(seed, chain, step), beta, d_xp, d_xm, n, d_out);
  MLMCPI_LAUNCH_CHECK("test_expcos_kernel");
  return MLMCPI_OK;
}

int mlmcpi_test_vs_draw(uint64_t seed, uint32_t chain, uint32_t step, double scale, const double *d_xp, const double *d_xm,
                        uint32_t n, double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_xp && d_xm && d_out && n > 0 && scale >= 0.0 && scale <= kVsKappaMax, "bad arguments");
  const uint32_t *d_table = nullptr;
  if (int rc = vs_table_device(scale, &d_table)) return rc;
  hipLaunchKernelGGL(test_vs_draw_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), make_key(seed, chain, step),
                     scale, d_xp, d_xm, n, d_table, d_out);
  MLMCPI_LAUNCH_CHECK("test_vs_draw_kernel");
  return MLMCPI_OK;
}

int mlmcpi_schwinger_chit_analytical(double beta, uint32_t n_plaq, double *chit) {
  MLMCPI_REQUIRE(chit && beta > 0.0 && beta <= 2000.0 && n_plaq > 0, "bad arguments (0 < beta <= 2000: the series is unstable beyond, auxilliary.cc:46-52)");
  *chit = schwinger_chit(beta, n_plaq);
  return MLMCPI_OK;
}

int mlmcpi_schwinger_beta_coarse_nonperturbative(double beta, uint32_t n_plaq, int32_t rho_refine, double *beta_coarse) {
  // the root is bracketed on x in [0.01, 2], so Phi_chit is evaluated at 2 beta: the reference aborts there for
  // 2 beta > 2000 (auxilliary.cc:46-52, the series is unstable beyond), and so does this
  MLMCPI_REQUIRE(beta_coarse && beta > 0.0 && 2.0 * beta <= 2000.0 && (rho_refine == 2 || rho_refine == 4) && n_plaq >= (uint32_t)rho_refine,
                 "bad arguments (0 < beta <= 1000: the bracket [0.01, 2] beta must stay below 2000; rho_refine 2 or 4; n_plaq >= rho_refine)");
  const double target = schwinger_chit(beta, n_plaq);
  auto f = [&](double x) { return schwinger_chit(x * beta, n_plaq / rho_refine) - target; };  // quenchedschwingerrenormalisation.hh:125-133
  double x_lo = 0.01, x_hi = 2.0, f_lo = f(x_lo), f_hi = f(x_hi), x;
  if ((f_lo > 0 && f_hi > 0) || (f_lo < 0 && f_hi < 0)) {
    x = rho_refine == 4 ? 0.25 : 0.5;  // no root in the interval: the reference's fall-back
  } else {
    for (int k = 0; k < 100; ++k) {  // bisection, relative tolerance 1e-12 on the interval (gsl_root_test_interval)
      x = 0.5 * (x_lo + x_hi);
      const double f_mid = f(x);
      if ((f_mid > 0) == (f_lo > 0)) { x_lo = x; f_lo = f_mid; } else { x_hi = x; }
      if (fabs(x_hi - x_lo) < 1e-12 * fmin(fabs(x_lo), fabs(x_hi))) break;
    }
    x = 0.5 * (x_lo + x_hi);
  }
  *beta_coarse = x * beta;
  return MLMCPI_OK;
}

int mlmcpi_vs_table(double scale, uint8_t *sel, float *lw) {
  MLMCPI_REQUIRE(sel && lw && scale >= 0.0 && scale <= kVsKappaMax, "bad arguments (0 <= scale <= %g)", kVsKappaMax);
  int q[kVsClasses * kVsBins];
  vs_build_tables(scale, q, lw);
  for (int c = 0; c < kVsClasses; ++c) {
    int pos = 0;
    for (int k = 0; k < kVsBins; ++k)
      for (int i = 0; i < q[c * kVsBins + k]; ++i) sel[c * kVsSel + pos++] = (uint8_t)k;
  }
  return MLMCPI_OK;
}

int mlmcpi_test_expsin2(uint64_t seed, uint32_t chain, uint32_t step, const double *d_sigma, uint32_t n,
                        double *d_out, void *stream) {
  MLMCPI_REQUIRE(d_sigma && d_out && n > 0, "bad arguments");
  hipLaunchKernelGGL(test_expsin2_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                     make_key(seed, chain, step), d_sigma, n, d_out);
  MLMCPI_LAUNCH_CHECK("test_expsin2_kernel");
  return MLMCPI_OK;
}

}  // extern "C"
