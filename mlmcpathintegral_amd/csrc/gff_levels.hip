// gff_levels.hip -- the multilevel glue of the Gaussian free field:
//   GFFAction on any level of a CoarsenRotate / CoarsenBoth hierarchy      action/qft/gffaction.{hh,cc}
//     evaluate (stencil form, or phi^T Qhat phi of the Gibbs-smoothed coarse actions), draw (exact draw + Gibbs
//     smoothing sweeps = the coarse-level sampler, GFFSamplerFactory), copy_from_coarse / copy_from_fine
//   GFFConditionedFineAction::fill_fine_points / evaluate                   action/qft/gffconditionedfineaction.cc:7-49
//   TwoLevelMetropolisStep::draw for a pair of GFF levels                   montecarlo/twolevelmetropolisstep.cc:35-89
//
// The reference builds the coarse actions with dense Eigen algebra (gffaction.cc:126-173): Qhat = (Sigma_eff +
// G (Sigma - Sigma_eff) G^T)^-1 with Sigma = Q^-1 (plain stencil of the level), Sigma_eff = Q_eff^-1 (the exact
// marginal of the finer level: 9-point stencil) and G = (1 - M^-1 Q_eff)^n the iteration matrix of n lexicographic
// Gibbs (SOR) sweeps.  G is not translation invariant, so Qhat is a genuinely dense matrix; it is built here the same
// way, on the host, with Cholesky-based inverses (every matrix inverted is symmetric positive definite), for levels of
// up to kMaxDense vertices -- the reference's own practical limit -- and applied on the device as a dense
// quadratic form.  Index maps come from tables (neighbours, fine-only vertices, fine -> coarse pairs) built with the
// host index functions that are pinned to the reference's Lattice2D (rotated levels included), exactly as the reference
// drives its loops from tables.
#include <algorithm>
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace mlmcpi {

constexpr uint32_t kMaxDense = 4096;  // vertices of a level that gets dense matrices (128 MiB per matrix at the cap)
constexpr uint32_t P_GFF_GIBBS = 11;  // Gibbs smoothing sweeps of GFFAction::draw: pair l >> 1, branch l & 1, sub = sweep
constexpr uint32_t P_GFF_EXACT = 12;  // white noise of the exact draw on a dense level: pair l >> 1, branch l & 1

// ---- dense symmetric algebra on the host (row major, n x n) ---------------------------------------------------------
// in place: lower Cholesky factor of a symmetric positive definite matrix (upper part left untouched); false if it fails
static bool cholesky(std::vector<double> &A, uint32_t n) {
  for (uint32_t j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (uint32_t k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (uint32_t i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      const double *ri = &A[(size_t)i * n], *rj = &A[(size_t)j * n];
      for (uint32_t k = 0; k < j; ++k) s -= ri[k] * rj[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}

// inverse of a lower-triangular matrix (lower part of L; result lower triangular, zeros above)
static std::vector<double> lower_inverse(const std::vector<double> &L, uint32_t n) {
  std::vector<double> X((size_t)n * n, 0.0);
  for (uint32_t c = 0; c < n; ++c) {  // column c of the inverse: forward substitution on e_c
    X[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
    for (uint32_t i = c + 1; i < n; ++i) {
      double s = 0.0;
      for (uint32_t k = c; k < i; ++k) s -= L[(size_t)i * n + k] * X[(size_t)k * n + c];
      X[(size_t)i * n + c] = s / L[(size_t)i * n + i];
    }
  }
  return X;
}

// C = A B (row major); B is walked row-wise so that the inner loop streams
static std::vector<double> matmul(const std::vector<double> &A, const std::vector<double> &B, uint32_t n) {
  std::vector<double> C((size_t)n * n, 0.0);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t k = 0; k < n; ++k) {
      const double a = A[(size_t)i * n + k];
      if (a == 0.0) continue;
      const double *b = &B[(size_t)k * n];
      double *c = &C[(size_t)i * n];
      for (uint32_t j = 0; j < n; ++j) c[j] += a * b[j];
    }
  return C;
}

static std::vector<double> transpose(const std::vector<double> &A, uint32_t n) {
  std::vector<double> T((size_t)n * n);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < n; ++j) T[(size_t)j * n + i] = A[(size_t)i * n + j];
  return T;
}

// inverse of a symmetric positive definite matrix: A = L L^T, A^-1 = L^-T L^-1
static bool spd_inverse(std::vector<double> A, uint32_t n, std::vector<double> &out) {
  if (!cholesky(A, n)) return false;
  const std::vector<double> Li = lower_inverse(A, n);
  out.assign((size_t)n * n, 0.0);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j <= i; ++j) {
      double s = 0.0;
      for (uint32_t k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];  // k >= max(i, j) = i
      out[(size_t)i * n + j] = out[(size_t)j * n + i] = s;
    }
  return true;
}

// GFFAction::buildPrecisionMatrix (gffaction.cc:176-197): stencil[0] on the diagonal, stencil[1] on neighbours 0..3,
// stencil[2] on neighbours 4..7; coinciding neighbours of tiny lattices add up, as Eigen's setFromTriplets does
static std::vector<double> precision_matrix(const std::vector<uint32_t> &nb, uint32_t n, const double *stencil, int n_shells) {
  std::vector<double> Q((size_t)n * n, 0.0);
  for (uint32_t l = 0; l < n; ++l) {
    Q[(size_t)l * n + l] += stencil[0];
    for (int s = 0; s < n_shells; ++s)
      for (int k = 0; k < 4; ++k) Q[(size_t)l * n + nb[8 * (size_t)l + 4 * s + k]] += stencil[s + 1];
  }
  return Q;
}

}  // namespace mlmcpi

using namespace mlmcpi;

struct mlmcpi_gff_level {
  uint32_t Mt = 0, Mx = 0, N = 0;
  int32_t ctype = 0, level = 0, rotated = 0, n_gibbs = 0;
  double mass = 0, mu2 = 0, omega = 1.0;
  // coarsening to the next-coarser level (lattice2d.cc:82-134); empty when the lattice cannot be coarsened
  uint32_t n_coarse = 0, n_fineonly = 0, Mt_c = 0, Mx_c = 0;
  int32_t rotated_c = 0;
  std::vector<uint32_t> nb, fineonly, pairs;  // [8 N], [n_fineonly], [2 n_coarse] = (fine index, coarse index)
  std::vector<double> Qhat, Linv;             // dense [N N]: smoothed precision matrix; inverse of the Cholesky factor of Q
  uint32_t *d_nb = nullptr, *d_fineonly = nullptr, *d_pairs = nullptr;
  double *d_Qhat = nullptr, *d_Linv = nullptr;
};

namespace mlmcpi {

// lattice2d.cc:20-134 for one level: extents and orientation of the next-coarser lattice, the vertices that survive
static bool coarsening(uint32_t Mt, uint32_t Mx, int32_t ctype, int32_t level, bool rotated, uint32_t &mt, uint32_t &mx, int &rt,
                       int &rx) {
  rt = rx = 1;
  bool ok = true;
  switch (ctype) {
    case 0: rt = rx = 2; break;                                     // CoarsenBoth
    case 1: rt = 2; break;                                          // CoarsenTemporal
    case 2: rx = 2; break;                                          // CoarsenSpatial
    case 3: (level % 2 == 0 ? rt : rx) = 2; break;                  // CoarsenAlternate
    case 4:                                                         // CoarsenRotate
      if (rotated) { rt = rx = 2; ok = !((Mt % 2) || (Mx % 2)); }
      break;
    default: ok = false;
  }
  mt = Mt; mx = Mx;
  if (rt > 1) { if (Mt % rt) ok = false; mt = Mt / rt; }
  if (rx > 1) { if (Mx % rx) ok = false; mx = Mx / rx; }
  return ok && mt > 1 && mx > 1;
}

static int build_tables(mlmcpi_gff_level &L) {
  L.nb.resize(8 * (size_t)L.N);
  if (int rc = mlmcpi_neighbours_2d(L.Mt, L.Mx, L.rotated, L.nb.data())) return rc;
  uint32_t mt, mx;
  int rt, rx;
  if (coarsening(L.Mt, L.Mx, L.ctype, L.level, L.rotated, mt, mx, rt, rx)) {
    L.Mt_c = mt; L.Mx_c = mx;
    L.rotated_c = (L.ctype == 4) && ((L.level + 1) % 2);
    for (uint32_t ell = 0; ell < L.N; ++ell) {  // ascending ell = the sorted lists of the reference
      int i, j;
      mlmcpi_vertex_lin2cart(L.Mt, L.Mx, L.rotated, ell, &i, &j);
      bool coarse;
      if (L.ctype == 4) coarse = L.rotated ? (i % 2 == 0 && j % 2 == 0) : ((i + j) % 2 == 0);
      else coarse = (i % rt == 0) && (j % rx == 0);
      if (coarse) {
        L.pairs.push_back(ell);
        L.pairs.push_back(mlmcpi_vertex_cart2lin(mt, mx, L.rotated_c, i / rt, j / rx));
      } else {
        L.fineonly.push_back(ell);
      }
    }
    L.n_coarse = (uint32_t)L.pairs.size() / 2;
    L.n_fineonly = (uint32_t)L.fineonly.size();
  }
  return MLMCPI_OK;
}

// gffaction.cc:126-173
static int build_dense(mlmcpi_gff_level &L, bool want_qhat, bool want_linv) {
  const uint32_t n = L.N;
  if (n > kMaxDense)
    return fail(MLMCPI_ERR_UNSUPPORTED, "GFF level with %u vertices needs dense matrices (limit %u; gffaction.cc:126-173 is dense)", n, kMaxDense);
  const double st[2] = {4. + L.mu2, -1.};
  const std::vector<double> Q = precision_matrix(L.nb, n, st, 1);
  if (want_linv && L.Linv.empty()) {
    std::vector<double> C = Q;
    if (!cholesky(C, n)) return fail(MLMCPI_ERR_INVALID, "GFF precision matrix is not positive definite");
    L.Linv = lower_inverse(C, n);
  }
  if (want_qhat && L.Qhat.empty()) {
    const double h = 4. + 0.5 * L.mu2;
    const double st_eff[3] = {h - 4. / h, -2. / h, -1. / h};
    const std::vector<double> Qeff = precision_matrix(L.nb, n, st_eff, 2);
    std::vector<double> Sigma, Sigma_eff;
    if (!spd_inverse(Q, n, Sigma) || !spd_inverse(Qeff, n, Sigma_eff))
      return fail(MLMCPI_ERR_INVALID, "GFF precision matrix is not positive definite");
    // G = (1 - M^-1 Q_eff)^n_gibbs, M = lower triangle of Q_eff (+ (1/omega - 1) diag)
    std::vector<double> G((size_t)n * n, 0.0);
    for (uint32_t i = 0; i < n; ++i) G[(size_t)i * n + i] = 1.0;
    if (L.n_gibbs > 0) {
      std::vector<double> M((size_t)n * n, 0.0);
      for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j <= i; ++j) M[(size_t)i * n + j] = Qeff[(size_t)i * n + j];
      if (std::fabs(L.omega - 1.0) > 1e-14)
        for (uint32_t i = 0; i < n; ++i) M[(size_t)i * n + i] += (1. / L.omega - 1.) * Qeff[(size_t)i * n + i];
      std::vector<double> Gt = matmul(lower_inverse(M, n), Qeff, n);  // M^-1 Q_eff
      for (size_t k = 0; k < Gt.size(); ++k) Gt[k] = -Gt[k];
      for (uint32_t i = 0; i < n; ++i) Gt[(size_t)i * n + i] += 1.0;
      for (int k = 0; k < L.n_gibbs; ++k) G = matmul(G, Gt, n);
    }
    std::vector<double> D(Sigma);
    for (size_t k = 0; k < D.size(); ++k) D[k] -= Sigma_eff[k];
    std::vector<double> S = matmul(matmul(G, D, n), transpose(G, n), n);
    for (size_t k = 0; k < S.size(); ++k) S[k] += Sigma_eff[k];
    for (uint32_t i = 0; i < n; ++i)  // symmetrise the rounding of the products before the Cholesky factorisation
      for (uint32_t j = 0; j < i; ++j) S[(size_t)i * n + j] = S[(size_t)j * n + i] = 0.5 * (S[(size_t)i * n + j] + S[(size_t)j * n + i]);
    if (!spd_inverse(S, n, L.Qhat)) return fail(MLMCPI_ERR_INVALID, "smoothed GFF covariance is not positive definite");
  }
  return MLMCPI_OK;
}

template <class T>
static int upload(const std::vector<T> &h, T **d) {
  if (*d || h.empty()) return MLMCPI_OK;
  MLMCPI_HIP_TRY(hipMalloc((void **)d, h.size() * sizeof(T)));
  MLMCPI_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return MLMCPI_OK;
}

// ---- kernels --------------------------------------------------------------------------------------------------------------
// gffaction.cc:8-30, stencil branch, neighbours from the table: partial[b] = 1/2 sum phi (kappa phi - sum_nn phi)
__global__ void __launch_bounds__(256) gff_table_energy_kernel(uint32_t n, double kappa, const uint32_t *__restrict__ nb,
                                                               const double *__restrict__ phi_all, double *__restrict__ out) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.x;
  const double *phi = phi_all + (size_t)b * n;
  double acc[1] = {0.0};
  for (uint32_t l = threadIdx.x; l < n; l += blockDim.x) {
    const double v = phi[l];
    double loc = kappa * v;
    for (int k = 0; k < 4; ++k) loc -= phi[nb[8 * (size_t)l + k]];
    acc[0] += v * loc;
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) out[b] = 0.5 * acc[0];
}

// gffaction.cc:26-28: 1/2 phi^T Qhat phi.  One workgroup per chain; phi in LDS; thread i owns row i and reads column i
// of the symmetric matrix (consecutive threads, consecutive addresses).
__global__ void __launch_bounds__(256) gff_dense_energy_kernel(uint32_t n, const double *__restrict__ Q,
                                                               const double *__restrict__ phi_all, double *__restrict__ out) {
  extern __shared__ double sphi[];
  __shared__ double red[4];
  const uint32_t b = blockIdx.x;
  const double *phi = phi_all + (size_t)b * n;
  for (uint32_t l = threadIdx.x; l < n; l += blockDim.x) sphi[l] = phi[l];
  __syncthreads();
  double acc[1] = {0.0};
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    double y = 0.0;
    for (uint32_t j = 0; j < n; ++j) y += Q[(size_t)j * n + i] * sphi[j];
    acc[0] += sphi[i] * y;
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) out[b] = 0.5 * acc[0];
}

// GFFAction::draw (gffaction.cc:200-213): psi ~ N(0, 1) per vertex, solve L^T phi = psi (phi_i = sum_{j >= i} Linv[j][i]
// psi_j), then n_gibbs sweeps of global_heatbath_update_eff (gffaction.cc:45-66) in lexicographic order.  One workgroup
// per chain: the white noise and the normals of a sweep are drawn by all threads into LDS, the triangular product is
// row-per-thread, the Gibbs recurrence (sequential by construction) runs on one thread over the LDS image.
__global__ void __launch_bounds__(256)
    gff_level_draw_kernel(uint32_t n, const double *__restrict__ Linv, const uint32_t *__restrict__ nb, double mu2, double omega,
                          int n_gibbs, double *__restrict__ phi_all, RngKey key0) {
  extern __shared__ double s[];  // psi / normals [n] | phi [n]
  double *noise = s, *sphi = s + n;
  const uint32_t b = blockIdx.x;
  RngKey key = key0;
  key.chain += b;
  auto fill_normals = [&](uint32_t purpose, uint32_t sub) {
    for (uint32_t p = threadIdx.x; 2 * p < n; p += blockDim.x) {
      double n0, n1;
      rng_normals(key, p, purpose, sub, n0, n1);
      noise[2 * p] = n0;
      if (2 * p + 1 < n) noise[2 * p + 1] = n1;
    }
  };
  fill_normals(P_GFF_EXACT, 0);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    double y = 0.0;
    for (uint32_t j = i; j < n; ++j) y += Linv[(size_t)j * n + i] * noise[j];
    sphi[i] = y;
  }
  __syncthreads();
  const double h = 4. + 0.5 * mu2, d0 = h - 4. / h;
  const double sigma_eff = 1. / sqrt(d0), kappa = omega / h, gamma = sqrt(omega * (2. - omega));
  for (int k = 0; k < n_gibbs; ++k) {
    fill_normals(P_GFF_GIBBS, (uint32_t)k);
    __syncthreads();
    if (threadIdx.x == 0) {
      for (uint32_t l = 0; l < n; ++l) {
        const uint32_t *q = nb + 8 * (size_t)l;
        double Delta = (1. - omega) * d0 * sphi[l];
        for (int m = 0; m < 4; ++m) Delta += 2. * kappa * sphi[q[m]];
        for (int m = 4; m < 8; ++m) Delta += kappa * sphi[q[m]];
        sphi[l] = sigma_eff * (gamma * noise[l] + sigma_eff * Delta);
      }
    }
    __syncthreads();
  }
  double *phi = phi_all + (size_t)b * n;
  for (uint32_t l = threadIdx.x; l < n; l += blockDim.x) phi[l] = sphi[l];
}

// GFFAction::copy_from_coarse / copy_from_fine (gffaction.cc:97-118): to_coarse = 0: fine[pair.first] = coarse[pair.second]
__global__ void __launch_bounds__(256) gff_pairs_copy_kernel(uint32_t n_pairs, const uint32_t *__restrict__ pairs, uint32_t n_fine,
                                                             uint32_t n_coarse, double *__restrict__ fine_all,
                                                             double *__restrict__ coarse_all, int to_coarse) {
  const uint32_t b = blockIdx.y;
  double *fine = fine_all + (size_t)b * n_fine, *coarse = coarse_all + (size_t)b * n_coarse;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
    const uint32_t lf = pairs[2 * p], lc = pairs[2 * p + 1];
    if (to_coarse) coarse[lc] = fine[lf]; else fine[lf] = coarse[lc];
  }
}

// GFFConditionedFineAction::fill_fine_points (gffconditionedfineaction.cc:7-26): fine-only vertex l gets
// sigma (n + sigma Delta), Delta = sum of its four nearest neighbours (all of them coarse vertices), n = the Philox normal
// of (site l, P_FILLIN); and ::evaluate (:29-49) of the same state in the same pass: out[b] = S_cfa(state).
template <bool FILL>
__global__ void __launch_bounds__(256)
    gff_cfa_kernel(uint32_t n, uint32_t n_fineonly, const uint32_t *__restrict__ fineonly, const uint32_t *__restrict__ nb, double mu2,
                   double *__restrict__ state_all, double *__restrict__ out, RngKey key0) {
  __shared__ double red[4];
  const uint32_t b = blockIdx.x;
  double *phi = state_all + (size_t)b * n;
  RngKey key = key0;
  key.chain += b;
  const double sigma2 = 1. / (4. + mu2), sigma = sqrt(sigma2), sigma2_inv = 4. + mu2;
  double acc[1] = {0.0};
  for (uint32_t p = threadIdx.x; p < n_fineonly; p += blockDim.x) {
    const uint32_t l = fineonly[p];
    double Delta = 0.0;
    for (int k = 0; k < 4; ++k) Delta += phi[nb[8 * (size_t)l + k]];
    double v;
    if (FILL) {
      v = sigma * (rng_normal0(key, l, P_FILLIN, 0) + sigma * Delta);
      phi[l] = v;
    } else {
      v = phi[l];
    }
    const double dphi = v - sigma2 * Delta;
    acc[0] += 0.5 * sigma2_inv * dphi * dphi;
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) out[b] = acc[0];
}

// twolevelmetropolisstep.cc:46-84.  en = [6][B]: S_f(theta'), S_f(theta), S_c(theta_C), S_c(phi_c), S_cfa(theta), S_cfa(theta')
__global__ void __launch_bounds__(256)
    gff_twolevel_accept_kernel(uint32_t n, double *__restrict__ theta, const double *__restrict__ theta_prime,
                               const double *__restrict__ en, uint32_t B, int32_t *__restrict__ accept, double *__restrict__ terms,
                               RngKey key0) {
  const uint32_t b = blockIdx.y;
  const double dS_fine = en[b] - en[B + b], dS_coarse = en[2 * B + b] - en[3 * B + b], dS_trial = en[4 * B + b] - en[5 * B + b];
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
    if (accept) accept[b] = acc ? 1 : 0;
    if (terms) { terms[3 * b] = dS_fine; terms[3 * b + 1] = dS_coarse; terms[3 * b + 2] = dS_trial; }
  }
  if (!acc) return;
  const size_t off = (size_t)b * n;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) theta[off + l] = theta_prime[off + l];
}

static int level_energy(mlmcpi_gff_level *L, const double *d_phi, uint32_t B, double *d_S, hipStream_t st) {
  if (L->n_gibbs == 0) {
    if (int rc = upload(L->nb, &L->d_nb)) return rc;
    hipLaunchKernelGGL(gff_table_energy_kernel, dim3(B), dim3(256), 0, st, L->N, 4. + L->mu2, (const uint32_t *)L->d_nb, d_phi, d_S);
    MLMCPI_LAUNCH_CHECK("gff_table_energy_kernel");
    return MLMCPI_OK;
  }
  if (int rc = build_dense(*L, true, false)) return rc;
  if (int rc = upload(L->Qhat, &L->d_Qhat)) return rc;
  hipLaunchKernelGGL(gff_dense_energy_kernel, dim3(B), dim3(256), (size_t)L->N * sizeof(double), st, L->N, (const double *)L->d_Qhat,
                     d_phi, d_S);
  MLMCPI_LAUNCH_CHECK("gff_dense_energy_kernel");
  return MLMCPI_OK;
}

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace mlmcpi

extern "C" {

int mlmcpi_gff_level_create(uint32_t Mt, uint32_t Mx, int32_t coarsening_type, int32_t level, double mass, int32_t n_gibbs_smooth,
                            double omega, mlmcpi_gff_level **out) {
  MLMCPI_REQUIRE(out, "out is NULL");
  MLMCPI_REQUIRE(Mt >= 2 && Mx >= 2 && Mt == Mx, "Lattice has to be squared for GFF action (%u x %u)", Mt, Mx);  // gffaction.hh:169-173
  MLMCPI_REQUIRE(coarsening_type >= 0 && coarsening_type <= 4 && level >= 0 && n_gibbs_smooth >= 0, "bad coarsening / smoothing arguments");
  MLMCPI_REQUIRE(omega > 0.0 && omega < 2.0, "SOR parameter omega = %g outside (0, 2)", omega);
  const bool rotated = (coarsening_type == 4) && (level % 2);
  MLMCPI_REQUIRE(!rotated || (Mt % 2 == 0 && Mx % 2 == 0), "Both Mx_lat and Mt_lat have to be even for rotated lattices.");
  mlmcpi_gff_level *L = new mlmcpi_gff_level;
  L->Mt = Mt; L->Mx = Mx; L->ctype = coarsening_type; L->level = level; L->rotated = rotated;
  L->N = rotated ? Mt * Mx / 2 : Mt * Mx;
  L->mass = mass; L->n_gibbs = n_gibbs_smooth; L->omega = omega;
  const double a_lat = rotated ? std::sqrt(2.) / Mt : 1. / Mt;  // gffaction.hh:174-181
  L->mu2 = a_lat * a_lat * mass * mass;
  if (int rc = build_tables(*L)) { delete L; return rc; }
  *out = L;
  return MLMCPI_OK;
}

int mlmcpi_gff_level_destroy(mlmcpi_gff_level *L) {
  if (!L) return MLMCPI_OK;
  for (void *p : {(void *)L->d_nb, (void *)L->d_fineonly, (void *)L->d_pairs, (void *)L->d_Qhat, (void *)L->d_Linv})
    if (p) (void)hipFree(p);
  delete L;
  return MLMCPI_OK;
}

int mlmcpi_gff_level_info(const mlmcpi_gff_level *L, uint32_t *n_vertices, uint32_t *n_coarse, uint32_t *Mt_coarse,
                          uint32_t *Mx_coarse, double *mu2) {
  MLMCPI_REQUIRE(L, "level is NULL");
  if (n_vertices) *n_vertices = L->N;
  if (n_coarse) *n_coarse = L->n_coarse;
  if (Mt_coarse) *Mt_coarse = L->Mt_c;
  if (Mx_coarse) *Mx_coarse = L->Mx_c;
  if (mu2) *mu2 = L->mu2;
  return MLMCPI_OK;
}

int mlmcpi_gff_level_tables(const mlmcpi_gff_level *L, uint32_t *pairs, uint32_t *fineonly) {
  MLMCPI_REQUIRE(L, "level is NULL");
  if (pairs) std::copy(L->pairs.begin(), L->pairs.end(), pairs);
  if (fineonly) std::copy(L->fineonly.begin(), L->fineonly.end(), fineonly);
  return MLMCPI_OK;
}

int mlmcpi_gff_level_matrix(mlmcpi_gff_level *L, int32_t which, double *h_out) {
  MLMCPI_REQUIRE(L && h_out && (which == 0 || which == 1), "bad arguments");
  if (int rc = build_dense(*L, which == 0, which == 1)) return rc;
  const std::vector<double> &M = which == 0 ? L->Qhat : L->Linv;
  std::copy(M.begin(), M.end(), h_out);
  return MLMCPI_OK;
}

int mlmcpi_gff_level_evaluate(mlmcpi_gff_level *L, const double *d_phi, uint32_t B, double *d_S, void *stream) {
  MLMCPI_REQUIRE(L && d_phi && d_S && B > 0, "bad arguments");
  return level_energy(L, d_phi, B, d_S, as_stream(stream));
}

int mlmcpi_gff_level_draw(mlmcpi_gff_level *L, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step,
                          void *stream) {
  MLMCPI_REQUIRE(L && d_phi && B > 0, "bad arguments");
  if (int rc = build_dense(*L, false, true)) return rc;
  if (int rc = upload(L->Linv, &L->d_Linv)) return rc;
  if (int rc = upload(L->nb, &L->d_nb)) return rc;
  const size_t lds = 2 * (size_t)L->N * sizeof(double);
  MLMCPI_HIP_TRY(hipFuncSetAttribute((const void *)gff_level_draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  MLMCPI_REQUIRE(lds <= 160 * 1024, "level too large for the LDS-resident draw (%u vertices)", L->N);
  hipLaunchKernelGGL(gff_level_draw_kernel, dim3(B), dim3(256), lds, as_stream(stream), L->N, (const double *)L->d_Linv,
                     (const uint32_t *)L->d_nb, L->mu2, L->omega, (int)L->n_gibbs, d_phi, make_key(seed, chain0, step));
  MLMCPI_LAUNCH_CHECK("gff_level_draw_kernel");
  return MLMCPI_OK;
}

static int copy_levels(mlmcpi_gff_level *fine, double *d_fine, double *d_coarse, uint32_t B, int to_coarse, hipStream_t st) {
  MLMCPI_REQUIRE(fine && d_fine && d_coarse && B > 0, "bad arguments");
  MLMCPI_REQUIRE(fine->n_coarse > 0, "cannot copy between levels: the lattice has no coarse level");
  if (int rc = upload(fine->pairs, &fine->d_pairs)) return rc;
  const uint32_t n_c = fine->rotated_c ? fine->Mt_c * fine->Mx_c / 2 : fine->Mt_c * fine->Mx_c;
  hipLaunchKernelGGL(gff_pairs_copy_kernel, dim3((fine->n_coarse + 255) / 256, B), dim3(256), 0, st, fine->n_coarse,
                     (const uint32_t *)fine->d_pairs, fine->N, n_c, d_fine, d_coarse, to_coarse);
  MLMCPI_LAUNCH_CHECK("gff_pairs_copy_kernel");
  return MLMCPI_OK;
}

int mlmcpi_gff_copy_from_fine(mlmcpi_gff_level *fine, const double *d_fine, double *d_coarse, uint32_t B, void *stream) {
  return copy_levels(fine, const_cast<double *>(d_fine), d_coarse, B, 1, as_stream(stream));
}

int mlmcpi_gff_copy_from_coarse(mlmcpi_gff_level *fine, const double *d_coarse, double *d_fine, uint32_t B, void *stream) {
  return copy_levels(fine, d_fine, const_cast<double *>(d_coarse), B, 0, as_stream(stream));
}

static int cfa_launch(mlmcpi_gff_level *fine, double *d_state, uint32_t B, double *d_S, bool fill, RngKey key, hipStream_t st) {
  MLMCPI_REQUIRE(fine->n_fineonly > 0, "no fine-only vertices: the lattice has no coarse level");
  if (int rc = upload(fine->nb, &fine->d_nb)) return rc;
  if (int rc = upload(fine->fineonly, &fine->d_fineonly)) return rc;
  if (fill)
    hipLaunchKernelGGL(gff_cfa_kernel<true>, dim3(B), dim3(256), 0, st, fine->N, fine->n_fineonly, (const uint32_t *)fine->d_fineonly,
                       (const uint32_t *)fine->d_nb, fine->mu2, d_state, d_S, key);
  else
    hipLaunchKernelGGL(gff_cfa_kernel<false>, dim3(B), dim3(256), 0, st, fine->N, fine->n_fineonly, (const uint32_t *)fine->d_fineonly,
                       (const uint32_t *)fine->d_nb, fine->mu2, d_state, d_S, key);
  MLMCPI_LAUNCH_CHECK("gff_cfa_kernel");
  return MLMCPI_OK;
}

int mlmcpi_gff_cfa_fill(mlmcpi_gff_level *fine, double *d_state, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step,
                        double *d_S, void *stream) {
  MLMCPI_REQUIRE(fine && d_state && d_S && B > 0, "bad arguments");
  return cfa_launch(fine, d_state, B, d_S, true, make_key(seed, chain0, step), as_stream(stream));
}

int mlmcpi_gff_cfa_evaluate(mlmcpi_gff_level *fine, const double *d_state, uint32_t B, double *d_S, void *stream) {
  MLMCPI_REQUIRE(fine && d_state && d_S && B > 0, "bad arguments");
  return cfa_launch(fine, const_cast<double *>(d_state), B, d_S, false, make_key(0, 0, 0), as_stream(stream));
}

// workspace: theta' [B N_f] | theta_C [B N_c] | energies [6][B]
int mlmcpi_gff_twolevel_workspace_bytes(const mlmcpi_gff_level *fine, uint32_t B, size_t *bytes) {
  MLMCPI_REQUIRE(fine && bytes && B > 0 && fine->n_coarse > 0, "bad arguments");
  const uint32_t n_c = fine->rotated_c ? fine->Mt_c * fine->Mx_c / 2 : fine->Mt_c * fine->Mx_c;
  *bytes = align256((size_t)B * fine->N * 8) + align256((size_t)B * n_c * 8) + align256((size_t)6 * B * 8);
  return MLMCPI_OK;
}

int mlmcpi_gff_twolevel_draw(mlmcpi_gff_level *fine, mlmcpi_gff_level *coarse, const double *d_phi_coarse, double *d_theta, uint32_t B,
                             uint64_t seed, uint32_t chain0, uint32_t step, void *d_work, int32_t *d_accept, double *d_terms,
                             void *stream) {
  MLMCPI_REQUIRE(fine && coarse && d_phi_coarse && d_theta && d_work && B > 0, "bad arguments");
  MLMCPI_REQUIRE(fine->n_coarse > 0 && fine->Mt_c == coarse->Mt && fine->Mx_c == coarse->Mx && fine->rotated_c == coarse->rotated &&
                     fine->n_coarse == coarse->N,
                 "coarse level %u x %u (rotated %d) is not the coarsening of the fine level %u x %u", coarse->Mt, coarse->Mx,
                 coarse->rotated, fine->Mt, fine->Mx);
  hipStream_t st = as_stream(stream);
  char *w = (char *)d_work;
  double *prime = (double *)w;
  double *theta_C = (double *)(w + align256((size_t)B * fine->N * 8));
  double *en = (double *)(w + align256((size_t)B * fine->N * 8) + align256((size_t)B * coarse->N * 8));
  const RngKey key = make_key(seed, chain0, step);
  // theta' = coarse proposal on the coarse vertices + conditioned fill-in on the others; S_cfa(theta') in the same pass
  if (int rc = copy_levels(fine, prime, const_cast<double *>(d_phi_coarse), B, 0, st)) return rc;
  if (int rc = cfa_launch(fine, prime, B, en + 5 * (size_t)B, true, key, st)) return rc;
  if (int rc = cfa_launch(fine, d_theta, B, en + 4 * (size_t)B, false, key, st)) return rc;
  if (int rc = level_energy(fine, prime, B, en, st)) return rc;
  if (int rc = level_energy(fine, d_theta, B, en + (size_t)B, st)) return rc;
  if (int rc = copy_levels(fine, d_theta, theta_C, B, 1, st)) return rc;
  if (int rc = level_energy(coarse, theta_C, B, en + 2 * (size_t)B, st)) return rc;
  if (int rc = level_energy(coarse, d_phi_coarse, B, en + 3 * (size_t)B, st)) return rc;
  uint32_t nb = (fine->N + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(gff_twolevel_accept_kernel, dim3(nb, B), dim3(256), 0, st, fine->N, d_theta, (const double *)prime,
                     (const double *)en, B, d_accept, d_terms, key);
  MLMCPI_LAUNCH_CHECK("gff_twolevel_accept_kernel");
  return MLMCPI_OK;
}

}  // extern "C"
