// ref_wrap.cc -- thin C surface over the REAL reference classes that build here without Eigen/GSL
// (lattice/lattice1d.cc, lattice/lattice2d.cc, common/statistics.cc, mpi/mpi_random.cc + their dependencies
// common/parameters.cc and mpi/mpi_wrapper.cc, compiled unmodified from /root/reference/src by
// oracle/Makefile into oracle/_ref/libref.so).  Written for this repository; it contains no
// reference code, only calls into it.  TEST INFRASTRUCTURE ONLY: used by tests/ to pin the
// restated index maps and statistics estimators in oracle.cc (and through them the HIP path)
// against the reference itself, bit for bit.
#include "common/statistics.hh"
#include "lattice/lattice1d.hh"
#include "lattice/lattice2d.hh"
#include "mpi/mpi_random.hh"
#include <memory>

extern "C" {

void *ref_lattice2d_new(unsigned Mt, unsigned Mx, int coarsening_type, int level) {
  return new std::shared_ptr<Lattice2D>(std::make_shared<Lattice2D>(Mt, Mx, (CoarseningType)coarsening_type, level));
}
void ref_lattice2d_free(void *h) { delete (std::shared_ptr<Lattice2D> *)h; }
static Lattice2D &L2(void *h) { return **(std::shared_ptr<Lattice2D> *)h; }
// walk down the coarse hierarchy; returns NULL when there is no coarser lattice
void *ref_lattice2d_coarse(void *h) {
  std::shared_ptr<Lattice2D> c = L2(h).get_coarse_lattice();
  return c ? new std::shared_ptr<Lattice2D>(c) : nullptr;
}
unsigned ref_lattice2d_Mt(void *h) { return L2(h).getMt_lat(); }
unsigned ref_lattice2d_Mx(void *h) { return L2(h).getMx_lat(); }
int ref_lattice2d_rotated(void *h) { return L2(h).is_rotated() ? 1 : 0; }
unsigned ref_lattice2d_nvertices(void *h) { return L2(h).getNvertices(); }
unsigned ref_lattice2d_nedges(void *h) { return L2(h).getNedges(); }
unsigned ref_vertex_cart2lin(void *h, int i, int j) { return L2(h).vertex_cart2lin(i, j); }
void ref_vertex_lin2cart(void *h, unsigned l, int *i, int *j) { L2(h).vertex_lin2cart(l, *i, *j); }
unsigned ref_link_cart2lin(void *h, int i, int j, int mu) { return L2(h).link_cart2lin(i, j, mu); }
void ref_link_lin2cart(void *h, unsigned l, int *i, int *j, int *mu) { L2(h).link_lin2cart(l, *i, *j, *mu); }
void ref_lattice2d_neighbours(void *h, unsigned *out) {  // nvertices x 8
  const auto &nb = L2(h).get_neighbour_vertices();
  for (size_t l = 0; l < nb.size(); ++l)
    for (size_t k = 0; k < nb[l].size(); ++k) out[8 * l + k] = nb[l][k];
}

// lattice2d.cc:82-134: the vertex lists that drive the 2-D multilevel glue
unsigned ref_lattice2d_n_fineonly(void *h) { return (unsigned)L2(h).get_fineonly_vertices().size(); }
void ref_lattice2d_fineonly(void *h, unsigned *out) {
  const auto &v = L2(h).get_fineonly_vertices();
  for (size_t k = 0; k < v.size(); ++k) out[k] = v[k];
}
unsigned ref_lattice2d_n_coarse(void *h) { return (unsigned)L2(h).get_fine2coarse_map().size(); }
void ref_lattice2d_fine2coarse(void *h, unsigned *out) {  // (fine, coarse) pairs in key order
  size_t k = 0;
  for (const auto &p : L2(h).get_fine2coarse_map()) { out[k++] = p.first; out[k++] = p.second; }
}

void ref_lattice1d_neighbours(unsigned M, double T_final, unsigned *out, double *a_lat) {  // M x 2
  Lattice1D lat(M, T_final);
  const auto &nb = lat.get_neighbour_vertices();
  for (size_t l = 0; l < nb.size(); ++l) {
    out[2 * l] = nb[l][0];
    out[2 * l + 1] = nb[l][1];
  }
  *a_lat = lat.geta_lat();
}

void *ref_stats_new(unsigned k_max) { return new Statistics("Q", k_max); }
void ref_stats_free(void *h) { delete (Statistics *)h; }
void ref_stats_record(void *h, const double *q, unsigned n) {
  for (unsigned i = 0; i < n; ++i) ((Statistics *)h)->record_sample(q[i]);
}
void ref_stats_reset(void *h, int hard) {
  if (hard) ((Statistics *)h)->hard_reset(); else ((Statistics *)h)->reset();
}
void ref_stats_get(void *h, double *out) {
  Statistics *s = (Statistics *)h;
  out[0] = s->average(); out[1] = s->variance(); out[2] = s->variance_error(); out[3] = s->tau_int();
  out[4] = s->error(); out[5] = (double)s->samples();
}
void ref_stats_autocorr(void *h, double *out) {  // k_max values
  const std::vector<double> c = ((Statistics *)h)->auto_corr();
  for (size_t k = 0; k < c.size(); ++k) out[k] = c[k];
}
unsigned ref_distribute_n(unsigned n) { return distribute_n(n); }

// mpi/mpi_random.cc:5-29 as shipped (this build has no USE_MPI: one rank, the seed list is {value})
void ref_parallel_mt19937_64(unsigned long long seed, unsigned n, unsigned long long *out) {
  parallel_mt19937_64 engine(seed);
  for (unsigned i = 0; i < n; ++i) out[i] = engine();
}

}  // extern "C"
