// lattice.hh -- Lattice1D / Lattice2D mirrors (lattice/lattice1d.hh:60-101, lattice/lattice2d.hh:98-437).
// Index arithmetic is delegated to the C ABI's host functions, which are tested bit for bit against
// the reference's own compiled Lattice classes (tests/test_abi_host.py).  The device kernels
// compute neighbours inline and never read these tables.
#ifndef MLMCPI_LATTICE_HH
#define MLMCPI_LATTICE_HH
#include <memory>
#include <vector>

#include "samplestate.hh"

namespace mlmcpi {

/** lattice/lattice2d.hh:18-26 */
enum CoarseningType {
  CoarsenUnspecified = -1,
  CoarsenBoth = 0,
  CoarsenTemporal = 1,
  CoarsenSpatial = 2,
  CoarsenAlternate = 3,
  CoarsenRotate = 4
};

class Lattice {
public:
  Lattice(const int coarsening_level_ = 0, const int dimension_ = -1)
      : dimension(dimension_), coarsening_level(coarsening_level_) {}
  virtual ~Lattice() {}
  int get_coarsening_level() const { return coarsening_level; }
  virtual unsigned int getNvertices() const = 0;
  const std::vector<std::vector<unsigned int>> &get_neighbour_vertices() { return neighbour_vertices; }
  const int dimension;

protected:
  int coarsening_level;
  std::vector<std::vector<unsigned int>> neighbour_vertices;
};

class Lattice1D : public Lattice {
public:
  Lattice1D(const unsigned int M_lat_, const double T_final_, const int coarsening_level_ = 0)
      : Lattice(coarsening_level_, 1), M_lat(M_lat_), T_final(T_final_), a_lat(T_final_ / M_lat_) {
    if (!(T_final > 0.0)) fatal("T_final has to be positive");
    std::vector<unsigned int> flat(2 * (size_t)M_lat);
    check(mlmcpi_neighbours_1d(M_lat, flat.data()), "neighbours_1d");
    for (unsigned int l = 0; l < M_lat; ++l) neighbour_vertices.push_back({flat[2 * l], flat[2 * l + 1]});
  }
  unsigned int getM_lat() const { return M_lat; }
  double getT_final() const { return T_final; }
  double geta_lat() const { return a_lat; }
  /** lattice1d.hh:80-89 */
  std::shared_ptr<Lattice1D> coarse_lattice() {
    if (M_lat % 2) fatal("cannot coarsen 1d lattice with M = " + std::to_string(M_lat) + " points.");
    return std::make_shared<Lattice1D>(M_lat / 2, T_final, coarsening_level + 1);
  }
  unsigned int getNvertices() const override { return M_lat; }

protected:
  const unsigned int M_lat;
  const double T_final, a_lat;
};

/** Unrotated and rotated periodic 2-D lattices.  The coarse hierarchy (lattice2d.cc:12-82) is built
 *  on demand.  The vertex lists of the 2-D multilevel glue (coarse / fine-only vertices, fine -> coarse map,
 *  lattice2d.cc:82-134) live with the level objects of the library (mlmcpi_gff_level_tables, csrc/gff_levels.hip),
 *  pinned bit for bit to the reference's own class (tests/test_gff_levels.py). */
class Lattice2D : public Lattice {
public:
  Lattice2D(const unsigned int Mt_lat_, const unsigned int Mx_lat_, const CoarseningType coarsening_type_,
            const int coarsening_level_ = 0)
      : Lattice(coarsening_level_, 2), Mt_lat(Mt_lat_), Mx_lat(Mx_lat_), coarsening_type(coarsening_type_),
        rotated((coarsening_type_ == CoarsenRotate) && (coarsening_level_ % 2)) {
    if (rotated && ((Mx_lat % 2) || (Mt_lat % 2))) fatal("Both Mx_lat and Mt_lat have to be even for rotated lattices.");
    std::vector<unsigned int> flat(8 * (size_t)getNvertices());
    check(mlmcpi_neighbours_2d(Mt_lat, Mx_lat, rotated, flat.data()), "neighbours_2d");
    for (unsigned int l = 0; l < getNvertices(); ++l)
      neighbour_vertices.push_back(std::vector<unsigned int>(flat.begin() + 8 * l, flat.begin() + 8 * l + 8));
  }
  unsigned int getMt_lat() const { return Mt_lat; }
  unsigned int getMx_lat() const { return Mx_lat; }
  unsigned int getNedges() const { return rotated ? Mt_lat * Mx_lat : 2 * Mt_lat * Mx_lat; }
  unsigned int getNvertices() const override { return rotated ? Mt_lat * Mx_lat / 2 : Mt_lat * Mx_lat; }
  unsigned int getNcells() const { return getNvertices(); }
  bool is_rotated() const { return rotated; }
  CoarseningType get_coarsening_type() const { return coarsening_type; }
  unsigned int vertex_cart2lin(const int i, const int j) const { return mlmcpi_vertex_cart2lin(Mt_lat, Mx_lat, rotated, i, j); }
  void vertex_lin2cart(const unsigned int ell, int &i, int &j) const { mlmcpi_vertex_lin2cart(Mt_lat, Mx_lat, rotated, ell, &i, &j); }
  unsigned int link_cart2lin(const int i, const int j, const int mu) const {
    if (rotated) fatal("links can only be handled on non-rotated lattices");
    return mlmcpi_link_cart2lin(Mt_lat, Mx_lat, i, j, mu);
  }
  void link_lin2cart(const unsigned int ell, int &i, int &j, int &mu) const {
    if (rotated) fatal("links can only be handled on non-rotated lattices");
    mlmcpi_link_lin2cart(Mt_lat, Mx_lat, ell, &i, &j, &mu);
  }
  /** lattice2d.cc:20-82: extents of the next-coarser lattice, nullptr if it cannot be coarsened */
  std::shared_ptr<Lattice2D> get_coarse_lattice() {
    int rt = 1, rx = 1;
    bool ok = true;
    switch (coarsening_type) {
      case CoarsenBoth: rt = rx = 2; break;
      case CoarsenTemporal: rt = 2; break;
      case CoarsenSpatial: rx = 2; break;
      case CoarsenAlternate: (coarsening_level % 2 == 0 ? rt : rx) = 2; break;
      case CoarsenRotate:
        if (rotated) { rt = rx = 2; ok = !((Mt_lat % 2) || (Mx_lat % 2)); }
        break;
      default: ok = false;
    }
    unsigned int mt = Mt_lat, mx = Mx_lat;
    if (rt > 1) { if (Mt_lat % rt) ok = false; mt = Mt_lat / rt; }
    if (rx > 1) { if (Mx_lat % rx) ok = false; mx = Mx_lat / rx; }
    if (!(ok && mt > 1 && mx > 1)) return nullptr;
    return std::make_shared<Lattice2D>(mt, mx, coarsening_type, coarsening_level + 1);
  }

protected:
  const unsigned int Mt_lat, Mx_lat;
  const CoarseningType coarsening_type;
  bool rotated;
};

}  // namespace mlmcpi
#endif
