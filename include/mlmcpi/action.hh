// action.hh -- Action interface (action/action.hh:28-163) and the five actions of the sweep path,
// evaluated on the device through the C ABI.  Signatures follow the reference; states may hold a
// batch of chains, in which case evaluate() returns the value for chain 0 and evaluate_batch() all.
#ifndef MLMCPI_ACTION_HH
#define MLMCPI_ACTION_HH
#include <cmath>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "lattice.hh"

namespace mlmcpi {

/** action/renormalisation.hh:17-21 */
enum RenormalisationType { RenormalisationNone = 0, RenormalisationPerturbative = 1, RenormalisationNonperturbative = 2 };

class Action {
public:
  Action(const RenormalisationType renormalisation_ = RenormalisationNone) : renormalisation(renormalisation_) {}
  virtual ~Action() {}
  virtual unsigned int sample_size() const = 0;
  virtual double evaluation_cost() const { return sample_size(); }
  virtual std::shared_ptr<Action> coarse_action() { fatal("cannot coarsen action"); }

  /** action.hh:60-61 */
  virtual const double evaluate(const std::shared_ptr<SampleState> phi_state) const {
    return evaluate_batch(phi_state)[0];
  }
  virtual std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi_state) const = 0;
  /** action.hh:112-113 */
  virtual void force(const std::shared_ptr<SampleState> phi_state, std::shared_ptr<SampleState> p_state) const = 0;
  /** action.hh:122-123; chains of a batch get the Philox streams chain0, chain0+1, ... */
  virtual void initialise_state(std::shared_ptr<SampleState> phi_state) const = 0;

  /** action.hh:73-96: the update of site / link `ell`, on every chain of the state.  Site-at-a-time updates are the
   *  reference's CPU inner loop; the device's unit of work is a whole multicolour sweep (OverrelaxedHeatBathSampler::draw),
   *  but a caller that walks an index set itself is served: actions with local updates forward to
   *  mlmcpi_{path,lattice}_site_updates (one thread per chain, sequential within a chain as in the reference).  Every call
   *  draws from fresh Philox numbers, like an engine: (site, chain, site_step) with site_step advancing per call. */
  virtual void heatbath_update(std::shared_ptr<SampleState> phi_state, const unsigned int ell) {
    site_updates(phi_state, nullptr, 1, ell, true);
  }
  virtual void overrelaxation_update(std::shared_ptr<SampleState> phi_state, const unsigned int ell) {
    site_updates(phi_state, nullptr, 1, ell, false);
  }
  /** The same for a whole index list in ONE call: d_sites = n site indices in device memory, visited in list order (the
   *  loops of overrelaxedheatbathsampler.cc:8-31 over a lexicographic or shuffled index set); the sites of a list should
   *  be distinct -- they share one Philox step, which advances per call. */
  virtual void site_updates(std::shared_ptr<SampleState>, const uint32_t *, unsigned int, unsigned int, bool heat) {
    fatal(std::string(heat ? "heat bath" : "overrelaxation") + " update not implemented for this action ");
  }
  /** the Philox step the next site-at-a-time call draws from */
  void set_site_step(uint32_t s) { site_step = s; }
  uint32_t get_site_step() const { return site_step; }
  /** Does OverrelaxedHeatBathSampler work with this action? */
  virtual bool has_local_updates() const { return false; }
  /** Device sweeps: n_overrelax overrelaxation then n_heatbath heat-bath sweeps. */
  virtual void sweep(std::shared_ptr<SampleState>, std::shared_ptr<SampleState>, unsigned, unsigned, uint32_t) {
    fatal("overrelaxation update not implemented for this action");
  }
  /** The same sweeps with the input left untouched: reads d_src, alternates between the work buffers d_w0 and d_w1 (d_w1
   *  may be d_src); returns 0 / 1 = the work buffer that holds the result.  batch = number of chains. */
  virtual int sweep_from(const double *, double *, double *, unsigned, unsigned, unsigned, uint32_t) {
    fatal("overrelaxation update not implemented for this action");
  }
  /** sweep_from with the QoI of the new sample summed inside the last launch (qoi_kind: QoI::fused_kind()); returns -1
   *  when the action cannot fuse it (the caller then evaluates the QoI on its own) */
  virtual int sweep_from_qoi(const double *, double *, double *, unsigned, unsigned, unsigned, uint32_t, int, double *, double * = nullptr) { return -1; }
  /** action.hh:130-143: transfers between this level and the next coarser / finer one */
  virtual void copy_from_coarse(const std::shared_ptr<SampleState>, std::shared_ptr<SampleState>) { fatal("cannot copy from coarse lattice."); }
  virtual void copy_from_fine(const std::shared_ptr<SampleState>, std::shared_ptr<SampleState>) { fatal("cannot copy from fine lattice."); }
  const std::vector<unsigned int> &get_heatbath_indexset() const { return heatbath_indexset; }
  virtual int get_coarsening_level() const = 0;
  virtual std::string info_string() const = 0;

  /** RNG stream selection for the device chains of this action */
  void set_seed(uint64_t seed_, uint32_t chain0_ = 0) { seed = seed_; chain0 = chain0_; }
  uint64_t get_seed() const { return seed; }
  uint32_t get_chain0() const { return chain0; }

protected:
  const RenormalisationType renormalisation;
  std::vector<unsigned int> heatbath_indexset;
  uint64_t seed = 2481317;  // the reference's Schwinger / GFF engine seed, reused as the Philox key
  uint32_t chain0 = 0;
  uint32_t site_step = 0x40000000u;  // site-at-a-time calls: far from the sweep counters of the samplers
};

// ---- quantum mechanics: 1-D paths (action/qm/qmaction.hh:79-215) ---------------------------------
class QMAction : public Action {
public:
  QMAction(const std::shared_ptr<Lattice1D> lattice_, const RenormalisationType r, int kind, double m0_, double mu2_ = 0,
           double lambda_ = 0, double x0_ = 0)
      : Action(r), lattice(lattice_), M_lat(lattice_->getM_lat()), T_final(lattice_->getT_final()),
        a_lat(lattice_->geta_lat()), m0(m0_) {
    abi.kind = kind; abi.M = M_lat; abi.T_final = T_final; abi.m0 = m0_; abi.mu2 = mu2_; abi.lambda = lambda_; abi.x0 = x0_;
  }
  unsigned int sample_size() const override { return M_lat; }
  double evaluation_cost() const override { return M_lat; }
  std::shared_ptr<Lattice1D> get_lattice() { return lattice; }
  double getT_final() const { return T_final; }
  double getm0() const { return m0; }
  int get_coarsening_level() const override { return lattice->get_coarsening_level(); }
  const mlmcpi_path_action &abi_action() const { return abi; }

  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> x) const override {
    DeviceVector out(x->batch());
    check(mlmcpi_path_evaluate(&abi, x->device(), x->batch(), (double *)out.ptr(), nullptr), "path_evaluate");
    return out.download<double>();
  }
  void force(const std::shared_ptr<SampleState> x, std::shared_ptr<SampleState> p) const override {
    check(mlmcpi_path_force(&abi, x->device(), p->device_mutable(), x->batch(), nullptr), "path_force");
  }
  void initialise_state(std::shared_ptr<SampleState> x) const override {
    check(mlmcpi_path_initialise(&abi, x->device_mutable(), x->batch(), seed, chain0, nullptr), "path_initialise");
  }
  /** action/qm/qmaction.cc:7-24 */
  void copy_from_coarse(const std::shared_ptr<SampleState> x_coarse, std::shared_ptr<SampleState> x_path) override {
    if (x_path->size() != M_lat || 2 * x_coarse->size() != M_lat) fatal("cannot copy from coarse lattice.");
    check(mlmcpi_path_copy_from_coarse(x_coarse->device(), x_path->device_mutable(), M_lat / 2, x_path->batch(), nullptr), "path_copy_from_coarse");
  }
  void copy_from_fine(const std::shared_ptr<SampleState> x_fine, std::shared_ptr<SampleState> x_path) override {
    if (x_path->size() != M_lat || x_fine->size() != 2 * M_lat) fatal("cannot copy from fine lattice.");
    check(mlmcpi_path_copy_from_fine(x_fine->device(), x_path->device_mutable(), M_lat, x_path->batch(), nullptr), "path_copy_from_fine");
  }
  std::string info_string() const override {
    std::stringstream s;
    s << "M_lat = " << M_lat << ", T_final = " << T_final << ", a_lat = " << a_lat << ", m0 = " << m0;
    return s.str();
  }

protected:
  const std::shared_ptr<Lattice1D> lattice;
  const unsigned int M_lat;
  const double T_final, a_lat, m0;
  mlmcpi_path_action abi;
};

/** action/qm/harmonicoscillatoraction.hh:83-262 (sweep-path part) */
class HarmonicOscillatorAction : public QMAction {
public:
  HarmonicOscillatorAction(const std::shared_ptr<Lattice1D> lattice_, const RenormalisationType r, const double m0_,
                           const double mu2_)
      : QMAction(lattice_, r, MLMCPI_HARMONIC, m0_, mu2_), mu2(mu2_) {}
  /** harmonicoscillatoraction.cc:69-74 */
  double Xsquared_analytical() const {
    const double R = 1. + 0.5 * a_lat * a_lat * mu2 - a_lat * std::sqrt(mu2) * std::sqrt(1. + 0.25 * a_lat * a_lat * mu2);
    return 1. / (2. * m0 * std::sqrt(mu2) * std::sqrt(1 + 0.25 * a_lat * a_lat * mu2)) * (1. + std::pow(R, M_lat)) /
           (1. - std::pow(R, M_lat));
  }
  /** harmonicoscillatoraction.hh:115-122 with RenormalisedHOParameters (harmonicoscillatorrenormalisation.hh:36-67) */
  std::shared_ptr<Action> coarse_action() override {
    double m0c = m0, mu2c = mu2;
    if (renormalisation == RenormalisationPerturbative) {
      m0c = m0 * (1. - 0.5 * a_lat * a_lat * mu2);
      mu2c = mu2 * (1. + 0.25 * a_lat * a_lat * mu2);
    } else if (renormalisation == RenormalisationNonperturbative) {
      m0c = m0 / (1. + 0.5 * a_lat * a_lat * mu2);
      mu2c = mu2 * (1. + 0.25 * a_lat * a_lat * mu2);
    }
    return std::make_shared<HarmonicOscillatorAction>(lattice->coarse_lattice(), renormalisation, m0c, mu2c);
  }
  const double mu2;
};

/** action/qm/quarticoscillatoraction.hh:78-203 */
class QuarticOscillatorAction : public QMAction {
public:
  QuarticOscillatorAction(const std::shared_ptr<Lattice1D> lattice_, const RenormalisationType r, const double m0_,
                          const double mu2_, const double lambda_, const double x0_)
      : QMAction(lattice_, r, MLMCPI_QUARTIC, m0_, mu2_, lambda_, x0_), mu2(mu2_), lambda(lambda_), x0(x0_) {}
  /** quarticoscillatoraction.hh:105-110: coarse levels keep the parameters */
  std::shared_ptr<Action> coarse_action() override {
    return std::make_shared<QuarticOscillatorAction>(lattice->coarse_lattice(), renormalisation, m0, mu2, lambda, x0);
  }
  const double mu2, lambda, x0;
};

/** common/auxilliary.cc:197-209: <phi^2> of the Gaussian free field on the periodic Mt x Mx lattice */
inline double gff_phi_squared_analytical(const double mass, const int Mt_lat, const int Mx_lat) {
  const double mu2 = mass * mass / (1.0 * Mt_lat * Mx_lat);
  double s = 0.0;
  for (int k1 = 0; k1 < Mt_lat; ++k1)
    for (int k2 = 0; k2 < Mx_lat; ++k2) {
      const double s1 = std::sin(M_PI * k1 / Mt_lat), s2 = std::sin(M_PI * k2 / Mx_lat);
      s += 1. / (4. * (s1 * s1 + s2 * s2) + mu2);
    }
  return s / (1.0 * Mt_lat * Mx_lat);
}

/** common/auxilliary.cc:7-27: ratio of theta-function sums  sum_m m^p exp(-xi m^2/2) / sum_m exp(-xi m^2/2), m in Z
 *  (99 terms on each side, as in the reference) */
inline double Sigma_hat(const double xi, const unsigned int p) {
  if (p % 2) return 0.0;
  if (p == 0) return 1.0;
  double num = 0.0, denom = 1.0;
  for (unsigned int m = 1; m < 100; ++m) {
    const double e = std::exp(-0.5 * xi * m * m);
    num += 2. * std::pow((double)m, (double)p) * e;
    denom += 2. * e;
  }
  return num / denom;
}

/** action/qm/rotoraction.hh:93-289 */
class RotorAction : public QMAction {
public:
  RotorAction(const std::shared_ptr<Lattice1D> lattice_, const RenormalisationType r, const double m0_)
      : QMAction(lattice_, r, MLMCPI_ROTOR, m0_) {
    seed = 21172817;  // rotoraction.hh:106
  }
  /** rotoraction.hh:120-125 with RenormalisedRotorParameters (rotorrenormalisation.hh:22-71, .cc:8-14):
   *  m0_coarse = (1 + delta_I(T/m0) a/m0) m0 under perturbative renormalisation */
  std::shared_ptr<Action> coarse_action() override {
    double m0c = m0;
    if (renormalisation == RenormalisationPerturbative) {
      const double xi = lattice->getT_final() / m0, s2 = Sigma_hat(xi, 2), s4 = Sigma_hat(xi, 4);
      const double delta_I = 0.5 * (1. - 2. * xi * s2 + 0.5 * xi * xi * (s4 - s2 * s2)) / (1. - 2. * xi * s2 + xi * xi * (s4 - s2 * s2));
      m0c = (1. + delta_I * a_lat / m0) * m0;
    } else if (renormalisation == RenormalisationNonperturbative) {
      fatal("nonperturbative renormalisation not implemented for rotor action ");
    }
    return std::make_shared<RotorAction>(lattice->coarse_lattice(), renormalisation, m0c);
  }
  /** rotoraction.cc:97-115: chi_t to O(a/m0), and its continuum limit */
  double chit_perturbative() const {
    const double xi = lattice->getT_final() / m0, z = a_lat / m0, s2 = Sigma_hat(xi, 2), s4 = Sigma_hat(xi, 4);
    return 1. / (4. * M_PI * M_PI * m0) * (1. - xi * s2 + (0.5 - xi * s2 + 0.25 * xi * xi * (s4 - s2 * s2)) * z);
  }
  double chit_continuum() const {
    const double xi = lattice->getT_final() / m0;
    return 1. / (4. * M_PI * M_PI * m0) * (1. - xi * Sigma_hat(xi, 2));
  }
  bool has_local_updates() const override { return true; }
  /** rotoraction.cc:20-56 */
  void site_updates(std::shared_ptr<SampleState> x, const uint32_t *d_sites, unsigned int n, unsigned int ell, bool heat) override {
    if (x->size() != M_lat) fatal("site update on a path of wrong size.");
    check(mlmcpi_path_site_updates(&abi, x->device_mutable(), x->batch(), d_sites, n, ell, heat ? 1 : 0, seed, chain0, site_step++,
                                   nullptr), "path_site_updates");
  }
  void sweep(std::shared_ptr<SampleState> x, std::shared_ptr<SampleState> scratch, unsigned n_or, unsigned n_hb,
             uint32_t sweep0) override {
    check(mlmcpi_path_sweep_draw(&abi, x->device_mutable(), scratch->device_mutable(), x->batch(), n_or, n_hb, seed,
                                 chain0, sweep0, nullptr), "path_sweep_draw");
  }
  int sweep_from(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb,
                 uint32_t sweep0) override {
    int32_t where = 0;
    check(mlmcpi_path_sweep_draw_from(&abi, d_src, d_w0, d_w1, batch, n_or, n_hb, seed, chain0, sweep0, &where, nullptr),
          "path_sweep_draw_from");
    return where;
  }
  /** kind 4 = QoISusceptibility, summed inside the draw's last launch (and recorded into d_acc, if given) */
  int sweep_from_qoi(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb, uint32_t sweep0,
                     int qoi_kind, double *d_q, double *d_acc = nullptr) override {
    if (qoi_kind != 4 || n_or + n_hb == 0) return -1;
    int32_t where = 0;
    check(mlmcpi_path_sweep_draw_qoi(&abi, d_src, d_w0, d_w1, batch, n_or, n_hb, seed, chain0, sweep0, d_q, d_acc, &where, nullptr),
          "path_sweep_draw_qoi");
    return where;
  }
  /** rotoraction.hh:195-213 */
  double getWcurvature(const double x_m, const double x_p) const { return 2.0 * m0 / a_lat * std::fabs(std::cos(0.5 * (x_p - x_m))); }
  double getWminimum(const double x_m, const double x_p) const {
    return std::atan2(std::sin(x_p) + std::sin(x_m), std::cos(x_p) + std::cos(x_m));
  }
};

// ---- quantum field theory: 2-D lattices (action/qft/qftaction.hh:79-120) -------------------------------
class QFTAction : public Action {
public:
  QFTAction(const std::shared_ptr<Lattice2D> lattice_, const std::shared_ptr<Lattice2D> fine_lattice_,
            const RenormalisationType r, int kind, double beta, double mass)
      : Action(r), lattice(lattice_), fine_lattice(fine_lattice_) {
    abi.kind = kind; abi.Mt = lattice->getMt_lat(); abi.Mx = lattice->getMx_lat(); abi.beta = beta; abi.mass = mass;
  }
  std::shared_ptr<Lattice2D> get_lattice() { return lattice; }
  int get_coarsening_level() const override { return lattice->get_coarsening_level(); }
  const mlmcpi_lattice_action &abi_action() const { return abi; }
  bool has_local_updates() const override { return true; }

  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi) const override {
    DeviceVector out(phi->batch());
    check(mlmcpi_lattice_evaluate(&abi, phi->device(), phi->batch(), (double *)out.ptr(), nullptr), "lattice_evaluate");
    return out.download<double>();
  }
  void force(const std::shared_ptr<SampleState> phi, std::shared_ptr<SampleState> p) const override {
    check(mlmcpi_lattice_force(&abi, phi->device(), p->device_mutable(), phi->batch(), nullptr), "lattice_force");
  }
  void initialise_state(std::shared_ptr<SampleState> phi) const override {
    check(mlmcpi_lattice_initialise(&abi, phi->device_mutable(), phi->batch(), seed, chain0, nullptr), "lattice_initialise");
  }
  /** gffaction.cc:33-42,68-77; quenchedschwingeraction.cc:46-65 */
  void site_updates(std::shared_ptr<SampleState> phi, const uint32_t *d_sites, unsigned int n, unsigned int ell, bool heat) override {
    if (phi->size() != sample_size()) fatal("site update on a state of wrong size.");
    check(mlmcpi_lattice_site_updates(&abi, phi->device_mutable(), phi->batch(), d_sites, n, ell, heat ? 1 : 0, seed, chain0,
                                      site_step++, nullptr), "lattice_site_updates");
  }
  void sweep(std::shared_ptr<SampleState> phi, std::shared_ptr<SampleState> scratch, unsigned n_or, unsigned n_hb,
             uint32_t sweep0) override {
    int32_t in_scratch = 0;
    check(mlmcpi_lattice_sweep_draw_pingpong(&abi, phi->device_mutable(), scratch->device_mutable(), phi->batch(), n_or,
                                             n_hb, seed, chain0, sweep0, fuse, &in_scratch, nullptr), "lattice_sweep_draw");
    if (in_scratch) phi->swap_device(*scratch);  // no copy: the buffers exchange roles
  }
  int sweep_from(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb,
                 uint32_t sweep0) override {
    int32_t where = 0;
    check(mlmcpi_lattice_sweep_draw_from(&abi, d_src, d_w0, d_w1, batch, n_or, n_hb, seed, chain0, sweep0, fuse, &where, nullptr),
          "lattice_sweep_draw_from");
    return where;
  }
  int sweep_from_qoi(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb, uint32_t sweep0,
                     int qoi_kind, double *d_q, double *d_acc = nullptr) override {
    // kinds 1, 2 (plaquette QoIs) belong to the Schwinger action, 3 (phi^2) to the GFF
    if (n_hb == 0 || qoi_kind == 0 || (qoi_kind == 3) != (abi.kind == MLMCPI_GFF)) return -1;
    int32_t where = 0;
    if (d_acc)  // record_sample in the same call (per-chain moments on the device)
      check(mlmcpi_lattice_sweep_draw_qoi_record(&abi, d_src, d_w0, d_w1, batch, n_or, n_hb, seed, chain0, sweep0, fuse, qoi_kind, d_q,
                                                 d_acc, &where, nullptr), "lattice_sweep_draw_qoi_record");
    else
      check(mlmcpi_lattice_sweep_draw_qoi(&abi, d_src, d_w0, d_w1, batch, n_or, n_hb, seed, chain0, sweep0, fuse, qoi_kind, d_q, &where,
                                          nullptr), "lattice_sweep_draw_qoi");
    return where;
  }
  /** quenchedschwingeraction.cc:92-195, gffaction.cc:97-118: `this` is the level being written to */
  void copy_from_coarse(const std::shared_ptr<SampleState> phi_coarse, std::shared_ptr<SampleState> phi_state) override {
    unsigned rt = 0, rx = 0;
    if (!factors(lattice, phi_coarse, rt, rx)) fatal("cannot copy from coarse lattice.");
    check(mlmcpi_lattice_copy_from_coarse(&abi, rt, rx, phi_coarse->device(), phi_state->device_mutable(), phi_state->batch(), nullptr), "lattice_copy_from_coarse");
  }
  void copy_from_fine(const std::shared_ptr<SampleState> phi_fine, std::shared_ptr<SampleState> phi_state) override {
    if (!fine_lattice) fatal("cannot copy from fine lattice.");
    mlmcpi_lattice_action fine_abi = abi;
    fine_abi.Mt = fine_lattice->getMt_lat();
    fine_abi.Mx = fine_lattice->getMx_lat();
    const unsigned rt = fine_abi.Mt / abi.Mt, rx = fine_abi.Mx / abi.Mx;
    if (rt * abi.Mt != fine_abi.Mt || rx * abi.Mx != fine_abi.Mx || rt * rx < 2 || rt > 2 || rx > 2) fatal("cannot copy from fine lattice.");
    check(mlmcpi_lattice_copy_from_fine(&fine_abi, rt, rx, phi_fine->device(), phi_state->device_mutable(), phi_state->batch(), nullptr), "lattice_copy_from_fine");
  }
  std::string info_string() const override {
    std::stringstream s;
    s << "Mt_lat = " << abi.Mt << ", Mx_lat = " << abi.Mx;
    return s.str();
  }
  unsigned fuse = 0;  // sweeps fused per launch (0 = library default); results do not depend on it

protected:
  /** coarsening factors between `fine` (this level's lattice) and the state of the next-coarser level */
  bool factors(const std::shared_ptr<Lattice2D> fine, const std::shared_ptr<SampleState> coarse_state, unsigned &rt, unsigned &rx) const {
    auto c = fine->get_coarse_lattice();
    if (!c || c->is_rotated()) return false;
    rt = fine->getMt_lat() / c->getMt_lat();
    rx = fine->getMx_lat() / c->getMx_lat();
    const unsigned per = (abi.kind == MLMCPI_SCHWINGER) ? 2u : 1u;
    return coarse_state->size() == per * c->getMt_lat() * c->getMx_lat() && rt * rx >= 2;
  }
  const std::shared_ptr<Lattice2D> lattice, fine_lattice;
  mlmcpi_lattice_action abi;
};

/** action/qft/gffaction.hh:152-350 with n_gibbs_smooth = 0.  The dense N x N matrices the reference
 *  builds in its constructor (gffaction.cc:133-174) are not needed by any sweep-path method and are
 *  not built (they make the reference unusable beyond ~64 x 64). */
class GFFAction : public QFTAction {
public:
  /** gffaction.hh:120-146.  n_gibbs_smooth > 0: the coarse action of a hierarchy, phi^T Qhat phi with the dense smoothed
   *  precision matrix of gffaction.cc:126-173 (levels of up to 4096 vertices).  Rotated lattices (odd levels of
   *  CoarsenRotate) and smoothed levels run through the level object of the C ABI (mlmcpi_gff_level_*); the plain
   *  unrotated level -- the finest level of a run -- keeps the stencil kernels and the FFT sampler of mlmcpi_lattice_*. */
  GFFAction(const std::shared_ptr<Lattice2D> lattice_, const std::shared_ptr<Lattice2D> fine_lattice_, const double mass_,
            const int n_gibbs_smooth_ = 0, const double omega_ = 1.0)
      : QFTAction(lattice_, fine_lattice_, RenormalisationNone, MLMCPI_GFF, 0.0, mass_), mass(mass_), n_gibbs_smooth(n_gibbs_smooth_),
        omega(omega_) {
    if (lattice->getMt_lat() != lattice->getMx_lat()) fatal("Lattice has to be squared for GFF action ");
    const double a = (lattice->is_rotated() ? std::sqrt(2.) : 1.) / lattice->getMt_lat();  // gffaction.hh:174-181
    mu2 = a * a * mass * mass;
    check(mlmcpi_gff_level_create(lattice->getMt_lat(), lattice->getMx_lat(), (int32_t)lattice->get_coarsening_type(),
                                  lattice->get_coarsening_level(), mass, n_gibbs_smooth, omega, &level), "gff_level_create");
  }
  ~GFFAction() override {
    mlmcpi_gff_level_destroy(level);
    mlmcpi_gff_level_destroy(fine_level);
  }
  unsigned int sample_size() const override { return lattice->getNvertices(); }
  double getmu2() const { return mu2; }
  /** the level whose kernels need no table: unrotated, not smoothed */
  bool plain() const { return !lattice->is_rotated() && n_gibbs_smooth == 0; }
  mlmcpi_gff_level *level_handle() const { return level; }
  /** gffaction.hh:201-208 */
  std::shared_ptr<Action> coarse_action() override {
    std::shared_ptr<Lattice2D> coarse_lattice = lattice->get_coarse_lattice();
    if (!coarse_lattice)
      fatal("cannot coarsen 2d lattice with M_{t,lat} = " + std::to_string(lattice->getMt_lat()) + " , M_{x,lat} = " + std::to_string(lattice->getMx_lat()) + ".");
    return std::make_shared<GFFAction>(coarse_lattice, lattice, mass, 2, 1.0);
  }
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi) const override {
    if (plain()) return QFTAction::evaluate_batch(phi);
    DeviceVector out(phi->batch());
    check(mlmcpi_gff_level_evaluate(level, phi->device(), phi->batch(), (double *)out.ptr(), nullptr), "gff_level_evaluate");
    return out.download<double>();
  }
  void force(const std::shared_ptr<SampleState> phi, std::shared_ptr<SampleState> p) const override {
    if (!plain()) fatal("force of the GFF action is only built for the plain (unrotated, unsmoothed) level");
    QFTAction::force(phi, p);
  }
  void sweep(std::shared_ptr<SampleState> phi, std::shared_ptr<SampleState> scratch, unsigned n_or, unsigned n_hb, uint32_t sweep0) override {
    if (!plain()) fatal("heat bath / overrelaxation sweeps of the GFF action are only built for the plain level");
    QFTAction::sweep(phi, scratch, n_or, n_hb, sweep0);
  }
  int sweep_from(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb, uint32_t sweep0) override {
    if (!plain()) fatal("heat bath / overrelaxation sweeps of the GFF action are only built for the plain level");
    return QFTAction::sweep_from(d_src, d_w0, d_w1, batch, n_or, n_hb, sweep0);
  }
  int sweep_from_qoi(const double *d_src, double *d_w0, double *d_w1, unsigned batch, unsigned n_or, unsigned n_hb, uint32_t sweep0,
                     int qoi_kind, double *d_q, double *d_acc = nullptr) override {
    if (!plain()) fatal("heat bath / overrelaxation sweeps of the GFF action are only built for the plain level");
    return QFTAction::sweep_from_qoi(d_src, d_w0, d_w1, batch, n_or, n_hb, sweep0, qoi_kind, d_q, d_acc);
  }
  /** GFFAction::draw (gffaction.cc:200-213): exact draw + n_gibbs_smooth Gibbs sweeps; `step` numbers the draws */
  void draw_level(std::shared_ptr<SampleState> phi, uint32_t step) const {
    check(mlmcpi_gff_level_draw(level, phi->device_overwrite(), phi->batch(), seed ^ 0x45584143ull, chain0, step, nullptr), "gff_level_draw");
  }
  /** gffaction.cc:121-123: initialise_state = draw */
  void initialise_state(std::shared_ptr<SampleState> phi) const override {
    if (plain()) QFTAction::initialise_state(phi); else draw_level(phi, 0xFFFFFFu);
  }
  /** gffaction.cc:97-118 */
  void copy_from_coarse(const std::shared_ptr<SampleState> phi_coarse, std::shared_ptr<SampleState> phi_state) override {
    check(mlmcpi_gff_copy_from_coarse(level, phi_coarse->device(), phi_state->device_mutable(), phi_state->batch(), nullptr), "gff_copy_from_coarse");
  }
  void copy_from_fine(const std::shared_ptr<SampleState> phi_fine, std::shared_ptr<SampleState> phi_state) override {
    if (!fine_lattice) fatal("cannot copy from fine lattice.");
    if (!fine_level)  // the finer lattice's vertex lists (tables only)
      check(mlmcpi_gff_level_create(fine_lattice->getMt_lat(), fine_lattice->getMx_lat(), (int32_t)fine_lattice->get_coarsening_type(),
                                    fine_lattice->get_coarsening_level(), mass, 0, 1.0, &fine_level), "gff_level_create");
    check(mlmcpi_gff_copy_from_fine(fine_level, phi_fine->device(), phi_state->device_mutable(), phi_state->batch(), nullptr), "gff_copy_from_fine");
  }
  std::string info_string() const override { return QFTAction::info_string() + ", mu2 = " + std::to_string(mu2); }

private:
  const double mass;
  const int n_gibbs_smooth;
  const double omega;
  double mu2;
  mlmcpi_gff_level *level = nullptr;
  mutable mlmcpi_gff_level *fine_level = nullptr;
};

/** action/qft/quenchedschwingeraction.hh:100-277 */
class QuenchedSchwingerAction : public QFTAction {
public:
  QuenchedSchwingerAction(const std::shared_ptr<Lattice2D> lattice_, const std::shared_ptr<Lattice2D> fine_lattice_,
                          const RenormalisationType r, const double beta_)
      : QFTAction(lattice_, fine_lattice_, r, MLMCPI_SCHWINGER, beta_, 0.0), beta(beta_) {
    // quenchedschwingeraction.hh:117-130: links exist only on unrotated lattices
    if (lattice->get_coarsening_type() == CoarsenRotate) fatal("quenched Schwinger action cannot be used with rotated coarsening");
  }
  unsigned int sample_size() const override { return lattice->getNedges(); }
  double getbeta() const { return beta; }
  /** quenchedschwingeraction.hh:147-165 with RenormalisedQuenchedSchwingerParameters::beta_coarse
   *  (quenchedschwingerrenormalisation.hh:52-83): beta/4 when both directions are coarsened, beta/2
   *  otherwise; the perturbative correction and the non-perturbative matching of the topological susceptibility
   *  (quenchedschwingerrenormalisation.cc:7-64; host code behind mlmcpi_schwinger_beta_coarse_nonperturbative) apply for
   *  beta > 4. */
  std::shared_ptr<Action> coarse_action() override {
    std::shared_ptr<Lattice2D> coarse_lattice = lattice->get_coarse_lattice();
    if (!coarse_lattice)
      fatal("cannot coarsen 2d lattice with M_{t,lat} = " + std::to_string(lattice->getMt_lat()) + " , M_{x,lat} = " + std::to_string(lattice->getMx_lat()) + ".");
    const bool both = lattice->get_coarsening_type() == CoarsenBoth;
    double beta_c = (both ? 0.25 : 0.5) * beta;
    if (renormalisation == RenormalisationPerturbative && beta > 4.0) beta_c = (both ? 0.25 : 0.5) * (1. + (both ? 1.5 : 0.5) / beta) * beta;
    if (renormalisation == RenormalisationNonperturbative && beta > 4.0)
      check(mlmcpi_schwinger_beta_coarse_nonperturbative(beta, lattice->getNcells(), both ? 4 : 2, &beta_c), "schwinger_beta_coarse_nonperturbative");
    return std::make_shared<QuenchedSchwingerAction>(coarse_lattice, lattice, renormalisation, beta_c);
  }
  std::string info_string() const override { return QFTAction::info_string() + ", beta = " + std::to_string(beta); }

private:
  const double beta;
};

}  // namespace mlmcpi
#endif
