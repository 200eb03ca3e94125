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

  /** action.hh:73-96.  Site-at-a-time updates are the reference's CPU inner loop; on the device
   *  the unit of work is a whole multicolour sweep (OverrelaxedHeatBathSampler::draw).  Calling the
   *  per-site form is an error, exactly like calling it on an action that does not implement it. */
  virtual void heatbath_update(std::shared_ptr<SampleState>, const unsigned int) {
    fatal("heat bath update not implemented for this action (use OverrelaxedHeatBathSampler::draw, which runs device sweeps)");
  }
  virtual void overrelaxation_update(std::shared_ptr<SampleState>, const unsigned int) {
    fatal("overrelaxation update not implemented for this action (use OverrelaxedHeatBathSampler::draw, which runs device sweeps)");
  }
  /** Does OverrelaxedHeatBathSampler work with this action? */
  virtual bool has_local_updates() const { return false; }
  /** Device sweeps: n_overrelax overrelaxation then n_heatbath heat-bath sweeps. */
  virtual void sweep(std::shared_ptr<SampleState>, std::shared_ptr<SampleState>, unsigned, unsigned, uint32_t) {
    fatal("overrelaxation update not implemented for this action");
  }
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

/** action/qm/rotoraction.hh:93-289 */
class RotorAction : public QMAction {
public:
  RotorAction(const std::shared_ptr<Lattice1D> lattice_, const RenormalisationType r, const double m0_)
      : QMAction(lattice_, r, MLMCPI_ROTOR, m0_) {
    seed = 21172817;  // rotoraction.hh:106
  }
  bool has_local_updates() const override { return true; }
  void sweep(std::shared_ptr<SampleState> x, std::shared_ptr<SampleState> scratch, unsigned n_or, unsigned n_hb,
             uint32_t sweep0) override {
    check(mlmcpi_path_sweep_draw(&abi, x->device_mutable(), scratch->device_mutable(), x->batch(), n_or, n_hb, seed,
                                 chain0, sweep0, nullptr), "path_sweep_draw");
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
  void sweep(std::shared_ptr<SampleState> phi, std::shared_ptr<SampleState> scratch, unsigned n_or, unsigned n_hb,
             uint32_t sweep0) override {
    int32_t in_scratch = 0;
    check(mlmcpi_lattice_sweep_draw_pingpong(&abi, phi->device_mutable(), scratch->device_mutable(), phi->batch(), n_or,
                                             n_hb, seed, chain0, sweep0, fuse, &in_scratch, nullptr), "lattice_sweep_draw");
    if (in_scratch) phi->swap_device(*scratch);  // no copy: the buffers exchange roles
  }
  std::string info_string() const override {
    std::stringstream s;
    s << "Mt_lat = " << abi.Mt << ", Mx_lat = " << abi.Mx;
    return s.str();
  }
  unsigned fuse = 0;  // sweeps fused per launch (0 = library default); results do not depend on it

protected:
  const std::shared_ptr<Lattice2D> lattice, fine_lattice;
  mlmcpi_lattice_action abi;
};

/** action/qft/gffaction.hh:152-350 with n_gibbs_smooth = 0.  The dense N x N matrices the reference
 *  builds in its constructor (gffaction.cc:133-174) are not needed by any sweep-path method and are
 *  not built (they make the reference unusable beyond ~64 x 64). */
class GFFAction : public QFTAction {
public:
  GFFAction(const std::shared_ptr<Lattice2D> lattice_, const std::shared_ptr<Lattice2D> fine_lattice_, const double mass_)
      : QFTAction(lattice_, fine_lattice_, RenormalisationNone, MLMCPI_GFF, 0.0, mass_), mass(mass_) {
    if (lattice->getMt_lat() != lattice->getMx_lat()) fatal("Lattice has to be squared for GFF action ");
    if (lattice->is_rotated()) fatal("rotated lattices are not supported by the device GFF action");
    const double a = 1. / lattice->getMt_lat();
    mu2 = a * a * mass * mass;
  }
  unsigned int sample_size() const override { return lattice->getNvertices(); }
  double getmu2() const { return mu2; }
  std::string info_string() const override { return QFTAction::info_string() + ", mu2 = " + std::to_string(mu2); }

private:
  const double mass;
  double mu2;
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
  std::string info_string() const override { return QFTAction::info_string() + ", beta = " + std::to_string(beta); }

private:
  const double beta;
};

}  // namespace mlmcpi
#endif
