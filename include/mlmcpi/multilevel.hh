// multilevel.hh -- the multilevel glue of the reference (1-D actions; Schwinger lattice with semi-coarsening), on
// device chains:
//   ConditionedFineAction            action/conditionedfineaction.hh:38-67
//   GaussianConditionedFineAction    action/qm/gaussianconditionedfineaction.{hh,cc}
//   RotorConditionedFineAction       action/qm/rotorconditionedfineaction.{hh,cc}
//   QuenchedSchwingerSemiConditionedFineAction (+ factory)  action/qft/quenchedschwingerconditionedfineaction.{hh,cc}
//   TwoLevelMetropolisStep           montecarlo/twolevelmetropolisstep.{hh,cc}
//   HierarchicalSampler              sampler/hierarchicalsampler.{hh,cc}
//   MonteCarloTwoLevel               montecarlo/montecarlotwolevel.{hh,cc}
//   MonteCarloMultiLevel             montecarlo/montecarlomultilevel.{hh,cc}
// The two-level step (copy_from_coarse, Gaussian fill-in, three action differences, Metropolis test,
// copy of accepted states) is one ABI call, mlmcpi_path_twolevel_draw.
#ifndef MLMCPI_MULTILEVEL_HH
#define MLMCPI_MULTILEVEL_HH
#include <chrono>
#include <cmath>

#include "montecarlo.hh"

namespace mlmcpi {

class ConditionedFineAction {
public:
  virtual ~ConditionedFineAction() {}
  /** the fine-level action whose fine-only points this object fills in */
  virtual std::shared_ptr<Action> fine_action() const = 0;
  /** selector of mlmcpi_lattice_twolevel_draw_cfa: 0 = the class the reference's factory picks for the lattice */
  virtual int cfa_kind() const { return 0; }
};

/** Gaussian fill-in x_{2j+1} ~ N(Wminimum(x_2j, x_2j+2), 1/Wcurvature): harmonic and quartic oscillator.
 *  fill_fine_points / evaluate run inside mlmcpi_path_twolevel_draw. */
class GaussianConditionedFineAction : public ConditionedFineAction {
public:
  explicit GaussianConditionedFineAction(const std::shared_ptr<QMAction> action_) : action(action_) {
    if (action->abi_action().kind == MLMCPI_ROTOR) fatal("Gaussian conditioned fine action not defined for the rotor action");
  }
  std::shared_ptr<Action> fine_action() const override { return action; }

private:
  const std::shared_ptr<QMAction> action;
};

/** action/qm/rotorconditionedfineaction.{hh,cc}: fill-in x_{2j+1} = mod_2pi(Wminimum + ExpSin2(2 Wcurvature)).
 *  fill_fine_points / evaluate run inside mlmcpi_path_twolevel_draw. */
class RotorConditionedFineAction : public ConditionedFineAction {
public:
  explicit RotorConditionedFineAction(const std::shared_ptr<RotorAction> action_) : action(action_) {}
  std::shared_ptr<Action> fine_action() const override { return action; }

private:
  const std::shared_ptr<RotorAction> action;
};

class ConditionedFineActionFactory {
public:
  virtual ~ConditionedFineActionFactory() {}
  virtual std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) = 0;
};

class GaussianConditionedFineActionFactory : public ConditionedFineActionFactory {
public:
  std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) override {
    auto qm = std::dynamic_pointer_cast<QMAction>(action);
    if (!qm) fatal("Gaussian conditioned fine action needs a 1-D action");
    return std::make_shared<GaussianConditionedFineAction>(qm);
  }
};

class RotorConditionedFineActionFactory : public ConditionedFineActionFactory {
public:
  std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) override {
    auto rotor = std::dynamic_pointer_cast<RotorAction>(action);
    if (!rotor) fatal("rotor conditioned fine action needs a RotorAction");
    return std::make_shared<RotorConditionedFineAction>(rotor);
  }
};

/** action/qft/quenchedschwingerconditionedfineaction.hh:134-170 (QuenchedSchwingerSemiConditionedFineAction):
 *  semi-coarsened lattices; uniform shifts of the split links and an ExpCos draw of the links in between run
 *  inside mlmcpi_lattice_twolevel_draw. */
class QuenchedSchwingerSemiConditionedFineAction : public ConditionedFineAction {
public:
  explicit QuenchedSchwingerSemiConditionedFineAction(const std::shared_ptr<QuenchedSchwingerAction> action_) : action(action_) {}
  std::shared_ptr<Action> fine_action() const override { return action; }

private:
  const std::shared_ptr<QuenchedSchwingerAction> action;
};

/** quenchedschwingerconditionedfineaction.hh:44-96 (QuenchedSchwingerConditionedFineAction): lattices coarsened in
 *  both directions; Bessel-product fill-in (beta <= 8) or its approximation, inside mlmcpi_lattice_twolevel_draw. */
class QuenchedSchwingerConditionedFineAction : public ConditionedFineAction {
public:
  explicit QuenchedSchwingerConditionedFineAction(const std::shared_ptr<QuenchedSchwingerAction> action_) : action(action_) {}
  std::shared_ptr<Action> fine_action() const override { return action; }

private:
  const std::shared_ptr<QuenchedSchwingerAction> action;
};

/** quenchedschwingerconditionedfineaction.hh:98-131 (QuenchedSchwingerGaussianConditionedFineAction): the Gaussian variant
 *  of the fill-in for lattices coarsened in both directions (GaussianFillinDistribution); the reference's factory never
 *  selects it, a caller can (QuenchedSchwingerGaussianConditionedFineActionFactory).  Runs inside
 *  mlmcpi_lattice_twolevel_draw_cfa(cfa_kind = 1). */
class QuenchedSchwingerGaussianConditionedFineAction : public ConditionedFineAction {
public:
  explicit QuenchedSchwingerGaussianConditionedFineAction(const std::shared_ptr<QuenchedSchwingerAction> action_) : action(action_) {
    if (action->get_lattice()->get_coarsening_type() != CoarsenBoth) fatal("Gaussian conditioned fine action needs CoarsenBoth");
  }
  std::shared_ptr<Action> fine_action() const override { return action; }
  int cfa_kind() const override { return 1; }

private:
  const std::shared_ptr<QuenchedSchwingerAction> action;
};

class QuenchedSchwingerGaussianConditionedFineActionFactory : public ConditionedFineActionFactory {
public:
  std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) override {
    auto schwinger = std::dynamic_pointer_cast<QuenchedSchwingerAction>(action);
    if (!schwinger) fatal("Schwinger conditioned fine action needs a QuenchedSchwingerAction");
    return std::make_shared<QuenchedSchwingerGaussianConditionedFineAction>(schwinger);
  }
};

/** quenchedschwingerconditionedfineaction.hh:218-238: the fill-in follows the coarsening type of the lattice */
class QuenchedSchwingerConditionedFineActionFactory : public ConditionedFineActionFactory {
public:
  std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) override {
    auto schwinger = std::dynamic_pointer_cast<QuenchedSchwingerAction>(action);
    if (!schwinger) fatal("Schwinger conditioned fine action needs a QuenchedSchwingerAction");
    if (schwinger->get_lattice()->get_coarsening_type() == CoarsenBoth)
      return std::make_shared<QuenchedSchwingerConditionedFineAction>(schwinger);
    return std::make_shared<QuenchedSchwingerSemiConditionedFineAction>(schwinger);
  }
};

/** action/qft/gffconditionedfineaction.{hh,cc}: a fine-only vertex is Gaussian around the mean of its four nearest
 *  neighbours, all of them coarse vertices (red-black coarsening, CoarsenRotate); fill_fine_points / evaluate run inside
 *  mlmcpi_gff_twolevel_draw, and are exposed here for direct use. */
class GFFConditionedFineAction : public ConditionedFineAction {
public:
  explicit GFFConditionedFineAction(const std::shared_ptr<GFFAction> action_) : action(action_) {}
  std::shared_ptr<Action> fine_action() const override { return action; }
  void fill_fine_points(std::shared_ptr<SampleState> phi_state, uint32_t step = 0) const {
    DeviceVector S(phi_state->batch());
    check(mlmcpi_gff_cfa_fill(action->level_handle(), phi_state->device_mutable(), phi_state->batch(), action->get_seed() ^ 0x115147ull,
                              action->get_chain0(), step, (double *)S.ptr(), nullptr), "gff_cfa_fill");
  }
  double evaluate(const std::shared_ptr<SampleState> phi_state) const {
    DeviceVector S(phi_state->batch());
    check(mlmcpi_gff_cfa_evaluate(action->level_handle(), phi_state->device(), phi_state->batch(), (double *)S.ptr(), nullptr), "gff_cfa_evaluate");
    return S.download<double>()[0];
  }

private:
  const std::shared_ptr<GFFAction> action;
};

class GFFConditionedFineActionFactory : public ConditionedFineActionFactory {
public:
  std::shared_ptr<ConditionedFineAction> get(std::shared_ptr<Action> action) override {
    auto gff = std::dynamic_pointer_cast<GFFAction>(action);
    if (!gff) fatal("GFF conditioned fine action needs a GFFAction");
    return std::make_shared<GFFConditionedFineAction>(gff);
  }
};

/** twolevelmetropolisstep.{hh,cc}: draws a fine-level sample from a coarse-level proposal. */
class TwoLevelMetropolisStep : public MCMCStep {
public:
  TwoLevelMetropolisStep(const std::shared_ptr<Action> coarse_action_, const std::shared_ptr<Action> fine_action_,
                         const std::shared_ptr<ConditionedFineAction> conditioned_fine_action_, unsigned int batch = 1,
                         unsigned int n_meas = 200)
      : MCMCStep(), coarse(coarse_action_), fine(fine_action_), qm_coarse(std::dynamic_pointer_cast<QMAction>(coarse_action_)),
        qm_fine(std::dynamic_pointer_cast<QMAction>(fine_action_)), qft_coarse(std::dynamic_pointer_cast<QFTAction>(coarse_action_)),
        qft_fine(std::dynamic_pointer_cast<QFTAction>(fine_action_)), gff_coarse(std::dynamic_pointer_cast<GFFAction>(coarse_action_)),
        gff_fine(std::dynamic_pointer_cast<GFFAction>(fine_action_)), cfa(conditioned_fine_action_), B(batch),
        accept_flags(batch, sizeof(int32_t)), cost_per_sample_(0.0) {
    if (!cfa || !((qm_coarse && qm_fine) || (qft_coarse && qft_fine))) fatal("TwoLevelMetropolisStep: actions have no device implementation");
    size_t bytes = 0;
    if (gff_fine && gff_coarse) check(mlmcpi_gff_twolevel_workspace_bytes(gff_fine->level_handle(), B, &bytes), "gff_twolevel_workspace_bytes");
    else if (qm_fine) check(mlmcpi_path_twolevel_workspace_bytes(&qm_fine->abi_action(), B, &bytes), "twolevel_workspace_bytes");
    else check(mlmcpi_lattice_twolevel_workspace_bytes(&qft_fine->abi_action(), &qft_coarse->abi_action(), B, &bytes), "twolevel_workspace_bytes");
    check(mlmcpi_malloc(&work, bytes), "mlmcpi_malloc");
    theta_fine = std::make_shared<SampleState>(fine->sample_size(), B);
    // twolevelmetropolisstep.cc:23-31: cost per sample by timing draws (10 000 in the reference)
    auto phi_fine = std::make_shared<SampleState>(fine->sample_size(), B);
    auto phi_coarse = std::make_shared<SampleState>(coarse->sample_size(), B);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int k = 0; k < n_meas; ++k) draw(phi_coarse, phi_fine);
    check(mlmcpi_stream_synchronize(nullptr), "synchronize");
    cost_per_sample_ = 1.E6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (n_meas ? n_meas : 1);
    theta_fine->data = std::make_shared<SampleState>(fine->sample_size(), B)->data;  // back to zeros
    reset_stats();
  }
  ~TwoLevelMetropolisStep() { mlmcpi_free(work); }

  /** twolevelmetropolisstep.cc:35-89 */
  void draw(const std::shared_ptr<SampleState> phi_coarse_state, std::shared_ptr<SampleState> phi_state) {
    if (gff_fine && gff_coarse)
      check(mlmcpi_gff_twolevel_draw(gff_fine->level_handle(), gff_coarse->level_handle(), phi_coarse_state->device(),
                                     theta_fine->device_mutable(), B, level_seed(), fine->get_chain0(), step++, work,
                                     (int32_t *)accept_flags.ptr(), nullptr, nullptr), "gff_twolevel_draw");
    else if (qm_fine)
      check(mlmcpi_path_twolevel_draw(&qm_fine->abi_action(), &qm_coarse->abi_action(), phi_coarse_state->device(),
                                      theta_fine->device_mutable(), B, level_seed(), fine->get_chain0(), step++, work,
                                      (int32_t *)accept_flags.ptr(), nullptr, nullptr), "path_twolevel_draw");
    else
      check(mlmcpi_lattice_twolevel_draw_cfa(&qft_fine->abi_action(), &qft_coarse->abi_action(), cfa->cfa_kind(), phi_coarse_state->device(),
                                             theta_fine->device_mutable(), B, level_seed(), fine->get_chain0(), step++, work,
                                             (int32_t *)accept_flags.ptr(), nullptr, nullptr), "lattice_twolevel_draw");
    std::vector<int32_t> flags = accept_flags.download<int32_t>();
    double acc = 0;
    for (int32_t f : flags) acc += f;
    accept = flags[0] != 0;
    n_total_samples++;
    n_accepted_samples += acc / B;
    if (copy_if_rejected || accept || B > 1) phi_state->data = theta_fine->data;
  }
  void set_state(std::shared_ptr<SampleState> phi_state) override { theta_fine->data = phi_state->data; }
  double cost_per_sample() override { return cost_per_sample_; }

private:
  /** every level of a hierarchy gets its own Philox key: the fill-in streams of two levels share
   *  (site, chain, step) counters */
  uint64_t level_seed() const {
    return (fine->get_seed() ^ 0x5517A4B3ull) + 0x9E3779B97F4A7C15ull * (uint64_t)(fine->get_coarsening_level() + 1);
  }
  const std::shared_ptr<Action> coarse, fine;
  const std::shared_ptr<QMAction> qm_coarse, qm_fine;
  const std::shared_ptr<QFTAction> qft_coarse, qft_fine;
  const std::shared_ptr<GFFAction> gff_coarse, gff_fine;
  const std::shared_ptr<ConditionedFineAction> cfa;
  const unsigned int B;
  std::shared_ptr<SampleState> theta_fine;
  void *work = nullptr;
  DeviceVector accept_flags;
  uint32_t step = 0;
  double cost_per_sample_;
};

/** sampler/hierarchicalsampler.hh: parameters */
struct HierarchicalParameters {
  unsigned int n_max_level = 2;
  unsigned int batch = 1;
  unsigned int n_meas = 200;  // draws timed for cost_per_sample (10 000 in the reference)
};

/** hierarchicalsampler.cc:8-118: delayed acceptance through the level hierarchy; the coarsest level is
 *  sampled by the sampler the factory provides (HMC in the reference's drivers). */
class HierarchicalSampler : public Sampler {
public:
  HierarchicalSampler(const std::shared_ptr<Action> fine_action, const std::shared_ptr<SamplerFactory> coarse_sampler_factory,
                      const std::shared_ptr<ConditionedFineActionFactory> cfa_factory, const HierarchicalParameters p)
      : Sampler(), n_level(p.n_max_level - fine_action->get_coarsening_level()), cost_per_sample_(0.0) {
    if (n_level < 1) fatal("hierarchical sampler needs at least one level");
    action.push_back(fine_action);
    for (unsigned int ell = 0; ell + 1 < n_level; ++ell) {
      std::shared_ptr<Action> c = action[ell]->coarse_action();
      c->set_seed(fine_action->get_seed(), fine_action->get_chain0());
      action.push_back(c);
      twolevel_step.push_back(std::make_shared<TwoLevelMetropolisStep>(c, action[ell], cfa_factory->get(action[ell]), p.batch, p.n_meas));
    }
    for (unsigned int ell = 0; ell < n_level; ++ell)
      phi_sampler_state.push_back(std::make_shared<SampleState>(action[ell]->sample_size(), p.batch));
    coarse_sampler = coarse_sampler_factory->get(action[n_level - 1]);
    auto meas_state = std::make_shared<SampleState>(fine_action->sample_size(), p.batch);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int k = 0; k < p.n_meas; ++k) draw(meas_state);
    check(mlmcpi_stream_synchronize(nullptr), "synchronize");
    cost_per_sample_ = 1.E6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (p.n_meas ? p.n_meas : 1);
  }

  /** hierarchicalsampler.cc:55-81 */
  void draw(std::shared_ptr<SampleState> phi_state) override {
    accept = true;
    for (unsigned int ell = 1; ell < n_level; ++ell) action[ell]->copy_from_fine(phi_sampler_state[ell - 1], phi_sampler_state[ell]);
    for (int ell = (int)n_level - 1; ell >= 0; --ell) {
      if (ell == (int)n_level - 1) {
        coarse_sampler->set_state(phi_sampler_state[ell]);
        coarse_sampler->draw(phi_sampler_state[ell]);
        accept = accept && coarse_sampler->accepted();
      } else {
        twolevel_step[ell]->set_state(phi_sampler_state[ell]);
        twolevel_step[ell]->draw(phi_sampler_state[ell + 1], phi_sampler_state[ell]);
        accept = accept && twolevel_step[ell]->accepted();
      }
      if (!accept) break;
    }
    n_total_samples++;
    n_accepted_samples += accept ? 1 : 0;
    if (accept || copy_if_rejected) phi_state->data = phi_sampler_state[0]->data;
  }
  void set_state(std::shared_ptr<SampleState> phi_state) override { phi_sampler_state[0]->data = phi_state->data; }
  double cost_per_sample() override { return cost_per_sample_; }
  void show_stats() override {
    std::cout << std::setprecision(4) << std::fixed << "  cost per sample = " << cost_per_sample() << " mu s" << std::endl;
    std::cout << "  acceptance rate = " << p_accept() << std::endl;
    for (unsigned int ell = 0; ell < n_level; ++ell)
      std::cout << "  level " << ell << " : p = " << (ell == n_level - 1 ? coarse_sampler->p_accept() : twolevel_step[ell]->p_accept()) << std::endl;
  }

private:
  const unsigned int n_level;
  std::vector<std::shared_ptr<Action>> action;
  std::vector<std::shared_ptr<TwoLevelMetropolisStep>> twolevel_step;
  std::vector<std::shared_ptr<SampleState>> phi_sampler_state;
  std::shared_ptr<Sampler> coarse_sampler;
  double cost_per_sample_;
};

class HierarchicalSamplerFactory : public SamplerFactory {
public:
  HierarchicalSamplerFactory(std::shared_ptr<SamplerFactory> coarse, std::shared_ptr<ConditionedFineActionFactory> cfa, HierarchicalParameters p)
      : coarse_factory(coarse), cfa_factory(cfa), param(p) {}
  std::shared_ptr<Sampler> get(std::shared_ptr<Action> action) override {
    return std::make_shared<HierarchicalSampler>(action, coarse_factory, cfa_factory, param);
  }
private:
  std::shared_ptr<SamplerFactory> coarse_factory;
  std::shared_ptr<ConditionedFineActionFactory> cfa_factory;
  const HierarchicalParameters param;
};

/** sampler/multilevelsampler.{hh,cc}: independent samples are passed up the hierarchy -- a level only hands
 *  its state to the next finer two-level step once ceil(tau_int) draws have been made on it since the last
 *  hand-over (multilevelsampler.cc:71-113). */
class MultilevelSampler : public Sampler {
public:
  MultilevelSampler(const std::shared_ptr<Action> fine_action, const std::shared_ptr<QoIFactory> qoi_factory,
                    const std::shared_ptr<SamplerFactory> coarse_sampler_factory,
                    const std::shared_ptr<ConditionedFineActionFactory> cfa_factory, unsigned int n_autocorr_window,
                    const HierarchicalParameters p)
      : Sampler(), n_level(p.n_max_level - fine_action->get_coarsening_level()), t_indep(n_level, 0.0),
        n_indep(n_level, 0), t_sampler(n_level, 0), cost_per_sample_(0.0) {
    if (n_level < 1) fatal("multilevel sampler needs at least one level");
    action.push_back(fine_action);
    for (unsigned int ell = 0; ell + 1 < n_level; ++ell) {
      std::shared_ptr<Action> c = action[ell]->coarse_action();
      c->set_seed(fine_action->get_seed() + 104729 * (ell + 1), fine_action->get_chain0());
      action.push_back(c);
      twolevel_step.push_back(std::make_shared<TwoLevelMetropolisStep>(c, action[ell], cfa_factory->get(action[ell]), 1, p.n_meas));
    }
    for (unsigned int ell = 0; ell < n_level; ++ell) {
      qoi.push_back(qoi_factory->get(action[ell]));
      phi_sampler_state.push_back(std::make_shared<SampleState>(action[ell]->sample_size()));
      stats_sampler.push_back(std::make_shared<Statistics>("   Q_{sampler}[" + std::to_string(ell) + "]", n_autocorr_window));
    }
    coarse_sampler = coarse_sampler_factory->get(action[n_level - 1]);
    auto meas_state = std::make_shared<SampleState>(fine_action->sample_size());
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int k = 0; k < p.n_meas; ++k) draw(meas_state);
    check(mlmcpi_stream_synchronize(nullptr), "synchronize");
    cost_per_sample_ = 1.E6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (p.n_meas ? p.n_meas : 1);
  }

  void draw(std::shared_ptr<SampleState> phi_state) override {
    accept = true;
    int level = (int)n_level - 1;
    do {
      if (level == (int)n_level - 1) coarse_sampler->draw(phi_sampler_state[level]);
      else twolevel_step[level]->draw(phi_sampler_state[level + 1], phi_sampler_state[level]);
      stats_sampler[level]->record_sample(qoi[level]->evaluate(phi_sampler_state[level]));
      t_sampler[level]++;
      double tau = std::ceil(stats_sampler[level]->tau_int());
      if (!(tau <= 1. + 2. * stats_sampler[level]->autocorr_window())) tau = 1. + 2. * stats_sampler[level]->autocorr_window();
      if (t_sampler[level] >= tau) {
        t_indep[level] = (n_indep[level] * t_indep[level] + t_sampler[level]) / (1.0 + n_indep[level]);
        n_indep[level]++;
        t_sampler[level] = 0;
        level--;  // a new independent sample: hand it to the next finer level
      } else {
        level = (int)n_level - 1;
      }
    } while (level >= 0);
    n_total_samples++;
    n_accepted_samples++;
    phi_state->data = phi_sampler_state[0]->data;
  }
  void set_state(std::shared_ptr<SampleState> phi_state) override { phi_sampler_state[0]->data = phi_state->data; }
  double cost_per_sample() override { return cost_per_sample_; }
  void show_stats() override {
    std::cout << std::setprecision(3) << std::fixed << "  cost per sample = " << cost_per_sample() << " mu s" << std::endl;
    for (unsigned int ell = 0; ell < n_level; ++ell)
      std::cout << " level " << ell << " : average spacing between samples " << t_indep[ell] << std::endl << *stats_sampler[ell];
  }

private:
  const unsigned int n_level;
  std::vector<std::shared_ptr<Action>> action;
  std::vector<std::shared_ptr<TwoLevelMetropolisStep>> twolevel_step;
  std::vector<std::shared_ptr<QoI>> qoi;
  std::vector<std::shared_ptr<SampleState>> phi_sampler_state;
  std::vector<std::shared_ptr<Statistics>> stats_sampler;
  std::shared_ptr<Sampler> coarse_sampler;
  std::vector<double> t_indep;
  std::vector<unsigned int> n_indep, t_sampler;
  double cost_per_sample_;
};

/** montecarlo/montecarlotwolevel.hh: parameters */
struct TwoLevelMCParameters {
  unsigned int n_burnin = 100, n_samples = 100;
  unsigned int n_coarse_autocorr_window = 10, n_fine_autocorr_window = 10, n_delta_autocorr_window = 10;
  unsigned int n_autocorr_window = 20;  // statistics of the coarse sampler (StatisticsParameters)
  unsigned int n_meas = 200;            // draws timed for the two-level step's cost_per_sample
};

/** montecarlo/montecarlotwolevel.{hh,cc}: mean and variance of Q_fine, Q_coarse and of their difference over the
 *  pairs (phi_coarse, phi_fine) a two-level step produces from (nearly) independent coarse samples. */
class MonteCarloTwoLevel {
public:
  MonteCarloTwoLevel(const std::shared_ptr<Action> fine_action_, const std::shared_ptr<QoIFactory> qoi_factory,
                     const std::shared_ptr<SamplerFactory> sampler_factory, const std::shared_ptr<ConditionedFineActionFactory> cfa_factory,
                     const TwoLevelMCParameters p)
      : param(p), fine_action(fine_action_), coarse_action(fine_action_->coarse_action()), qoi_fine(qoi_factory->get(fine_action_)),
        qoi_coarse(qoi_factory->get(coarse_action)), stats_fine("QoI[fine]", p.n_fine_autocorr_window),
        stats_coarse("QoI[coarse]", p.n_coarse_autocorr_window), stats_diff("delta QoI", p.n_delta_autocorr_window),
        stats_coarse_sampler("QoI[coarsesampler]", p.n_autocorr_window) {
    coarse_action->set_seed(fine_action->get_seed() + 7919, fine_action->get_chain0());
    coarse_sampler = sampler_factory->get(coarse_action);
    twolevel_step = std::make_shared<TwoLevelMetropolisStep>(coarse_action, fine_action, cfa_factory->get(fine_action), 1, p.n_meas);
  }

  /** montecarlotwolevel.cc:39-84 */
  void evaluate_difference() {
    auto phi_state = std::make_shared<SampleState>(fine_action->sample_size());
    auto phi_coarse_state = std::make_shared<SampleState>(coarse_action->sample_size());
    stats_coarse.hard_reset(); stats_coarse_sampler.hard_reset(); stats_fine.hard_reset(); stats_diff.hard_reset();
    for (unsigned int k = 0; k < param.n_burnin; ++k) record_pair(phi_coarse_state, phi_state);
    stats_coarse_sampler.reset();
    // a hard reset: the variances are what this estimator is after
    stats_coarse.hard_reset(); stats_fine.hard_reset(); stats_diff.hard_reset();
    for (unsigned int k = 0; k < param.n_samples; ++k) record_pair(phi_coarse_state, phi_state);
  }
  void show_statistics() const {
    std::cout << stats_fine << std::endl << stats_coarse << std::endl << stats_diff << std::endl << std::endl;
    std::cout << "=== Coarse level sampler statistics === " << std::endl << stats_coarse_sampler << std::endl;
    coarse_sampler->show_stats();
    std::cout << std::endl << "=== Two level sampler statistics === " << std::endl;
    twolevel_step->show_stats();
  }
  const Statistics &fine_statistics() const { return stats_fine; }
  const Statistics &coarse_statistics() const { return stats_coarse; }
  const Statistics &difference_statistics() const { return stats_diff; }
  std::shared_ptr<TwoLevelMetropolisStep> get_twolevel_step() { return twolevel_step; }

private:
  void record_pair(std::shared_ptr<SampleState> phi_coarse_state, std::shared_ptr<SampleState> phi_state) {
    draw_coarse_sample(phi_coarse_state);
    twolevel_step->draw(phi_coarse_state, phi_state);
    const double qf = qoi_fine->evaluate(phi_state), qc = qoi_coarse->evaluate(phi_coarse_state);
    stats_fine.record_sample(qf);
    stats_coarse.record_sample(qc);
    stats_diff.record_sample(qf - qc);
  }
  /** montecarlotwolevel.cc:87-98: ceil(2 tau_int) <= 100 draws of the coarse sampler between uses */
  void draw_coarse_sample(std::shared_ptr<SampleState> phi_state) {
    double two_tau_int = std::fmin(100., std::ceil(2. * stats_coarse_sampler.tau_int()));
    if (!(two_tau_int >= 1.)) two_tau_int = 1.;  // NaN from a degenerate first window
    while (t_sampler < two_tau_int) {
      coarse_sampler->draw(phi_state);
      stats_coarse_sampler.record_sample(qoi_coarse->evaluate(phi_state));
      t_sampler++;
    }
    t_indep = (n_indep * t_indep + t_sampler) / (1.0 + n_indep);
    n_indep++;
    t_sampler = 0;
  }

  const TwoLevelMCParameters param;
  std::shared_ptr<Action> fine_action, coarse_action;
  std::shared_ptr<Sampler> coarse_sampler;
  std::shared_ptr<QoI> qoi_fine, qoi_coarse;
  std::shared_ptr<TwoLevelMetropolisStep> twolevel_step;
  double t_indep = 0.0;
  int n_indep = 0, t_sampler = 0;
  Statistics stats_fine, stats_coarse, stats_diff, stats_coarse_sampler;
};

/** montecarlo/montecarlomultilevel.hh: parameters */
struct MultiLevelMCParameters {
  unsigned int n_level = 3;
  unsigned int n_burnin = 100;
  double epsilon = 1.0e-2;
  unsigned int n_autocorr_window = 20, n_min_samples_qoi = 100;
  bool sub_sample_coarse = true;  // hierarchical coarse samplers: draw ceil(2 tau_int) times between uses
  unsigned int n_meas = 200;
  // level sharding: this process is rank `level_rank` of `level_ranks`; it owns, builds and samples
  // the levels l with l % level_ranks == level_rank.  1 rank = the reference's single-process estimator.
  unsigned int level_rank = 0, level_ranks = 1;
};

/** What the ranks of a level-sharded estimator exchange once per pass of the do-while of
 *  montecarlomultilevel.cc:115-165: a table [n_level][5] = (samples, mean, variance, tau_int, cost) in which every
 *  rank fills the rows of its own levels and leaves the others zero, summed element-wise over ranks (an all-reduce
 *  of 5 n_level doubles, which for disjoint rows is the all-gather).  Implementations: MPI_Allreduce in the
 *  reference's build, ncclAllReduce / torch.distributed over RCCL (mlmcpathintegral_amd/chains.py), none for 1 rank. */
class LevelExchange {
public:
  virtual ~LevelExchange() {}
  virtual void allreduce_sum(std::vector<double> &table) = 0;
};

/** the level table over any Exchange of exchange.hh (RcclExchange in production) */
class LevelExchangeOver : public LevelExchange {
public:
  explicit LevelExchangeOver(std::shared_ptr<Exchange> e_) : e(e_) {}
  void allreduce_sum(std::vector<double> &table) override { e->allreduce_sum(table.data(), table.size()); }

private:
  std::shared_ptr<Exchange> e;
};

/** montecarlomultilevel.cc:7-282: telescoping-sum estimator Q = sum_l E[Y_l], Y_L = Q_L on the coarsest
 *  level, Y_l = Q_l(x_f) - Q_{l+1}(x_c) from the two-level step; sample numbers from the variance / cost
 *  model of :148-164.  Levels are independent estimators (each owns its samplers and states), which is
 *  what lets them be placed on different GPUs. */
class MonteCarloMultiLevel {
public:
  MonteCarloMultiLevel(std::shared_ptr<Action> fine_action_, std::shared_ptr<QoIFactory> qoi_factory,
                       std::shared_ptr<SamplerFactory> sampler_factory, std::shared_ptr<ConditionedFineActionFactory> cfa_factory,
                       const MultiLevelMCParameters p)
      : param(p), n_level(p.n_level) {
    if (n_level < 2) fatal("multilevel method needs at least two levels");
    if (p.level_ranks < 1 || p.level_rank >= p.level_ranks) fatal("multilevel method: invalid level partition");
    action.push_back(fine_action_);
    for (unsigned int level = 0; level + 1 < n_level; ++level) {
      std::shared_ptr<Action> c = action[level]->coarse_action();
      c->set_seed(fine_action_->get_seed() + 7919 * (level + 1), fine_action_->get_chain0());
      action.push_back(c);
      // level `level` needs its two-level step and the sampler of level + 1; the coarsest level its own sampler
      const bool needed = owns(level) || (level + 2 == n_level && owns(level + 1));
      twolevel_step.push_back(owns(level) ? std::make_shared<TwoLevelMetropolisStep>(c, action[level], cfa_factory->get(action[level]), 1, p.n_meas) : nullptr);
      coarse_sampler.push_back(needed ? sampler_factory->get(c) : nullptr);  // sampler for level + 1
    }
    for (unsigned int level = 0; level < n_level; ++level) {
      qoi.push_back(qoi_factory->get(action[level]));
      phi_state.push_back(std::make_shared<SampleState>(action[level]->sample_size()));
      phi_coarse_state.push_back(std::make_shared<SampleState>(action[level]->sample_size()));
      stats_qoi.push_back(std::make_shared<Statistics>("Y_" + std::to_string(level), p.n_autocorr_window));
      if (level + 1 < n_level) stats_coarse_sampler.push_back(std::make_shared<Statistics>("Q_sampler_" + std::to_string(level + 1), p.n_autocorr_window));
    }
    t_indep.assign(n_level - 1, 1.0);
    n_indep.assign(n_level - 1, 0.0);
    t_sampler.assign(n_level - 1, 0.0);
    n_target.assign(n_level, p.n_min_samples_qoi);
  }

  bool owns(unsigned int level) const { return level % param.level_ranks == param.level_rank; }
  void set_exchange(std::shared_ptr<LevelExchange> e) { exchange = e; }

  /** montecarlomultilevel.cc:71-167 = begin(); do { table = pass(); exchange; } while (!update(table)); */
  void evaluate() {
    if (param.level_ranks > 1 && !exchange) fatal("level-sharded multilevel estimator needs a LevelExchange");
    begin();
    bool sufficient;
    do {
      std::vector<double> table = pass();
      if (exchange) exchange->allreduce_sum(table);
      sufficient = update(table);
    } while (!sufficient);
  }

  /** burn-in of the owned levels (montecarlomultilevel.cc:83-100): straight from the coarse samplers, no sub-sampling */
  void begin() {
    for (unsigned int level = 0; level < n_level; ++level) stats_qoi[level]->hard_reset();
    for (auto &s : stats_coarse_sampler) s->hard_reset();
    for (int level = (int)n_level - 1; level >= 0; --level) {
      if (!owns(level)) continue;
      for (unsigned int j = 0; j < param.n_burnin; ++j) {
        double qoi_Y;
        if (level == (int)n_level - 1) {
          coarse_sampler[level - 1]->draw(phi_state[level]);
          qoi_Y = qoi[level]->evaluate(phi_state[level]);
        } else {
          coarse_sampler[level]->draw(phi_coarse_state[level + 1]);
          twolevel_step[level]->draw(phi_coarse_state[level + 1], phi_state[level]);
          qoi_Y = qoi[level]->evaluate(phi_state[level]) - qoi[level + 1]->evaluate(phi_coarse_state[level + 1]);
        }
        stats_qoi[level]->record_sample(qoi_Y);
      }
    }
    for (unsigned int level = 0; level < n_level; ++level) {
      stats_qoi[level]->reset();
      n_target[level] = param.n_min_samples_qoi;
    }
    summary.assign(5 * n_level, 0.0);
  }

  /** one pass of montecarlomultilevel.cc:115-147 over the owned levels; returns the table with the owned rows filled */
  std::vector<double> pass() {
    std::vector<double> table(5 * n_level, 0.0);
    for (int level = (int)n_level - 1; level >= 0; --level) {
      if (!owns(level)) continue;
      for (unsigned int j = stats_qoi[level]->samples(); j < n_target[level]; ++j) stats_qoi[level]->record_sample(sample_Y(level));
      if (verbose) std::cout << "  level " << level << ": " << stats_qoi[level]->samples() << " samples" << std::endl;
      double *row = &table[5 * level];
      row[0] = stats_qoi[level]->samples();
      row[1] = stats_qoi[level]->average();
      row[2] = stats_qoi[level]->variance();
      row[3] = stats_qoi[level]->tau_int();
      row[4] = cost_eff(level);
    }
    return table;
  }

  /** sample targets from the complete table (montecarlomultilevel.cc:148-164); true when every level has enough */
  bool update(const std::vector<double> &table) {
    summary = table;
    const double two_epsilon_inv2 = 2. / (param.epsilon * param.epsilon);
    bool sufficient = true;
    double sum = 0;
    for (unsigned int ell = 0; ell < n_level; ++ell) sum += std::sqrt(table[5 * ell + 2] * table[5 * ell + 4]);
    for (unsigned int ell = 0; ell < n_level; ++ell) {
      const double V = table[5 * ell + 2], C = table[5 * ell + 4];
      // a level without variance or cost information yet (V or C zero / NaN) keeps its target instead of casting a
      // non-finite number (undefined behaviour); the reference has the same division and no guard
      const double want = std::ceil(two_epsilon_inv2 * sum * std::sqrt(V / C) * table[5 * ell + 3]);
      if (std::isfinite(want) && want < 4294967295.0) n_target[ell] = (unsigned int)want;
      sufficient = sufficient && (table[5 * ell] >= n_target[ell]);
    }
    return sufficient;
  }
  /** montecarlomultilevel.cc:255-271 */
  double numerical_result() const {
    double q = 0;
    for (unsigned int ell = 0; ell < n_level; ++ell) q += summary[5 * ell + 1];
    return q;
  }
  /** sum of the levels' error^2 = tau_int variance / samples (statistics.cc:62-64) */
  double statistical_error() const {
    double e2 = 0;
    for (unsigned int ell = 0; ell < n_level; ++ell) e2 += summary[5 * ell + 3] * summary[5 * ell + 2] / summary[5 * ell];
    return std::sqrt(e2);
  }
  void show_statistics() {
    for (unsigned int ell = 0; ell < n_level; ++ell)
      if (owns(ell)) std::cout << *stats_qoi[ell] << "  target samples = " << n_target[ell] << std::endl;
    std::cout << " Q = " << std::setprecision(6) << numerical_result() << " +/- " << statistical_error() << std::endl;
  }
  std::shared_ptr<Statistics> level_statistics(unsigned int ell) { return stats_qoi[ell]; }
  bool verbose = false;

private:
  double sample_Y(int level) {
    if (level == (int)n_level - 1) {
      draw_coarse_sample(level, phi_state[level]);
      return qoi[level]->evaluate(phi_state[level]);
    }
    draw_coarse_sample(level + 1, phi_coarse_state[level + 1]);
    twolevel_step[level]->draw(phi_coarse_state[level + 1], phi_state[level]);
    return qoi[level]->evaluate(phi_state[level]) - qoi[level + 1]->evaluate(phi_coarse_state[level + 1]);
  }
  /** montecarlomultilevel.cc:170-190 */
  void draw_coarse_sample(const unsigned int level, std::shared_ptr<SampleState> state) {
    const unsigned int k = level - 1;
    if (param.sub_sample_coarse) {
      // tau_int estimated from a window of n_autocorr_window lags is at most 1 + 2 n_autocorr_window; a
      // larger value (or NaN) only arises from rounding noise when the first few samples coincide
      double tau = std::ceil(2. * stats_coarse_sampler[k]->tau_int());
      const double tau_max = 2. * (1. + 2. * param.n_autocorr_window);
      if (!(tau <= tau_max)) tau = tau_max;
      while (t_sampler[k] < tau) {
        coarse_sampler[k]->draw(state);
        stats_coarse_sampler[k]->record_sample(qoi[level]->evaluate(state));
        t_sampler[k]++;
      }
    } else {
      coarse_sampler[k]->draw(state);
      t_sampler[k] = 1;
    }
    t_indep[k] = (n_indep[k] * t_indep[k] + t_sampler[k]) / (1.0 + n_indep[k]);
    n_indep[k]++;
    t_sampler[k] = 0;
  }
  /** montecarlomultilevel.cc:193-204 */
  double cost_eff(const int ell) const {
    const double cost = ell == (int)n_level - 1
                            ? t_indep[ell - 1] * coarse_sampler[ell - 1]->cost_per_sample()
                            : twolevel_step[ell]->cost_per_sample() + t_indep[ell] * coarse_sampler[ell]->cost_per_sample();
    return std::ceil(stats_qoi[ell]->tau_int()) * cost;
  }

  const MultiLevelMCParameters param;
  const unsigned int n_level;
  std::vector<std::shared_ptr<Action>> action;
  std::vector<std::shared_ptr<TwoLevelMetropolisStep>> twolevel_step;
  std::vector<std::shared_ptr<Sampler>> coarse_sampler;
  std::vector<std::shared_ptr<QoI>> qoi;
  std::vector<std::shared_ptr<SampleState>> phi_state, phi_coarse_state;
  std::vector<std::shared_ptr<Statistics>> stats_qoi, stats_coarse_sampler;
  std::vector<double> t_indep, n_indep, t_sampler;
  std::vector<unsigned int> n_target;
  std::vector<double> summary;  // the last exchanged table
  std::shared_ptr<LevelExchange> exchange;
};

/** QoI factories (qoi/qm/qoixsquared.hh etc.): a QoI per level */
class QoIXsquaredFactory : public QoIFactory {
public:
  std::shared_ptr<QoI> get(std::shared_ptr<Action> action) override {
    auto qm = std::dynamic_pointer_cast<QMAction>(action);
    if (!qm) fatal("QoIXsquared needs a 1-D action");
    return std::make_shared<QoIXsquared>(qm->get_lattice());
  }
};

/** qoi/qm/qoisusceptibility.hh, qoi/qft/{qoi2dsusceptibility,qoiavgplaquette,qoi2dphisquared}.hh factories */
class QoISusceptibilityFactory : public QoIFactory {
public:
  std::shared_ptr<QoI> get(std::shared_ptr<Action> action) override {
    auto qm = std::dynamic_pointer_cast<QMAction>(action);
    if (!qm) fatal("QoISusceptibility needs a 1-D action");
    return std::make_shared<QoISusceptibility>(qm->get_lattice());
  }
};
template <class Q>
class QFTQoIFactory : public QoIFactory {
public:
  std::shared_ptr<QoI> get(std::shared_ptr<Action> action) override {
    auto qft = std::dynamic_pointer_cast<QFTAction>(action);
    if (!qft) fatal("this QoI needs a 2-D action");
    return std::make_shared<Q>(qft->get_lattice());
  }
};
typedef QFTQoIFactory<QoI2DSusceptibility> QoI2DSusceptibilityFactory;
typedef QFTQoIFactory<QoIAvgPlaquette> QoIAvgPlaquetteFactory;
typedef QFTQoIFactory<QoI2DPhiSquared> QoI2DPhiSquaredFactory;

}  // namespace mlmcpi
#endif
