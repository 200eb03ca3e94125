// sampler.hh -- MCMCStep (montecarlo/mcmcstep.hh:21-72), Sampler / SamplerFactory
// (sampler/sampler.hh:20-43), HMCSampler (sampler/hmcsampler.{hh,cc}) and
// OverrelaxedHeatBathSampler (sampler/overrelaxedheatbathsampler.{hh,cc}) driving device chains.
#ifndef MLMCPI_SAMPLER_HH
#define MLMCPI_SAMPLER_HH
#include <cmath>
#include <iomanip>
#include <iostream>
#include <algorithm>
#include <memory>
#include <random>
#include <typeinfo>
#include <vector>

#include "action.hh"

namespace mlmcpi {

class MCMCStep {
public:
  MCMCStep() : accept(false), copy_if_rejected(false) { reset_stats(); }
  virtual ~MCMCStep() {}
  void reset_stats() { n_total_samples = 0; n_accepted_samples = 0; }
  /** acceptance probability; with a batch, averaged over its chains */
  double p_accept() { return n_accepted_samples / (1. * n_total_samples); }
  virtual void show_stats() {
    std::cout << std::setprecision(5) << std::fixed;
    std::cout << "  acceptance probability  p = " << p_accept() << std::endl;
    std::cout << "  rejection probability 1-p = " << 1. - p_accept() << std::endl;
  }
  bool accepted() const { return accept; }
  virtual void set_state(std::shared_ptr<SampleState> x_state) = 0;
  virtual double cost_per_sample() { fatal(std::string(" Cost per sample not defined for class ") + typeid(*this).name()); }

protected:
  mutable double n_accepted_samples;  // fractional for batches: mean over chains per draw
  mutable unsigned int n_total_samples;
  mutable bool accept;
  mutable bool copy_if_rejected;
};

class Sampler : public MCMCStep {
public:
  Sampler() : MCMCStep() {}
  virtual ~Sampler() {}
  /** sampler/sampler.hh:36 */
  virtual void draw(std::shared_ptr<SampleState> phi_state) = 0;
};

class SamplerFactory {
public:
  virtual ~SamplerFactory() {}
  virtual std::shared_ptr<Sampler> get(std::shared_ptr<Action> action) = 0;
};

/** sampler/hmcsampler.hh:21-65 */
struct HMCParameters {
  unsigned int nt = 100;
  double dt = 0.1;
  unsigned int n_burnin = 100;
  unsigned int n_rep = 1;
  bool autotune = true;             // the reference always tunes (hmcsampler.hh:107)
  unsigned int batch = 1;           // independent chains advanced together
  unsigned int tune_iterations = 100, tune_samples = 1000;  // hmcsampler.cc:78,89
};

/** HMC on device chains.  One draw = n_rep fused trajectories (mlmcpi_path_hmc_draw for 1-D
 *  actions, mlmcpi_lattice_hmc_draw for 2-D ones).  Construction follows hmcsampler.hh:84-109:
 *  initialise_state, n_burnin draws, step-size auto-tuning, reset of the counters. */
class HMCSampler : public Sampler {
public:
  HMCSampler(const std::shared_ptr<Action> action_, const HMCParameters hmc_param_)
      : Sampler(), action(action_), nt_hmc(hmc_param_.nt), dt_hmc(hmc_param_.dt), n_rep(hmc_param_.n_rep),
        n_burnin(hmc_param_.n_burnin), B(hmc_param_.batch), accept_flags(hmc_param_.batch, sizeof(int32_t)),
        energies(4 * (size_t)hmc_param_.batch) {
    qm = dynamic_cast<QMAction *>(action.get());
    qft = dynamic_cast<QFTAction *>(action.get());
    if (!qm && !qft) fatal("HMCSampler: action has no device implementation");
    size_t bytes = 0;
    if (qm) check(mlmcpi_path_hmc_workspace_bytes(&qm->abi_action(), B, nt_hmc, &bytes), "hmc_workspace_bytes");
    else check(mlmcpi_lattice_hmc_workspace_bytes(&qft->abi_action(), B, &bytes), "hmc_workspace_bytes");
    check(mlmcpi_malloc(&work, bytes), "mlmcpi_malloc");
    phi_state_cur = std::make_shared<SampleState>(action->sample_size(), B);
    action->initialise_state(phi_state_cur);
    std::shared_ptr<SampleState> tmp = std::make_shared<SampleState>(action->sample_size(), B);
    for (unsigned int i = 0; i < n_burnin; ++i) draw(tmp);
    if (hmc_param_.autotune) autotune_stepsize(0.8, hmc_param_.tune_iterations, hmc_param_.tune_samples);
    reset_stats();
  }
  virtual ~HMCSampler() { mlmcpi_free(work); }

  /** hmcsampler.cc:8-19.  With a batch, `accepted()` reports chain 0 and rejected chains keep their
   *  previous content in phi_state (copy_if_rejected == false), like the reference. */
  void draw(std::shared_ptr<SampleState> phi_state) override {
    const double mean_acc = step();
    n_total_samples++;
    n_accepted_samples += mean_acc;
    // hmcsampler.cc:16-18: copy out only if accepted.  For a batch the whole current state is copied:
    // a rejected chain's current state equals what the caller received at its last acceptance.
    if (copy_if_rejected || accept || B > 1) phi_state->data = phi_state_cur->data;
  }
  void set_state(std::shared_ptr<SampleState> phi_state) override { phi_state_cur->data = phi_state->data; }
  double get_dt() const { return dt_hmc; }
  std::shared_ptr<SampleState> current_state() { return phi_state_cur; }

  /** hmcsampler.cc:77-113: bisection on [dt/2, 2dt], fixed iteration count, result kept only if some
   *  iterate came within 0.01 of the target.  Each iterate averages ~tune_samples trajectories. */
  void autotune_stepsize(const double p_accept_target, unsigned iterations = 100, unsigned n_autotune_samples = 1000) {
    const double tolerance = 1.E-2, dt_original = dt_hmc;
    double dt_min = 0.5 * dt_hmc, dt_max = 2. * dt_hmc;
    bool converged = false;
    std::cout << std::setprecision(4) << std::fixed;
    std::cout << " Auto-tuning HMC step size to achieve acceptance rate of " << p_accept_target << " ..." << std::endl;
    std::cout << "  Starting with dt_{HMC} = " << dt_hmc << std::endl;
    const unsigned draws = (n_autotune_samples + B - 1) / B;
    for (unsigned int k = 0; k < iterations; ++k) {
      reset_stats();
      dt_hmc = 0.5 * (dt_min + dt_max);
      for (unsigned int j = 0; j < draws; ++j) {
        n_accepted_samples += single_step();
        n_total_samples++;
      }
      if (p_accept() > p_accept_target) dt_min = dt_hmc; else dt_max = dt_hmc;
      if (std::fabs(p_accept() - p_accept_target) < tolerance) converged = true;
    }
    if (converged) {
      std::cout << "  Tuned         dt_{HMC} = " << dt_hmc << std::endl;
    } else {
      dt_hmc = dt_original;
      std::cout << "  FAILED to tune, reverting to " << dt_hmc << std::endl;
    }
    std::cout << std::endl;
    reset_stats();
  }

private:
  double launch(unsigned reps) {
    double *x = phi_state_cur->device_mutable();
    if (qm)
      check(mlmcpi_path_hmc_draw(&qm->abi_action(), x, B, nt_hmc, dt_hmc, reps, action->get_seed(), action->get_chain0(),
                                 traj, work, (int32_t *)accept_flags.ptr(), (double *)energies.ptr(), nullptr), "path_hmc_draw");
    else
      check(mlmcpi_lattice_hmc_draw(&qft->abi_action(), x, B, nt_hmc, dt_hmc, reps, action->get_seed(),
                                    action->get_chain0(), traj, work, (int32_t *)accept_flags.ptr(),
                                    (double *)energies.ptr(), nullptr), "lattice_hmc_draw");
    traj += reps;
    std::vector<int32_t> flags = accept_flags.download<int32_t>();
    double acc = 0;
    for (int32_t f : flags) acc += f;
    accept = flags[0] != 0;
    return acc / B;
  }
  double step() { return launch(n_rep); }      // one draw
  double single_step() { return launch(1); }   // hmcsampler.cc:22-69

protected:
  const std::shared_ptr<Action> action;
  const unsigned int nt_hmc;
  mutable double dt_hmc;
  const unsigned int n_rep, n_burnin, B;
  mutable std::shared_ptr<SampleState> phi_state_cur;
  QMAction *qm = nullptr;
  QFTAction *qft = nullptr;
  void *work = nullptr;
  DeviceVector accept_flags, energies;
  uint32_t traj = 0;
};

/** sampler/overrelaxedheatbathsampler.hh:22-71 */
struct OverrelaxedHeatBathParameters {
  unsigned int n_sweep_heatbath = 1;
  unsigned int n_sweep_overrelax = 10;
  unsigned int n_burnin = 100;
  // overrelaxedheatbathsampler.hh:60-63.  false (the default here; the reference's template says true): the fixed sweep
  // order -- on the device the multicolour order instead of the reference's lexicographic one (a different, equally
  // valid fixed order; said once on stderr).  true: the reference's own loop, every sweep over a freshly shuffled index
  // set, through the site-at-a-time updates (sequential within a chain, chains in parallel): exact semantics, slow.
  bool random_order = false;
  unsigned int batch = 1;
};

/** overrelaxedheatbathsampler.cc:8-37: n_sweep_overrelax overrelaxation sweeps, then
 *  n_sweep_heatbath heat-bath sweeps, every sample accepted.
 *
 *  No copy per draw.  The reference ends draw() with `phi_state->data = phi_state_cur->data` (:30); here the caller's
 *  state becomes a second holder of the buffer the new sample was written to (SampleState::share), and the sampler
 *  rotates a small pool of buffers: a draw reads the current sample (which the caller may still hold: it is not written)
 *  and lets the launches alternate between two buffers nobody else holds.  In the loop of
 *  MonteCarloSingleLevel::evaluate that settles at three buffers and zero copies. */
class OverrelaxedHeatBathSampler : public Sampler {
public:
  OverrelaxedHeatBathSampler(const std::shared_ptr<Action> action_, const OverrelaxedHeatBathParameters p)
      : Sampler(), action(action_), n_sweep_heatbath(p.n_sweep_heatbath), n_sweep_overrelax(p.n_sweep_overrelax),
        n_burnin(p.n_burnin), random_order(p.random_order) {
    if (!action->has_local_updates()) fatal("heat bath update not implemented for this action ");
    if (random_order) {  // overrelaxedheatbathsampler.hh:110-116: the action's index set, or every entry
      index_map = action->get_heatbath_indexset();
      if (index_map.empty()) {
        index_map.resize(action->sample_size());
        for (unsigned int l = 0; l < index_map.size(); ++l) index_map[l] = l;
      }
      d_index = std::make_shared<DeviceVector>((index_map.size() + 1) / 2);  // uint32 entries in 8-byte units
    } else {
      static bool said = false;
      if (!said) std::cerr << "NOTE: random_order = false: device sweeps run in multicolour order, not lexicographically" << std::endl;
      said = true;
    }
    phi_state_cur = std::make_shared<SampleState>(action->sample_size(), p.batch);
    action->initialise_state(phi_state_cur);
    std::shared_ptr<SampleState> tmp = std::make_shared<SampleState>(action->sample_size(), p.batch);
    for (unsigned int i = 0; i < n_burnin; ++i) draw(tmp);
    reset_stats();
  }
  void draw(std::shared_ptr<SampleState> phi_state) override {
    advance();
    if (phi_state != phi_state_cur) phi_state->share(*phi_state_cur);
  }
  /** draw + QoI in one pass over the state: d_q[b] = the QoI (QoI::fused_kind()) of the new sample, summed inside the
   *  draw's last launch; d_acc != NULL: stats->record_sample(q) into the per-chain device moments d_acc[batch][5] as well
   *  (mlmcpi_stats_accumulate's layout), in the same call.  Returns false (and does nothing) when the action cannot fuse it. */
  bool draw_with_qoi(std::shared_ptr<SampleState> phi_state, int qoi_kind, double *d_q, double *d_acc = nullptr) {
    if (!advance(qoi_kind, d_q, d_acc)) return false;
    if (phi_state != phi_state_cur) phi_state->share(*phi_state_cur);
    return true;
  }
  /** the draw without handing the sample out (callers that read current_state()) */
  bool advance(int qoi_kind = 0, double *d_q = nullptr, double *d_acc = nullptr) {
    if (qoi_kind && (n_sweep_heatbath == 0 || random_order)) return false;
    if (random_order) {  // overrelaxedheatbathsampler.cc:8-31 as written: shuffle, then one local update per index
      // (site_updates works in place; a sample handed out earlier keeps its values: SampleState::device_mutable detaches)
      for (unsigned int s = 0; s < n_sweep_overrelax + n_sweep_heatbath; ++s) {
        std::shuffle(index_map.begin(), index_map.end(), engine);
        check(mlmcpi_copy_h2d(d_index->ptr(), index_map.data(), index_map.size() * sizeof(uint32_t), nullptr), "copy_h2d");
        action->set_site_step(sweep_counter + s);
        action->site_updates(phi_state_cur, (const uint32_t *)d_index->ptr(), (unsigned int)index_map.size(), 0, s >= n_sweep_overrelax);
      }
    } else
    if (n_sweep_overrelax + n_sweep_heatbath > 0) {
      std::shared_ptr<DeviceBuffer> src = phi_state_cur->buffer();
      // `src` is held by phi_state_cur, by this local variable and by the pool if it came from there: one holder more
      // means somebody outside still reads the current sample
      long own = 2;
      for (auto &c : pool) own += (c == src) ? 1 : 0;
      const bool src_is_lent = src.use_count() > own;
      std::shared_ptr<DeviceBuffer> w0 = free_buffer(src, nullptr);
      std::shared_ptr<DeviceBuffer> w1 = src_is_lent ? free_buffer(src, w0) : src;
      const int where = qoi_kind ? action->sweep_from_qoi(src->p, w0->p, w1->p, phi_state_cur->batch(), n_sweep_overrelax,
                                                          n_sweep_heatbath, sweep_counter, qoi_kind, d_q, d_acc)
                                 : action->sweep_from(src->p, w0->p, w1->p, phi_state_cur->batch(), n_sweep_overrelax,
                                                      n_sweep_heatbath, sweep_counter);
      if (where < 0) return false;
      phi_state_cur->adopt(where == 0 ? w0 : w1);
    }
    sweep_counter += n_sweep_overrelax + n_sweep_heatbath;
    accept = true;
    n_total_samples++;
    n_accepted_samples++;
    return true;
  }
  void set_state(std::shared_ptr<SampleState> phi_state) override { phi_state_cur->data = phi_state->data; }
  std::shared_ptr<SampleState> current_state() { return phi_state_cur; }
  size_t pool_size() const { return pool.size(); }

protected:
  /** a pool buffer nobody else holds (other than `a`, `b`), allocated on first need */
  std::shared_ptr<DeviceBuffer> free_buffer(const std::shared_ptr<DeviceBuffer> &a, const std::shared_ptr<DeviceBuffer> &b) {
    for (auto &c : pool)
      if (c != a && c != b && c.use_count() == 1) return c;
    pool.push_back(std::make_shared<DeviceBuffer>(phi_state_cur->bytes()));
    return pool.back();
  }
  const std::shared_ptr<Action> action;
  const unsigned int n_sweep_heatbath, n_sweep_overrelax, n_burnin;
  bool random_order;
  std::vector<unsigned int> index_map;      // random_order: the index set, reshuffled per sweep
  std::shared_ptr<DeviceVector> d_index;    // ... and its device copy
  std::mt19937_64 engine{871417};           // overrelaxedheatbathsampler.hh:111
  mutable std::shared_ptr<SampleState> phi_state_cur;
  std::vector<std::shared_ptr<DeviceBuffer>> pool;  // buffers that have carried a sample (some may still be lent out)
  uint32_t sweep_counter = 0;
};

class HMCSamplerFactory : public SamplerFactory {
public:
  explicit HMCSamplerFactory(const HMCParameters p) : param(p) {}
  std::shared_ptr<Sampler> get(std::shared_ptr<Action> action) override { return std::make_shared<HMCSampler>(action, param); }
private:
  const HMCParameters param;
};

class OverrelaxedHeatBathSamplerFactory : public SamplerFactory {
public:
  explicit OverrelaxedHeatBathSamplerFactory(const OverrelaxedHeatBathParameters p) : param(p) {}
  std::shared_ptr<Sampler> get(std::shared_ptr<Action> action) override {
    return std::make_shared<OverrelaxedHeatBathSampler>(action, param);
  }
private:
  const OverrelaxedHeatBathParameters param;
};

/** The exact sampler of the harmonic oscillator.  In the reference HarmonicOscillatorAction is itself a Sampler
 *  (harmonicoscillatoraction.hh:83, draw at harmonicoscillatoraction.cc:59-66, Cholesky factor at :38-56) and the
 *  drivers select it with sampler = 'exact'.  Here the factor is built once on the host and every draw is one fp64
 *  matrix-core product for the whole batch of chains (mlmcpi_path_exact_draw). */
class HarmonicOscillatorExactSampler : public Sampler {
public:
  HarmonicOscillatorExactSampler(const std::shared_ptr<Action> action_, unsigned int batch = 1)
      : Sampler(), action(std::dynamic_pointer_cast<HarmonicOscillatorAction>(action_)), B(batch) {
    if (!action) fatal("exact sampler only for the harmonic oscillator action");
    const size_t M = action->sample_size();
    std::vector<double> LT(M * M);
    check(mlmcpi_ho_cholesky_factor(&action->abi_action(), LT.data()), "ho_cholesky_factor");
    factor = std::make_shared<DeviceVector>(M * M);
    check(mlmcpi_copy_h2d(factor->ptr(), LT.data(), M * M * sizeof(double), nullptr), "copy_h2d");
    state = std::make_shared<SampleState>(M, B);
  }
  void draw(std::shared_ptr<SampleState> x_path) override {
    check(mlmcpi_path_exact_draw(&action->abi_action(), (const double *)factor->ptr(), state->device_mutable(), B,
                                 action->get_seed() ^ 0x45584143ull, action->get_chain0(), step++, nullptr), "path_exact_draw");
    accept = true;
    n_total_samples++;
    n_accepted_samples++;
    x_path->data = state->data;
  }
  void set_state(std::shared_ptr<SampleState>) override {}  // independent draws: there is no chain state

private:
  const std::shared_ptr<HarmonicOscillatorAction> action;
  const unsigned int B;
  std::shared_ptr<DeviceVector> factor;
  std::shared_ptr<SampleState> state;
  uint32_t step = 0;
};

/** The exact sampler of the Gaussian free field (GFFAction is a Sampler in the reference, gffaction.hh:120,
 *  draw at gffaction.cc:200-213): spectral synthesis on the device, mlmcpi_lattice_exact_draw. */
class GFFExactSampler : public Sampler {
public:
  GFFExactSampler(const std::shared_ptr<Action> action_, unsigned int batch = 1)
      : Sampler(), action(std::dynamic_pointer_cast<GFFAction>(action_)), B(batch) {
    if (!action) fatal("exact sampler only for the GFF action");
    if (action->plain()) {
      size_t bytes = 0;
      check(mlmcpi_lattice_exact_workspace_bytes(&action->abi_action(), B, &bytes), "lattice_exact_workspace_bytes");
      check(mlmcpi_malloc(&work, bytes), "mlmcpi_malloc");
    }
    state = std::make_shared<SampleState>(action->sample_size(), B);
  }
  ~GFFExactSampler() { mlmcpi_free(work); }
  void draw(std::shared_ptr<SampleState> phi_state) override {
    if (action->plain())  // the finest level: spectral synthesis at any size
      check(mlmcpi_lattice_exact_draw(&action->abi_action(), state->device_mutable(), B, action->get_seed() ^ 0x45584143ull,
                                      action->get_chain0(), step++, work, nullptr), "lattice_exact_draw");
    else  // a level of the hierarchy (rotated and / or Gibbs smoothed): GFFAction::draw, gffaction.cc:200-213
      action->draw_level(state, step++);
    accept = true;
    n_total_samples++;
    n_accepted_samples++;
    phi_state->data = state->data;
  }
  void set_state(std::shared_ptr<SampleState>) override {}

private:
  const std::shared_ptr<GFFAction> action;
  const unsigned int B;
  void *work = nullptr;
  std::shared_ptr<SampleState> state;
  uint32_t step = 0;
};

/** sampler = 'exact' (driver_qm.cc / driver_qft.cc): the harmonic oscillator and the GFF are their own samplers
 *  (driver_qft.cc:84-85: GFFSamplerFactory for every level of a GFF hierarchy) */
class ExactSamplerFactory : public SamplerFactory {
public:
  explicit ExactSamplerFactory(unsigned int batch_ = 1) : batch(batch_) {}
  std::shared_ptr<Sampler> get(std::shared_ptr<Action> action) override {
    if (std::dynamic_pointer_cast<GFFAction>(action)) return std::make_shared<GFFExactSampler>(action, batch);
    return std::make_shared<HarmonicOscillatorExactSampler>(action, batch);
  }
private:
  const unsigned int batch;
};

}  // namespace mlmcpi
#endif
