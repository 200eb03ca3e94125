// montecarlo.hh -- the single-level estimator loop (montecarlo/montecarlosinglelevel.cc:23-94)
// driving the device samplers through the reference's interfaces: draw, evaluate, record_sample.
#ifndef MLMCPI_MONTECARLO_HH
#define MLMCPI_MONTECARLO_HH
#include <cmath>

#include "qoi.hh"
#include "sampler.hh"
#include "statistics.hh"

namespace mlmcpi {

struct SingleLevelMCParameters {
  unsigned int n_burnin = 100;
  unsigned int n_samples = 0;  // 0: adaptive (tau_int * 2/eps^2 * variance)
  double epsilon = 1.0;
  unsigned int n_autocorr_window = 20, n_min_samples_qoi = 100;
};

class MonteCarloSingleLevel {
public:
  /** `exchange` = nullptr: one rank, as the reference built without USE_MPI.  Otherwise every rank runs its own chain
   *  (the sampler's Philox chain index should be the rank, SamplerFactory's business) and the ranks meet in ONE
   *  all-reduce per pass of the do-while below. */
  MonteCarloSingleLevel(std::shared_ptr<Action> action_, std::shared_ptr<QoI> qoi_, std::shared_ptr<SamplerFactory> f,
                        const SingleLevelMCParameters p, std::shared_ptr<Exchange> exchange_ = nullptr)
      : action(action_), sampler(f->get(action_)), qoi(qoi_), param(p), exchange(exchange_),
        stats_Q(std::make_shared<Statistics>("Q", p.n_autocorr_window, exchange_)) {}

  /** montecarlosinglelevel.cc:23-94 */
  void evaluate() {
    std::shared_ptr<SampleState> phi_state = std::make_shared<SampleState>(action->sample_size());
    const int rank = exchange ? exchange->rank() : 0, size = exchange ? exchange->size() : 1;
    stats_Q->hard_reset();
    for (unsigned int i = 0; i < param.n_burnin; ++i) {
      sampler->draw(phi_state);
      stats_Q->record_sample(qoi->evaluate(phi_state));
    }
    const double two_epsilon_inv2 = 2. / (param.epsilon * param.epsilon);
    stats_Q->reset();
    unsigned int n_target = param.n_samples > 0 ? param.n_samples : param.n_min_samples_qoi;
    unsigned int n_local_target = distribute_n(n_target, rank, size);
    bool sufficient_stats = false;
    n_passes = 0;
    do {
      for (unsigned int k = stats_Q->local_samples(); k < n_local_target; ++k) {
        sampler->draw(phi_state);
        stats_Q->record_sample(qoi->evaluate(phi_state));
      }
      // The reference reads tau_int() and variance() (five scalar all-reduces, one vector all-reduce) and then
      // all-reduces the logical AND of "this rank has its share" (:78-86).  One reduction serves all of it: the packed
      // statistics buffer plus one slot per rank carrying that rank's local sample count.
      StatsSync sync(*stats_Q, (double)stats_Q->local_samples());
      if (param.n_samples == 0) n_target = (unsigned int)std::ceil(stats_Q->tau_int() * two_epsilon_inv2 * stats_Q->variance());
      n_local_target = distribute_n(n_target, rank, size);
      sufficient_stats = true;
      for (int r = 0; r < size; ++r) sufficient_stats = sufficient_stats && stats_Q->gathered(r) >= distribute_n(n_target, r, size);
      ++n_passes;
    } while (!sufficient_stats);
  }
  void show_statistics() {
    StatsSync sync(*stats_Q);  // collective: every rank calls show_statistics, rank 0 prints
    if (!exchange || exchange->rank() == 0) std::cout << *stats_Q << std::endl;
  }
  std::shared_ptr<Statistics> get_statistics() { return stats_Q; }
  std::shared_ptr<Sampler> get_sampler() { return sampler; }
  unsigned int passes() const { return n_passes; }  // collectives issued by the last evaluate()

private:
  std::shared_ptr<Action> action;
  std::shared_ptr<Sampler> sampler;
  std::shared_ptr<QoI> qoi;
  const SingleLevelMCParameters param;
  std::shared_ptr<Exchange> exchange;
  std::shared_ptr<Statistics> stats_Q;
  unsigned int n_passes = 0;
};

}  // namespace mlmcpi
#endif
