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
  MonteCarloSingleLevel(std::shared_ptr<Action> action_, std::shared_ptr<QoI> qoi_, std::shared_ptr<SamplerFactory> f,
                        const SingleLevelMCParameters p)
      : action(action_), sampler(f->get(action_)), qoi(qoi_), param(p), stats_Q(std::make_shared<Statistics>("Q", p.n_autocorr_window)) {}

  void evaluate() {
    std::shared_ptr<SampleState> phi_state = std::make_shared<SampleState>(action->sample_size());
    stats_Q->hard_reset();
    for (unsigned int i = 0; i < param.n_burnin; ++i) {
      sampler->draw(phi_state);
      stats_Q->record_sample(qoi->evaluate(phi_state));
    }
    const double two_epsilon_inv2 = 2. / (param.epsilon * param.epsilon);
    stats_Q->reset();
    unsigned int n_target = param.n_samples > 0 ? param.n_samples : param.n_min_samples_qoi;
    bool sufficient = false;
    do {
      for (unsigned int k = stats_Q->local_samples(); k < n_target; ++k) {
        sampler->draw(phi_state);
        stats_Q->record_sample(qoi->evaluate(phi_state));
      }
      if (param.n_samples == 0) n_target = (unsigned int)std::ceil(stats_Q->tau_int() * two_epsilon_inv2 * stats_Q->variance());
      sufficient = stats_Q->local_samples() >= n_target;
    } while (!sufficient);
  }
  void show_statistics() { std::cout << *stats_Q << std::endl; }
  std::shared_ptr<Statistics> get_statistics() { return stats_Q; }
  std::shared_ptr<Sampler> get_sampler() { return sampler; }

private:
  std::shared_ptr<Action> action;
  std::shared_ptr<Sampler> sampler;
  std::shared_ptr<QoI> qoi;
  const SingleLevelMCParameters param;
  std::shared_ptr<Statistics> stats_Q;
};

}  // namespace mlmcpi
#endif
