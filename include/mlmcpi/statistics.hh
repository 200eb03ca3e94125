// statistics.hh -- running statistics with windowed autocorrelations (common/statistics.{hh,cc}),
// host side, single rank; cross-rank combination goes through the packed moment buffer
// (mlmcpi_stats_accumulate + one all-reduce).
#ifndef MLMCPI_STATISTICS_HH
#define MLMCPI_STATISTICS_HH
#include <cmath>
#include <deque>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace mlmcpi {

class Statistics {
public:
  Statistics(const std::string label_, const unsigned int k_max_) : obj_label(label_), k_max(k_max_) { hard_reset(); }
  std::string label() const { return obj_label; }
  void reset() { n_samples = 0; avg = 0.0; }
  void hard_reset() {
    reset();
    Q_k.clear();
    S_k.assign(k_max, 0.0);
    avg_longterm = avg2_longterm = avg3_longterm = avg4_longterm = 0.0;
    n_samples_longterm = 0;
  }
  /** statistics.cc:4-27 */
  void record_sample(const double Q) {
    n_samples++;
    n_samples_longterm++;
    Q_k.push_front(Q);
    if (Q_k.size() > k_max) Q_k.pop_back();
    avg = ((n_samples - 1.0) * avg + Q) / (1.0 * n_samples);
    const double n = 1.0 * n_samples_longterm;
    avg_longterm = ((n - 1.0) * avg_longterm + Q) / n;
    avg2_longterm = ((n - 1.0) * avg2_longterm + Q * Q) / n;
    avg3_longterm = ((n - 1.0) * avg3_longterm + Q * Q * Q) / n;
    avg4_longterm = ((n - 1.0) * avg4_longterm + Q * Q * Q * Q) / n;
    for (unsigned int k = 0; k < Q_k.size(); ++k) {
      const unsigned int N_k = n_samples_longterm - k;
      S_k[k] = ((N_k - 1.0) * S_k[k] + Q_k[0] * Q_k[k]) / (1.0 * N_k);
    }
  }
  double variance() const { return 1.0 * n_samples_longterm / (n_samples_longterm - 1.0) * (S_k[0] - avg_longterm * avg_longterm); }
  double variance_error() const {
    const double a = avg_longterm;
    return std::sqrt(1.0 / n_samples_longterm * (avg4_longterm - 4 * a * avg3_longterm + 8 * a * a * avg2_longterm -
                                                  avg2_longterm * avg2_longterm - 4 * a * a * a * a));
  }
  double average() const { return avg; }
  double error() const { return std::sqrt(tau_int() * variance() / (1.0 * samples())); }
  std::vector<double> auto_corr() const {
    std::vector<double> c(S_k);
    for (double &v : c) v -= avg_longterm * avg_longterm;
    return c;
  }
  double tau_int() const {
    const std::vector<double> C = auto_corr();
    double t = 0.0;
    for (unsigned int k = 1; k < C.size(); ++k) t += (1. - k / (1.0 * n_samples_longterm)) * C[k];
    return std::fmax(1.0, 1.0 + 2.0 * t / C[0]);
  }
  unsigned int autocorr_window() const { return k_max; }
  unsigned int samples() const { return n_samples; }
  unsigned int local_samples() const { return n_samples; }

private:
  const std::string obj_label;
  const unsigned int k_max;
  unsigned int n_samples_longterm, n_samples;
  std::deque<double> Q_k;
  std::vector<double> S_k;
  double avg, avg_longterm, avg2_longterm, avg3_longterm, avg4_longterm;
};

inline std::ostream &operator<<(std::ostream &os, const Statistics &stats) {
  os << " " << std::setprecision(6) << std::fixed;
  os << stats.label() << ": Avg +/- Err = " << stats.average() << " +/- " << stats.error() << std::endl;
  os << " " << stats.label() << ": Var +/- Err = " << stats.variance() << " +/- " << stats.variance_error() << std::endl;
  os << std::setprecision(3) << std::fixed;
  os << " " << stats.label() << ": tau_{int}   = " << stats.tau_int() << std::endl;
  os << " " << stats.label() << ": window      = " << stats.autocorr_window() << std::endl;
  os << " " << stats.label() << ": # samples   = " << stats.samples() << std::endl;
  return os;
}

}  // namespace mlmcpi
#endif
