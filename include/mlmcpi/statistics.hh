// statistics.hh -- running statistics with windowed autocorrelations (common/statistics.{hh,cc}), single- and
// multi-rank.
//
// The state of the estimator IS the buffer that crosses ranks:
//     [ avg, avg_longterm, avg2_longterm, avg3_longterm, avg4_longterm, n, n_longterm, S_k[0 .. k_max) ]
// (SURVEY 2.3).  record_sample applies the reference's recurrences to it (statistics.cc:4-27, same operations in the
// same order: single-rank results are bit-identical to the reference's build without USE_MPI, tests/test_abi_host.py).
// Every estimator of the reference is a function of rank-AVERAGES of the first five entries and of S_k and of rank-SUMS
// of the two counts (statistics.cc:29-95: mpi_allreduce_avg / mpi_allreduce_sum, one scalar or vector call per
// quantity, ~10 collectives per convergence check).  Here they all come out of ONE all-reduce(sum) of the buffer through
// an Exchange (exchange.hh): RCCL over xGMI in production.  Without an open StatsSync every getter performs that
// reduction itself, like the reference's getters; MonteCarloSingleLevel opens one StatsSync per pass of its do-while.
#ifndef MLMCPI_STATISTICS_HH
#define MLMCPI_STATISTICS_HH
#include <cmath>
#include <deque>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "exchange.hh"

namespace mlmcpi {

class Statistics {
public:
  enum Slot { AVG = 0, AVG_LT = 1, AVG2_LT = 2, AVG3_LT = 3, AVG4_LT = 4, N = 5, N_LT = 6, SK0 = 7 };

  Statistics(const std::string label_, const unsigned int k_max_, std::shared_ptr<Exchange> exchange_ = nullptr)
      : obj_label(label_), k_max(k_max_), exchange(exchange_), synced(false) {
    hard_reset();
  }
  std::string label() const { return obj_label; }
  void set_exchange(std::shared_ptr<Exchange> e) { exchange = e; }
  int n_ranks() const { return exchange ? exchange->size() : 1; }

  /** statistics.hh:120-124 */
  void reset() { buf[N] = 0.0; buf[AVG] = 0.0; }
  /** statistics.hh:127-136 */
  void hard_reset() {
    buf.assign(SK0 + k_max, 0.0);
    Q_k.clear();
  }

  /** statistics.cc:4-27 */
  void record_sample(const double Q) {
    const double n = (buf[N] += 1.0), nl = (buf[N_LT] += 1.0);
    Q_k.push_front(Q);
    if (Q_k.size() > k_max) Q_k.pop_back();
    buf[AVG] = ((n - 1.0) * buf[AVG] + Q) / (1.0 * n);
    buf[AVG_LT] = ((nl - 1.0) * buf[AVG_LT] + Q) / (1.0 * nl);
    buf[AVG2_LT] = ((nl - 1.0) * buf[AVG2_LT] + Q * Q) / (1.0 * nl);
    buf[AVG3_LT] = ((nl - 1.0) * buf[AVG3_LT] + Q * Q * Q) / (1.0 * nl);
    buf[AVG4_LT] = ((nl - 1.0) * buf[AVG4_LT] + Q * Q * Q * Q) / (1.0 * nl);
    for (unsigned int k = 0; k < Q_k.size(); ++k) {
      const double N_k = nl - k;
      buf[SK0 + k] = ((N_k - 1.0) * buf[SK0 + k] + Q_k[0] * Q_k[k]) / (1.0 * N_k);
    }
  }

  /** statistics.cc:30-36 (note S_k[0], the windowed <Q^2>, not avg2_longterm: SURVEY A.3) */
  double variance() const {
    const std::vector<double> g = global();
    return 1.0 * g[N_LT] / (g[N_LT] - 1.0) * (g[SK0] - g[AVG_LT] * g[AVG_LT]);
  }
  /** statistics.cc:39-48 */
  double variance_error() const {
    const std::vector<double> g = global();
    const double a = g[AVG_LT];
    return std::sqrt(1.0 / g[N_LT] * (g[AVG4_LT] - 4 * a * g[AVG3_LT] + 8 * a * a * g[AVG2_LT] - g[AVG2_LT] * g[AVG2_LT] -
                                     4 * a * a * a * a));
  }
  /** statistics.cc:50-57 */
  double average() const { return global()[AVG]; }
  /** statistics.cc:59-62 */
  double error() const {
    const std::vector<double> g = global();
    return std::sqrt(tau_int_of(g) * (1.0 * g[N_LT] / (g[N_LT] - 1.0) * (g[SK0] - g[AVG_LT] * g[AVG_LT])) / (1.0 * g[N]));
  }
  /** statistics.cc:64-80 */
  std::vector<double> auto_corr() const {
    const std::vector<double> g = global();
    std::vector<double> c(g.begin() + SK0, g.begin() + SK0 + k_max);
    for (double &v : c) v -= g[AVG_LT] * g[AVG_LT];
    return c;
  }
  /** statistics.cc:82-91 */
  double tau_int() const { return tau_int_of(global()); }
  unsigned int autocorr_window() const { return k_max; }
  /** statistics.cc:93-95: samples over all ranks */
  unsigned int samples() const { return (unsigned int)global()[N]; }
  unsigned int local_samples() const { return (unsigned int)buf[N]; }

  /** the packed state of this rank (what travels) */
  const std::vector<double> &packed() const { return buf; }

  /** One reduction for everything asked until close_sync(): the buffer plus one slot per rank that carries
   *  `local_value` (MonteCarloSingleLevel: the rank's local sample count, for the termination test). */
  void open_sync(const double local_value = 0.0) {
    const int size = n_ranks(), rank = exchange ? exchange->rank() : 0;
    cache = buf;
    cache.resize(buf.size() + size, 0.0);
    cache[buf.size() + rank] = local_value;
    if (exchange && size > 1) exchange->allreduce_sum(cache.data(), cache.size());
    for (int s = AVG; s <= AVG4_LT; ++s) cache[s] /= size;  // mpi_allreduce_avg
    for (unsigned int k = 0; k < k_max; ++k) cache[SK0 + k] /= size;
    synced = true;
  }
  void close_sync() { synced = false; }
  /** `local_value` of rank r as handed to the open sync */
  double gathered(const int r) const { return cache[buf.size() + r]; }

private:
  std::vector<double> global() const {
    if (synced) return cache;
    const int size = n_ranks();
    std::vector<double> g(buf);
    if (exchange && size > 1) {
      exchange->allreduce_sum(g.data(), g.size());
      for (int s = AVG; s <= AVG4_LT; ++s) g[s] /= size;
      for (unsigned int k = 0; k < k_max; ++k) g[SK0 + k] /= size;
    }
    return g;
  }
  double tau_int_of(const std::vector<double> &g) const {
    const double a2 = g[AVG_LT] * g[AVG_LT];
    double t = 0.0;
    for (unsigned int k = 1; k < k_max; ++k) t += (1. - k / (1.0 * g[N_LT])) * (g[SK0 + k] - a2);
    return std::fmax(1.0, 1.0 + 2.0 * t / (g[SK0] - a2));
  }

  const std::string obj_label;
  const unsigned int k_max;
  std::shared_ptr<Exchange> exchange;
  std::vector<double> buf;    // packed state, see Slot
  std::deque<double> Q_k;     // the last k_max samples (rank local, never exchanged)
  std::vector<double> cache;  // reduced buffer + per-rank slots while a sync is open
  bool synced;
};

/** RAII form of open_sync / close_sync */
class StatsSync {
public:
  StatsSync(Statistics &s_, const double local_value = 0.0) : s(s_) { s.open_sync(local_value); }
  ~StatsSync() { s.close_sync(); }

private:
  Statistics &s;
};

/** statistics.cc:101-116 */
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
