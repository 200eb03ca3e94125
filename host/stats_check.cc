// stats_check.cc -- CPU-only harness for the cross-rank statistics of include/mlmcpi/statistics.hh (no GPU; RCCL only in
// the --rccl-join mode):
//   stats_check single  K NBURN            < samples      one rank
//   stats_check threads W K NBURN          < samples      W ranks as threads (ThreadExchange); sample i goes to rank i % W
//   stats_check loop    W K NMIN EPS N     synthetic AR(1) chains: the do-while of MonteCarloSingleLevel::evaluate
//                                          (montecarlosinglelevel.cc:57-87) with ONE reduction per pass
// Prints %.17g values, one per line, that tests/test_stats_exchange.py compares with the compiled reference
// (oracle/_ref) and with the reference's combination rules (statistics.cc:29-95).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "mlmcpi/statistics.hh"

using namespace mlmcpi;

static void dump(const Statistics &s) {
  // every getter is a collective when an exchange is attached: all ranks call dump(), rank 0 prints
  const double avg = s.average(), var = s.variance(), verr = s.variance_error(), tau = s.tau_int(), err = s.error();
  const unsigned n = s.samples();
  const std::vector<double> c = s.auto_corr();
  if (s.n_ranks() > 1 && false) return;
  std::printf("%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n%u\n", avg, var, verr, tau, err, n);
  for (double v : c) std::printf("%.17g\n", v);
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  if (mode == "--rccl-join" && argc >= 6) {
    // stats_check --rccl-join PATH RANK WORLD TIMEOUT_S: join an RCCL group through the rendezvous file, the way host/driver
    // does.  Used by the tests for the failure convention (an unusable file is "ERROR: ..." + EXIT_FAILURE before RCCL or a
    // GPU is touched); with a live group it verifies it and prints the communicator's rank count.
    RcclExchange ex(std::atoi(argv[3]), std::atoi(argv[4]), std::string(argv[2]), 0, std::atof(argv[5]));
    ex.verify(std::atoi(argv[4]));
    std::printf("%d\n", ex.size());
    return 0;
  }
  if (mode == "single" || mode == "threads") {
    const int W = mode == "threads" ? std::atoi(argv[2]) : 1;
    const unsigned K = std::atoi(argv[mode == "threads" ? 3 : 2]), nburn = std::atoi(argv[mode == "threads" ? 4 : 3]);
    std::vector<double> q;
    double v;
    while (std::scanf("%lf", &v) == 1) q.push_back(v);
    auto hub = std::make_shared<ThreadExchangeHub>(W);
    std::vector<std::thread> threads;
    for (int r = 0; r < W; ++r)
      threads.emplace_back([&, r] {
        std::shared_ptr<Exchange> ex;
        if (W > 1) ex = std::make_shared<ThreadExchange>(hub, r);
        Statistics s("Q", K, ex);
        unsigned mine = 0;
        for (size_t i = r; i < q.size(); i += W) {
          s.record_sample(q[i]);
          if (++mine == nburn) s.reset();  // montecarlosinglelevel.cc:27-37: soft reset after the burn-in
        }
        // once through the reference-style getters (each its own reduction), once through one StatsSync
        const double a1 = s.average(), v1 = s.variance(), t1 = s.tau_int(), e1 = s.error();
        StatsSync sync(s);
        const bool same = a1 == s.average() && v1 == s.variance() && t1 == s.tau_int() && e1 == s.error();
        if (r == 0) {
          dump(s);
          std::printf("%d\n", same ? 1 : 0);
        } else {
          (void)s.auto_corr();
        }
      });
    for (auto &t : threads) t.join();
    return 0;
  }
  if (mode == "loop") {
    const int W = std::atoi(argv[2]);
    const unsigned K = std::atoi(argv[3]), n_min = std::atoi(argv[4]);
    const double eps = std::atof(argv[5]);
    const unsigned n_fixed = argc > 6 ? std::atoi(argv[6]) : 0;
    auto hub = std::make_shared<ThreadExchangeHub>(W);
    std::vector<std::thread> threads;
    for (int r = 0; r < W; ++r)
      threads.emplace_back([&, r] {
        std::shared_ptr<Exchange> ex;
        if (W > 1) ex = std::make_shared<ThreadExchange>(hub, r);
        Statistics s("Q", K, ex);
        std::mt19937_64 eng(1234 + r);
        std::normal_distribution<double> g(0.0, 1.0);
        double x = 0.0;
        auto draw = [&] { x = 0.8 * x + 0.6 * g(eng); return 1.0 + x; };  // AR(1): tau_int = (1 + 0.8)/(1 - 0.8) = 9
        const double two_eps_inv2 = 2. / (eps * eps);
        unsigned n_target = n_fixed ? n_fixed : n_min, n_local = distribute_n(n_target, r, W), passes = 0;
        bool sufficient = false;
        do {
          for (unsigned k = s.local_samples(); k < n_local; ++k) s.record_sample(draw());
          StatsSync sync(s, (double)s.local_samples());
          if (!n_fixed) n_target = (unsigned)std::ceil(s.tau_int() * two_eps_inv2 * s.variance());
          n_local = distribute_n(n_target, r, W);
          sufficient = true;
          for (int q = 0; q < W; ++q) sufficient = sufficient && s.gathered(q) >= distribute_n(n_target, q, W);
          ++passes;
        } while (!sufficient && passes < 100);
        StatsSync sync(s);
        if (r == 0)
          std::printf("%u\n%u\n%u\n%.17g\n%.17g\n%.17g\n%.17g\n", passes, n_target, s.samples(), s.average(), s.error(), s.tau_int(),
                      s.variance());
      });
    for (auto &t : threads) t.join();
    return 0;
  }
  return 2;
}
