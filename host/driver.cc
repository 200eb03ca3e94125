// driver.cc -- counterpart of the reference's drivers (driver_qm.cc:98-429, driver_qft.cc:100-459) on device
// chains: builds lattice, action, QoI factory and sampler factory, runs MonteCarloSingleLevel::evaluate
// (method singlelevel) or MonteCarloMultiLevel::evaluate (method multilevel) and prints the statistics and, where
// the reference has one, the analytic value.  Parameters come from the command line (the reference's
// parameter-file parser is plumbing outside the hot path).
//   driver --action harmonicoscillator --M_lat 128 --T_final 4 --sampler hmc --n_samples 100000
//   driver --action schwinger --Mt_lat 16 --beta 1 --sampler heatbath --n_samples 20000
//   driver --method multilevel --action quarticoscillator --M_lat 256 --T_final 8 --sampler hierarchical --n_level 3 --epsilon 0.02
//   driver --method twolevel --action quarticoscillator --M_lat 256 --T_final 8 --coarsesampler hmc --n_samples 5000
//   driver --method multilevel --action schwinger --Mt_lat 16 --beta 2 --coarsening both --coarsesampler heatbath
//          --sampler hierarchical --n_level 2 --epsilon 0.005
//   driver --method throughput --action schwinger --Mt_lat 1024 --sampler heatbath --batch 32 --n_samples 20
//          (the sampling loop of MonteCarloSingleLevel on a batch of chains, statistics accumulated on the device:
//           prints link-updates/s of the C++ path)
// Several ranks (one process per GPU): start N copies with RANK / WORLD_SIZE / LOCAL_RANK set (torchrun, mpirun, a
// shell loop); they meet through RCCL (mlmcpi::RcclExchange, rendezvous file $MLMCPI_ID_FILE or
// /dev/shm/mlmcpi_id_<launcher pid>_$MASTER_PORT) and verify the group (RcclExchange::verify) or exit non-zero; rank r
// samples chain(s) r * batch ...
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <map>

#include "mlmcpi/multilevel.hh"

using namespace mlmcpi;

int main(int argc, char **argv) {
  std::map<std::string, std::string> o = {{"action", "harmonicoscillator"}, {"M_lat", "128"}, {"T_final", "4.0"},
      {"Mt_lat", "16"}, {"m0", "1.0"}, {"mu2", "1.0"}, {"lambda", "1.0"}, {"x0", "1.0"}, {"beta", "1.0"}, {"mass", "10.0"},
      {"sampler", "hmc"}, {"nt", "100"}, {"dt", "0.1"}, {"n_burnin", "100"}, {"n_samples", "20000"}, {"n_sweep_overrelax", "10"},
      {"n_sweep_heatbath", "1"}, {"autotune", "1"}, {"window", "20"}, {"method", "singlelevel"}, {"n_level", "3"},
      {"epsilon", "0.01"}, {"coarsening", "both"}, {"coarsesampler", "hmc"}, {"renormalisation", "none"}, {"n_meas", "200"},
      {"batch", "1"}, {"seed", "2481317"}, {"warmup", "5"}, {"random_order", "0"}};
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::strncmp(argv[i], "--", 2) || !o.count(argv[i] + 2)) fatal(std::string("unknown option ") + argv[i]);
    o[argv[i] + 2] = argv[i + 1];
  }
  auto num = [&](const char *k) { return std::stod(o[k]); };
  // ranks: one process per GPU, the launcher's environment names them
  auto env_int = [](const char *k, int d) { const char *v = std::getenv(k); return v ? std::atoi(v) : d; };
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", rank);
  const unsigned batch = (unsigned)num("batch");
  std::shared_ptr<Exchange> exchange;
  if (world > 1) {
    check(mlmcpi_set_device(local_rank), "mlmcpi_set_device");
    const char *idf = std::getenv("MLMCPI_ID_FILE");
    // unique per launch: the launcher's pid (the ranks are its children) and its port; a file a dead run left behind is
    // rejected by mlmcpi_comm_init_file in any case (it names its writer)
    const std::string id_file = idf ? idf : std::string("/dev/shm/mlmcpi_id_") + std::to_string((long)getppid()) + "_" +
                                                (std::getenv("MASTER_PORT") ? std::getenv("MASTER_PORT") : "0");
    auto rccl = std::make_shared<RcclExchange>(rank, world, id_file, local_rank);
    rccl->verify(world);  // the communicator's own rank count and a sum over ranks, or message + exit(EXIT_FAILURE)
    exchange = rccl;
    if (rank != 0) std::cout.setstate(std::ios_base::failbit);  // mpi_parallel::cout: the master prints
  }
  std::shared_ptr<Action> action;
  std::shared_ptr<QoI> qoi;
  std::shared_ptr<QoIFactory> qoi_factory;
  std::shared_ptr<ConditionedFineActionFactory> cfa_factory;
  double analytic = NAN;
  const std::string a = o["action"];
  const RenormalisationType renorm = o["renormalisation"] == "perturbative" ? RenormalisationPerturbative
                                     : o["renormalisation"] == "exact"      ? RenormalisationNonperturbative
                                                                            : RenormalisationNone;
  const std::map<std::string, CoarseningType> coarsenings = {{"both", CoarsenBoth}, {"temporal", CoarsenTemporal},
      {"spatial", CoarsenSpatial}, {"alternate", CoarsenAlternate}, {"rotate", CoarsenRotate}};
  if (!coarsenings.count(o["coarsening"])) fatal("unknown coarsening " + o["coarsening"]);
  if (a == "harmonicoscillator" || a == "quarticoscillator" || a == "rotor") {
    auto lat = std::make_shared<Lattice1D>((unsigned)num("M_lat"), num("T_final"));
    if (a == "harmonicoscillator") {
      auto act = std::make_shared<HarmonicOscillatorAction>(lat, renorm, num("m0"), num("mu2"));
      analytic = act->Xsquared_analytical();
      action = act;
      qoi = std::make_shared<QoIXsquared>(lat);
      qoi_factory = std::make_shared<QoIXsquaredFactory>();
      cfa_factory = std::make_shared<GaussianConditionedFineActionFactory>();
    } else if (a == "quarticoscillator") {
      action = std::make_shared<QuarticOscillatorAction>(lat, renorm, num("m0"), num("mu2"), num("lambda"), num("x0"));
      qoi = std::make_shared<QoIXsquared>(lat);
      qoi_factory = std::make_shared<QoIXsquaredFactory>();
      cfa_factory = std::make_shared<GaussianConditionedFineActionFactory>();
    } else {
      action = std::make_shared<RotorAction>(lat, renorm, num("m0"));
      qoi = std::make_shared<QoISusceptibility>(lat);
      qoi_factory = std::make_shared<QoISusceptibilityFactory>();
      cfa_factory = std::make_shared<RotorConditionedFineActionFactory>();
    }
  } else if (a == "schwinger" || a == "gff") {
    auto lat = std::make_shared<Lattice2D>((unsigned)num("Mt_lat"), (unsigned)num("Mt_lat"), coarsenings.at(o["coarsening"]));
    if (a == "schwinger") {
      action = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, renorm, num("beta"));
      qoi = std::make_shared<QoIAvgPlaquette>(lat);
      qoi_factory = std::make_shared<QoIAvgPlaquetteFactory>();
      cfa_factory = std::make_shared<QuenchedSchwingerConditionedFineActionFactory>();
    } else {
      action = std::make_shared<GFFAction>(lat, nullptr, num("mass"));
      qoi = std::make_shared<QoI2DPhiSquared>(lat);
      qoi_factory = std::make_shared<QoI2DPhiSquaredFactory>();           // driver_qft.cc:247-252
      cfa_factory = std::make_shared<GFFConditionedFineActionFactory>();  // driver_qft.cc:331-333 (use --coarsening rotate)
    }
  } else {
    fatal("unknown action " + a);
  }
  action->set_seed((uint64_t)num("seed"), (uint32_t)rank * batch);  // chain index = Philox counter word: distinct per rank
  std::cout << "Action: " << action->info_string() << std::endl;
  auto basic_factory = [&](const std::string &name) -> std::shared_ptr<SamplerFactory> {
    if (name == "hmc") {
      HMCParameters hp;
      hp.nt = (unsigned)num("nt"); hp.dt = num("dt"); hp.n_burnin = (unsigned)num("n_burnin"); hp.autotune = num("autotune") != 0;
      hp.batch = batch;
      return std::make_shared<HMCSamplerFactory>(hp);
    }
    if (name == "exact") return std::make_shared<ExactSamplerFactory>();  // driver_qm.cc: sampler = 'exact' (harmonic oscillator, GFF)
    if (name != "heatbath") fatal("unknown sampler " + name);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_overrelax = (unsigned)num("n_sweep_overrelax"); hb.n_sweep_heatbath = (unsigned)num("n_sweep_heatbath");
    hb.n_burnin = (unsigned)num("n_burnin");
    hb.batch = batch;
    // heatbath.random_order of the reference's parameter file (template: true).  Default here: 0, the multicolour kernels.
    hb.random_order = num("random_order") != 0;
    std::cerr << "heatbath: random_order = " << (hb.random_order ? "true (shuffled index set, site-at-a-time updates)"
                                                                 : "false (multicolour sweep kernels)") << std::endl;
    return std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb);
  };
  std::shared_ptr<SamplerFactory> factory;
  if (o["sampler"] == "hierarchical") {  // driver_qm.cc:61-75: hierarchical sampler over `coarsesampler`
    if (!cfa_factory) fatal("no conditioned fine action for action " + a);
    HierarchicalParameters hier;
    hier.n_max_level = (unsigned)num("n_level"); hier.n_meas = (unsigned)num("n_meas");
    factory = std::make_shared<HierarchicalSamplerFactory>(basic_factory(o["coarsesampler"]), cfa_factory, hier);
  } else {
    factory = basic_factory(o["sampler"]);
  }
  if (o["method"] == "multilevel") {  // driver_qm.cc:340-398
    if (!cfa_factory || !qoi_factory) fatal("multilevel method is not available for action " + a);
    MultiLevelMCParameters mlp;
    mlp.n_level = (unsigned)num("n_level"); mlp.n_burnin = (unsigned)num("n_burnin"); mlp.epsilon = num("epsilon");
    mlp.n_autocorr_window = (unsigned)num("window"); mlp.n_meas = (unsigned)num("n_meas");
    if (world > 1) {  // level l on rank l % world (multilevel.hh), one all-reduce of the level table per pass
      mlp.level_rank = (unsigned)rank;
      mlp.level_ranks = (unsigned)world;
    }
    MonteCarloMultiLevel mlmc(action, qoi_factory, factory, cfa_factory, mlp);
    if (exchange) mlmc.set_exchange(std::make_shared<LevelExchangeOver>(exchange));
    mlmc.evaluate();
    std::cout << std::endl << "=== Multilevel MC ===" << std::endl;
    mlmc.show_statistics();
    if (!std::isnan(analytic))
      std::cout << std::setprecision(6) << " analytic result = " << analytic << std::endl
                << " |analytic - numerical| / error = " << std::fabs(analytic - mlmc.numerical_result()) / mlmc.statistical_error()
                << std::endl;
    return 0;
  }
  if (o["method"] == "twolevel") {  // driver_qm.cc:313-338
    if (!cfa_factory || !qoi_factory) fatal("twolevel method is not available for action " + a);
    TwoLevelMCParameters tp;
    tp.n_burnin = (unsigned)num("n_burnin"); tp.n_samples = (unsigned)num("n_samples"); tp.n_meas = (unsigned)num("n_meas");
    tp.n_autocorr_window = (unsigned)num("window");
    MonteCarloTwoLevel two(action, qoi_factory, basic_factory(o["coarsesampler"]), cfa_factory, tp);
    two.evaluate_difference();
    std::cout << std::endl << "=== Two level MC ===" << std::endl;
    two.show_statistics();
    return 0;
  }
  if (o["method"] == "throughput") {
    // The loop of montecarlosinglelevel.cc:59-77 over `batch` chains with nothing on the host per sample: draw (no copy:
    // the sampler lends its buffer), QoI into device memory, per-chain moments accumulated on the device.
    std::shared_ptr<Sampler> sampler = factory->get(action);
    auto phi_state = std::make_shared<SampleState>(action->sample_size(), batch);
    DeviceVector q(batch), acc(5 * (size_t)batch);
    const unsigned n_samples = (unsigned)num("n_samples"), warmup = (unsigned)num("warmup");
    auto sweeper = std::dynamic_pointer_cast<OverrelaxedHeatBathSampler>(sampler);
    const int fused = sweeper ? qoi->fused_kind() : 0;
    auto one = [&]() {  // draw + QoI + record_sample as one call where the sampler can; else the three steps
      if (fused && sweeper->draw_with_qoi(phi_state, fused, (double *)q.ptr(), (double *)acc.ptr())) return;
      sampler->draw(phi_state);
      qoi->evaluate_device(phi_state, (double *)q.ptr());
      check(mlmcpi_stats_accumulate((double *)acc.ptr(), (const double *)q.ptr(), batch, nullptr), "stats_accumulate");
    };
    for (unsigned i = 0; i < warmup; ++i) one();
    check(mlmcpi_stream_synchronize(nullptr), "sync");
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 0; i < n_samples; ++i) one();
    check(mlmcpi_stream_synchronize(nullptr), "sync");
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::vector<double> m = acc.download<double>();
    double n = 0, s1 = 0;
    for (unsigned b = 0; b < batch; ++b) { n += m[5 * b]; s1 += m[5 * b + 1]; }
    std::vector<double> tot = {n, s1, el};
    if (exchange) {  // sum of counts and of QoI sums; the slowest rank's time
      std::vector<double> times(world, 0.0);
      times[rank] = el;
      tot.insert(tot.end(), times.begin(), times.end());
      exchange->allreduce_sum(tot.data(), tot.size());
      tot[2] = 0;
      for (int r = 0; r < world; ++r) tot[2] = std::max(tot[2], tot[3 + r]);
    }
    const double sweeps = num("n_sweep_overrelax") + num("n_sweep_heatbath");
    const bool sweeping = o["sampler"] == "heatbath";
    const double units = sweeping ? (double)action->sample_size() * sweeps : (double)action->sample_size() * (num("nt") + 1);
    std::cout << std::setprecision(6) << "{\"driver\": \"host/driver (C++ Sampler::draw + QoI::evaluate_device + stats_accumulate)\", "
              << "\"ranks\": " << (exchange ? exchange->size() : 1) << ", \"batch\": " << batch << ", \"samples\": " << n_samples
              << ", \"ms_per_sample\": " << 1e3 * tot[2] / n_samples << ", \"updates_per_s\": " << std::scientific
              << units * batch * world * n_samples / tot[2] << std::fixed << ", \"qoi_mean\": " << tot[1] / tot[0] << "}" << std::endl;
    return 0;
  }
  if (o["method"] != "singlelevel") fatal("unknown method " + o["method"]);
  SingleLevelMCParameters mp;
  mp.n_burnin = (unsigned)num("n_burnin"); mp.n_samples = (unsigned)num("n_samples"); mp.n_autocorr_window = (unsigned)num("window");
  MonteCarloSingleLevel mc(action, qoi, factory, mp, exchange);
  mc.evaluate();
  std::cout << std::endl << "=== Single level MC ===" << std::endl;
  mc.show_statistics();
  mc.get_sampler()->show_stats();
  if (!std::isnan(analytic)) {  // driver_qm.cc:411-425
    auto st = mc.get_statistics();
    std::cout << std::setprecision(6) << " analytic result = " << analytic << std::endl
              << " |analytic - numerical| / error = " << std::fabs(analytic - st->average()) / st->error() << std::endl;
  }
  return 0;
}
