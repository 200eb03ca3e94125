// driver_single_level.cc -- counterpart of the single-level branch of the reference's drivers
// (driver_qm.cc:98-429, driver_qft.cc:100-459) on device chains: builds lattice, action, QoI and
// sampler factory, runs MonteCarloSingleLevel::evaluate and prints the statistics and, where the
// reference has one, the analytic value.  Parameters come from the command line (the reference's
// parameter-file parser is plumbing outside the hot path).
//   driver_single_level --action harmonicoscillator --M_lat 128 --T_final 4 --sampler hmc --n_samples 100000
//   driver_single_level --action schwinger --Mt_lat 16 --beta 1 --sampler heatbath --n_samples 20000
#include <cstring>
#include <map>

#include "mlmcpi/montecarlo.hh"

using namespace mlmcpi;

int main(int argc, char **argv) {
  std::map<std::string, std::string> o = {{"action", "harmonicoscillator"}, {"M_lat", "128"}, {"T_final", "4.0"},
      {"Mt_lat", "16"}, {"m0", "1.0"}, {"mu2", "1.0"}, {"lambda", "1.0"}, {"x0", "1.0"}, {"beta", "1.0"}, {"mass", "10.0"},
      {"sampler", "hmc"}, {"nt", "100"}, {"dt", "0.1"}, {"n_burnin", "100"}, {"n_samples", "20000"}, {"n_sweep_overrelax", "10"},
      {"n_sweep_heatbath", "1"}, {"autotune", "1"}, {"window", "20"}};
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::strncmp(argv[i], "--", 2) || !o.count(argv[i] + 2)) fatal(std::string("unknown option ") + argv[i]);
    o[argv[i] + 2] = argv[i + 1];
  }
  auto num = [&](const char *k) { return std::stod(o[k]); };
  std::shared_ptr<Action> action;
  std::shared_ptr<QoI> qoi;
  double analytic = NAN;
  const std::string a = o["action"];
  if (a == "harmonicoscillator" || a == "quarticoscillator" || a == "rotor") {
    auto lat = std::make_shared<Lattice1D>((unsigned)num("M_lat"), num("T_final"));
    if (a == "harmonicoscillator") {
      auto act = std::make_shared<HarmonicOscillatorAction>(lat, RenormalisationNone, num("m0"), num("mu2"));
      analytic = act->Xsquared_analytical();
      action = act;
      qoi = std::make_shared<QoIXsquared>(lat);
    } else if (a == "quarticoscillator") {
      action = std::make_shared<QuarticOscillatorAction>(lat, RenormalisationNone, num("m0"), num("mu2"), num("lambda"), num("x0"));
      qoi = std::make_shared<QoIXsquared>(lat);
    } else {
      action = std::make_shared<RotorAction>(lat, RenormalisationNone, num("m0"));
      qoi = std::make_shared<QoISusceptibility>(lat);
    }
  } else if (a == "schwinger" || a == "gff") {
    auto lat = std::make_shared<Lattice2D>((unsigned)num("Mt_lat"), (unsigned)num("Mt_lat"), CoarsenBoth);
    if (a == "schwinger") {
      action = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, num("beta"));
      qoi = std::make_shared<QoIAvgPlaquette>(lat);
    } else {
      action = std::make_shared<GFFAction>(lat, nullptr, num("mass"));
      qoi = std::make_shared<QoI2DPhiSquared>(lat);
    }
  } else {
    fatal("unknown action " + a);
  }
  std::cout << "Action: " << action->info_string() << std::endl;
  std::shared_ptr<SamplerFactory> factory;
  if (o["sampler"] == "hmc") {
    HMCParameters hp;
    hp.nt = (unsigned)num("nt"); hp.dt = num("dt"); hp.n_burnin = (unsigned)num("n_burnin"); hp.autotune = num("autotune") != 0;
    factory = std::make_shared<HMCSamplerFactory>(hp);
  } else {
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_overrelax = (unsigned)num("n_sweep_overrelax"); hb.n_sweep_heatbath = (unsigned)num("n_sweep_heatbath");
    hb.n_burnin = (unsigned)num("n_burnin");
    factory = std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb);
  }
  SingleLevelMCParameters mp;
  mp.n_burnin = (unsigned)num("n_burnin"); mp.n_samples = (unsigned)num("n_samples"); mp.n_autocorr_window = (unsigned)num("window");
  MonteCarloSingleLevel mc(action, qoi, factory, mp);
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
