// test_host.cc -- self-checking test of the C++ host layer (include/mlmcpi/*.hh) on a GPU.
// Known answers: SURVEY.md 8(c) (recorded from the compiled reference).  Exit code 0 = all passed.
//   test_host            run all checks
//   test_host --fatal X  provoke error path X (the process must print "ERROR: ..." and exit(1))
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

#include "mlmcpi/multilevel.hh"

using namespace mlmcpi;

static int failures = 0;
#define EXPECT(cond, ...)                    \
  do {                                       \
    if (!(cond)) {                           \
      ++failures;                            \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);              \
      std::printf("\n");                     \
    }                                        \
  } while (0)

// Statistical comparison: prints the z-score and gates it.  The errors here are the reference's own windowed estimator
// (Statistics::error, tau_int over a window of 20-50 lags), which is known to come out low when the autocorrelation
// outlasts the window -- the reference's own config-1 run sits 3.2 of ITS sigmas from the closed form --
// hence the gate of 5 on them; the gates on robust chain-scatter errors live in tests/test_gpu_statistics.py (3 sigma,
// 2 sigma on the headline pair).
#define ZEXPECT(value, err, ref, gate, name)                                                          \
  do {                                                                                                \
    const double z__ = ((value) - (ref)) / (err);                                                     \
    std::printf(" [z] %s: %.6f +- %.6f vs %.6f  z = %+.2f (gate %g)\n", name, (double)(value), (double)(err), (double)(ref), z__, (double)(gate)); \
    EXPECT(std::fabs(z__) < (gate), "%s: z = %+.2f", name, z__);                                        \
  } while (0)

static bool close(double a, double b, double tol = 1e-12) { return std::fabs(a - b) <= tol * std::fmax(1.0, std::fabs(b)); }

static void fill_sin(std::shared_ptr<SampleState> s) {
  for (size_t l = 0; l < s->data.size(); ++l) s->data[l] = std::sin(l + 1.0);
}

int main(int argc, char **argv) {
  if (argc == 3 && !std::strcmp(argv[1], "--fatal")) {
    std::string what = argv[2];
    if (what == "gff_not_square") {
      GFFAction a(std::make_shared<Lattice2D>(8, 4, CoarsenBoth), nullptr, 1.0);
    } else if (what == "heatbath_on_quartic") {
      auto lat = std::make_shared<Lattice1D>(16, 4.0);
      auto act = std::make_shared<QuarticOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0, 1.0, 1.0);
      OverrelaxedHeatBathSampler s(act, OverrelaxedHeatBathParameters());
    } else if (what == "per_site_update") {  // action/action.hh:73-79: an action without local updates
      auto lat = std::make_shared<Lattice1D>(16, 4.0);
      auto act = std::make_shared<QuarticOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0, 1.0, 1.0);
      auto st = std::make_shared<SampleState>(16);
      act->heatbath_update(st, 3);
    } else if (what == "qoi_wrong_size") {
      auto lat = std::make_shared<Lattice1D>(16, 4.0);
      QoIXsquared q(lat);
      q.evaluate(std::make_shared<SampleState>(8));
    } else if (what == "coarsen_odd") {
      Lattice1D(7, 1.0).coarse_lattice();
    }
    return 0;  // not reached when the error path works
  }

  // ---- rotor M=16: S, force, and host/device mirroring ------------------------------------------
  {
    auto lat = std::make_shared<Lattice1D>(16, 4.0);
    auto act = std::make_shared<RotorAction>(lat, RenormalisationNone, 0.25);
    auto x = std::make_shared<SampleState>(16), f = std::make_shared<SampleState>(16);
    fill_sin(x);
    EXPECT(close(act->evaluate(x), 3.7784124306337965), "rotor S = %.17g", act->evaluate(x));
    act->force(x, f);
    const double want[4] = {0.83637060305662203, 0.76260008151417336, 0.087208194898979019, -0.58128546045789198};
    for (int j = 0; j < 4; ++j) EXPECT(close(f->data[j], want[j]), "rotor force[%d] = %.17g", j, (double)f->data[j]);
    x->data[0] = 0.5;  // host write must reach the device before the next kernel
    const double S1 = act->evaluate(x);
    x->data[0] = std::sin(1.0);
    EXPECT(!close(S1, 3.7784124306337965) && close(act->evaluate(x), 3.7784124306337965), "lazy host->device sync");
    QoISusceptibility chi(lat);
    EXPECT(std::isfinite(chi.evaluate(x)), "chi finite");
  }
  // ---- quartic M=16 ------------------------------------------------------------------------------
  {
    auto lat = std::make_shared<Lattice1D>(16, 4.0);
    auto act = std::make_shared<QuarticOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0, 1.0, 1.0);
    auto x = std::make_shared<SampleState>(16);
    fill_sin(x);
    EXPECT(close(act->evaluate(x), 20.749620303495885), "quartic S");
    EXPECT(close(QoIXsquared(lat).evaluate(x), 0.49705796311310979), "quartic X2");
    auto coarse = std::dynamic_pointer_cast<QMAction>(act->coarse_action());
    EXPECT(coarse && coarse->sample_size() == 8 && coarse->get_coarsening_level() == 1, "coarse action");
  }
  // ---- Schwinger 4x4 and GFF 4x4 --------------------------------------------------------------------
  {
    auto lat = std::make_shared<Lattice2D>(4, 4, CoarsenBoth);
    EXPECT(lat->link_cart2lin(-1, -1, 1) == 31, "link_cart2lin");
    const unsigned want_nb[8] = {1, 3, 4, 12, 5, 13, 7, 15};
    for (int k = 0; k < 8; ++k) EXPECT(lat->get_neighbour_vertices()[0][k] == want_nb[k], "neighbour %d", k);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
    auto x = std::make_shared<SampleState>(act->sample_size()), f = std::make_shared<SampleState>(act->sample_size());
    fill_sin(x);
    EXPECT(close(act->evaluate(x), 8.7055541417150231), "schwinger S");
    EXPECT(close(QoIAvgPlaquette(lat).evaluate(x), 0.45590286614281106), "plaquette");
    act->force(x, f);
    EXPECT(close(f->data[3], -1.9435851018986892), "schwinger force[3]");
    auto gff = std::make_shared<GFFAction>(lat, nullptr, 10.0);
    auto phi = std::make_shared<SampleState>(16);
    fill_sin(phi);
    EXPECT(close(gff->getmu2(), 6.25) && close(gff->evaluate(phi), 41.88143013055145), "gff S");
    EXPECT(close(QoI2DPhiSquared(lat).evaluate(phi), 0.49705796311310979), "phi2");
    // coarse action of the 4 x 4 lattice has beta = 0.25; copy_from_fine of the recorded state
    {
      auto cact = std::dynamic_pointer_cast<QuenchedSchwingerAction>(act->coarse_action());
      EXPECT(cact && close(cact->getbeta(), 0.25) && cact->sample_size() == 8, "coarse Schwinger action");
      fill_sin(x);
      x->data[0] = 0.61274301029796319; x->data[1] = -0.4828159090156936;
      x->data[2] = -0.58943551973302177; x->data[3] = -1.1740400357754384;
      auto xc = std::make_shared<SampleState>(8);
      cact->copy_from_fine(x, xc);
      const double want_c[4] = {0.023307490564941413, -1.0268370199050634, -0.30193767594434939, 0.71119185749594449};
      for (int k = 0; k < 4; ++k) EXPECT(close(xc->data[k], want_c[k], 1e-14), "coarse link %d = %.17g", k, (double)xc->data[k]);
      act->copy_from_coarse(xc, x);
      EXPECT(close(x->data[0], 0.5 * want_c[0], 1e-14) && close(x->data[2], 0.5 * want_c[0], 1e-14), "copy_from_coarse halves the link");
    }
    auto rot = std::make_shared<Lattice2D>(4, 4, CoarsenRotate)->get_coarse_lattice();
    const unsigned want_rot[8] = {4, 6, 5, 7, 1, 1, 2, 2};
    EXPECT(rot && rot->is_rotated() && rot->getNvertices() == 8, "rotated coarse lattice");
    for (int k = 0; k < 8 && rot; ++k) EXPECT(rot->get_neighbour_vertices()[0][k] == want_rot[k], "rotated neighbour %d", k);
  }
  // ---- BASELINE config 1: HO, M_lat = 128, single-level HMC through the estimator loop ------------------
  {
    auto lat = std::make_shared<Lattice1D>(128, 4.0);
    auto act = std::make_shared<HarmonicOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0);
    act->set_seed(8923759);  // hmcsampler.hh:89
    HMCParameters hp;
    hp.nt = 100; hp.dt = 0.1; hp.n_burnin = 100; hp.n_rep = 1;
    hp.tune_iterations = 30; hp.tune_samples = 500;  // shorter than the reference's 100 x 1000: same bisection
    SingleLevelMCParameters mp;
    mp.n_burnin = 500; mp.n_samples = 20000;
    MonteCarloSingleLevel mc(act, std::make_shared<QoIXsquared>(lat), std::make_shared<HMCSamplerFactory>(hp), mp);
    auto hmc = std::dynamic_pointer_cast<HMCSampler>(mc.get_sampler());
    EXPECT(std::fabs(hmc->get_dt() - 0.0558) < 0.004, "tuned dt = %.4f (reference 0.0558)", hmc->get_dt());
    mc.evaluate();
    mc.show_statistics();
    auto st = mc.get_statistics();
    const double exact = act->Xsquared_analytical();
    std::printf(" analytic <x^2> = %.6f, numerical %.6f +- %.6f, p_accept %.4f\n", exact, st->average(), st->error(),
                mc.get_sampler()->p_accept());
    EXPECT(std::fabs(st->average() - exact) < 5 * st->error(), "HO <x^2> %.6f vs %.6f +- %.6f", st->average(), exact, st->error());
    EXPECT(std::fabs(mc.get_sampler()->p_accept() - 0.8) < 0.05, "p_accept %.4f", mc.get_sampler()->p_accept());
  }
  // ---- HO exact sampler (sampler = 'exact'): independent draws, tau_int = 1 ----------------------------------
  {
    auto lat = std::make_shared<Lattice1D>(128, 4.0);
    auto act = std::make_shared<HarmonicOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0);
    SingleLevelMCParameters mp;
    mp.n_burnin = 10; mp.n_samples = 20000;
    MonteCarloSingleLevel mc(act, std::make_shared<QoIXsquared>(lat), std::make_shared<ExactSamplerFactory>(), mp);
    mc.evaluate();
    auto st = mc.get_statistics();
    std::printf(" exact sampler: <x^2> = %.6f +- %.6f (analytic %.6f), tau_int %.3f\n", st->average(), st->error(),
                act->Xsquared_analytical(), st->tau_int());
    ZEXPECT(st->average(), st->error(), act->Xsquared_analytical(), 5, "exact sampler <x^2>");
    EXPECT(st->tau_int() < 1.2, "exact sampler draws are independent");
  }
  // ---- GFF 64 x 64: exact sampler (spectral synthesis) and heat-bath sampler started from an exact draw ------------
  {
    auto lat = std::make_shared<Lattice2D>(64, 64, CoarsenBoth);
    auto act = std::make_shared<GFFAction>(lat, nullptr, 10.0);
    const double exact = gff_phi_squared_analytical(10.0, 64, 64);
    SingleLevelMCParameters mp;
    mp.n_burnin = 10; mp.n_samples = 4000;
    MonteCarloSingleLevel mc(act, std::make_shared<QoI2DPhiSquared>(lat), std::make_shared<ExactSamplerFactory>(), mp);
    mc.evaluate();
    auto st = mc.get_statistics();
    std::printf(" GFF exact sampler: <phi^2> = %.6f +- %.6f (analytic %.6f), tau_int %.3f\n", st->average(), st->error(), exact,
                st->tau_int());
    ZEXPECT(st->average(), st->error(), exact, 5, "GFF exact sampler <phi^2>");
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 2; hb.n_burnin = 0;   // initialise_state is an exact draw: no burn-in needed
    mp.n_burnin = 0; mp.n_samples = 3000; mp.n_autocorr_window = 50;
    MonteCarloSingleLevel mh(act, std::make_shared<QoI2DPhiSquared>(lat), std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb), mp);
    mh.evaluate();
    auto sh = mh.get_statistics();
    std::printf(" GFF heat bath from an exact initial state: <phi^2> = %.6f +- %.6f\n", sh->average(), sh->error());
    ZEXPECT(sh->average(), std::fmax(sh->error(), 2e-3), exact, 5, "GFF heat bath <phi^2> without burn-in");
  }
  // ---- Schwinger 16x16: OverrelaxedHeatBathSampler through the estimator loop, batch of chains ---------
  {
    auto lat = std::make_shared<Lattice2D>(16, 16, CoarsenBoth);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 1; hb.n_burnin = 100;
    SingleLevelMCParameters mp;
    mp.n_burnin = 100; mp.n_samples = 20000; mp.n_autocorr_window = 100;
    MonteCarloSingleLevel mc(act, std::make_shared<QoIAvgPlaquette>(lat), std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb), mp);
    mc.evaluate();
    auto st = mc.get_statistics();
    std::printf(" plaquette %.6f +- %.6f (I1/I0 = 0.446390)\n", st->average(), st->error());
    ZEXPECT(st->average(), st->error(), 0.446390, 5, "plaquette");
  }
  // ---- multilevel Monte Carlo (BASELINE config 5 shape, small): HO M_lat = 64, 3 levels, hierarchical
  //      sampler with HMC on its coarsest level, Gaussian fill-in -------------------------------------------------
  {
    auto lat = std::make_shared<Lattice1D>(64, 4.0);
    auto act = std::make_shared<HarmonicOscillatorAction>(lat, RenormalisationNone, 1.0, 1.0);
    HMCParameters hp;
    hp.nt = 20; hp.dt = 0.15; hp.n_burnin = 50; hp.tune_iterations = 12; hp.tune_samples = 300;
    HierarchicalParameters hier;
    hier.n_max_level = 3; hier.n_meas = 50;
    auto cfa = std::make_shared<GaussianConditionedFineActionFactory>();
    auto hfac = std::make_shared<HierarchicalSamplerFactory>(std::make_shared<HMCSamplerFactory>(hp), cfa, hier);
    // single-level reference point through the hierarchical sampler itself
    {
      SingleLevelMCParameters mp;
      mp.n_burnin = 200; mp.n_samples = 3000;
      MonteCarloSingleLevel mc(act, std::make_shared<QoIXsquared>(lat), hfac, mp);
      mc.evaluate();
      auto st = mc.get_statistics();
      std::printf(" hierarchical sampler, single level: <x^2> = %.6f +- %.6f (analytic %.6f), p_accept %.3f\n", st->average(),
                  st->error(), act->Xsquared_analytical(), mc.get_sampler()->p_accept());
      ZEXPECT(st->average(), st->error(), act->Xsquared_analytical(), 5, "hierarchical sampler <x^2>");
    }
    {  // sampler/multilevelsampler.cc: independent samples handed up the hierarchy
      MultilevelSampler ms(act, std::make_shared<QoIXsquaredFactory>(), std::make_shared<HMCSamplerFactory>(hp), cfa, 20, hier);
      auto st = std::make_shared<SampleState>(64);
      Statistics q("Q", 20);
      for (int k = 0; k < 1500; ++k) {
        ms.draw(st);
        if (k >= 200) q.record_sample(QoIXsquared(lat).evaluate(st));
      }
      std::printf(" multilevel sampler: <x^2> = %.6f +- %.6f (analytic %.6f)\n", q.average(), q.error(), act->Xsquared_analytical());
      ZEXPECT(q.average(), q.error(), act->Xsquared_analytical(), 5, "multilevel sampler <x^2>");
    }
    MultiLevelMCParameters mlp;
    mlp.n_level = 3; mlp.n_burnin = 100; mlp.epsilon = 2.5e-2; mlp.n_min_samples_qoi = 200; mlp.n_meas = 50;
    std::printf(" constructing the multilevel estimator ...\n"); std::fflush(stdout);
    MonteCarloMultiLevel mlmc(act, std::make_shared<QoIXsquaredFactory>(), hfac, cfa, mlp);
    std::printf(" running the multilevel estimator ...\n"); std::fflush(stdout);
    mlmc.verbose = true;
    mlmc.evaluate();
    mlmc.show_statistics();
    const double exact = act->Xsquared_analytical();
    std::printf(" MLMC <x^2> = %.6f +- %.6f (analytic %.6f)\n", mlmc.numerical_result(), mlmc.statistical_error(), exact);
    ZEXPECT(mlmc.numerical_result(), mlmc.statistical_error(), exact, 5, "MLMC estimate");
    EXPECT(mlmc.level_statistics(0)->variance() < mlmc.level_statistics(2)->variance(), "variance decays towards fine levels");
    {  // montecarlo/montecarlotwolevel.cc: variance of the fine / coarse QoI and of their difference
      TwoLevelMCParameters tp;
      tp.n_burnin = 200; tp.n_samples = 3000; tp.n_meas = 50;
      MonteCarloTwoLevel two(act, std::make_shared<QoIXsquaredFactory>(), std::make_shared<HMCSamplerFactory>(hp), cfa, tp);
      two.evaluate_difference();
      const Statistics &f = two.fine_statistics(), &c = two.coarse_statistics(), &d = two.difference_statistics();
      std::printf(" two-level MC: Q_fine %.4f (var %.4f), Q_coarse %.4f (var %.4f), difference %.5f (var %.5f)\n", f.average(),
                  f.variance(), c.average(), c.variance(), d.average(), d.variance());
      ZEXPECT(f.average(), f.error(), exact, 5, "two-level MC fine average");
      EXPECT(d.variance() < 0.1 * f.variance(), "two-level MC: the difference has a much smaller variance");
      EXPECT(std::fabs(d.average() - (f.average() - c.average())) < 1e-12, "two-level MC: difference of averages");
    }
    // level sharding: two ranks in lockstep, rank r owns the levels l with l % 2 == r; what they
    // exchange per pass is the 3 x 5 table, summed element-wise (the all-reduce of a real two-process run)
    MultiLevelMCParameters p0 = mlp, p1 = mlp;
    p0.level_ranks = p1.level_ranks = 2;
    p0.level_rank = 0; p1.level_rank = 1;
    MonteCarloMultiLevel r0(act, std::make_shared<QoIXsquaredFactory>(), hfac, cfa, p0);
    MonteCarloMultiLevel r1(act, std::make_shared<QoIXsquaredFactory>(), hfac, cfa, p1);
    r0.begin(); r1.begin();
    bool done0, done1;
    int passes = 0;
    do {
      std::vector<double> t0 = r0.pass(), t1 = r1.pass();
      for (size_t k = 0; k < t0.size(); ++k) t0[k] += t1[k];
      done0 = r0.update(t0);
      done1 = r1.update(t0);
      ++passes;
    } while (!done0);
    std::printf(" level-sharded MLMC (2 ranks, %d passes): <x^2> = %.6f +- %.6f (analytic %.6f)\n", passes, r0.numerical_result(),
                r0.statistical_error(), exact);
    EXPECT(done0 == done1 && r0.numerical_result() == r1.numerical_result(), "ranks agree on the combined estimate");
    ZEXPECT(r0.numerical_result(), r0.statistical_error(), exact, 5, "level-sharded MLMC estimate");
    EXPECT(r0.owns(0) && !r0.owns(1) && r0.owns(2) && r1.owns(1), "level ownership");
  }
  // ---- rotor: hierarchical sampler (heat bath on the coarsest level, ExpSin2 fill-in) vs the direct sampler ---
  {
    auto lat = std::make_shared<Lattice1D>(32, 4.0);
    auto act = std::make_shared<RotorAction>(lat, RenormalisationPerturbative, 0.25);
    auto coarse = std::dynamic_pointer_cast<RotorAction>(act->coarse_action());
    std::printf(" rotor: m0 %.6f -> coarse %.6f; chi_t perturbative %.6f, continuum %.6f\n", act->getm0(), coarse->getm0(),
                act->chit_perturbative(), act->chit_continuum());
    EXPECT(coarse->getm0() > act->getm0() && coarse->getm0() < 1.5 * act->getm0(), "perturbative m0 renormalisation");
    OverrelaxedHeatBathParameters op;
    op.n_sweep_overrelax = 1; op.n_sweep_heatbath = 1; op.n_burnin = 100;
    HierarchicalParameters hier;
    hier.n_max_level = 3; hier.n_meas = 20;
    auto hfac = std::make_shared<HierarchicalSamplerFactory>(std::make_shared<OverrelaxedHeatBathSamplerFactory>(op),
                                                             std::make_shared<RotorConditionedFineActionFactory>(), hier);
    SingleLevelMCParameters mp;
    mp.n_burnin = 300; mp.n_samples = 20000;
    MonteCarloSingleLevel mc(act, std::make_shared<QoISusceptibility>(lat), hfac, mp);
    mc.evaluate();
    MonteCarloSingleLevel direct(act, std::make_shared<QoISusceptibility>(lat), std::make_shared<OverrelaxedHeatBathSamplerFactory>(op), mp);
    direct.evaluate();
    auto a = mc.get_statistics(), b = direct.get_statistics();
    std::printf(" rotor chi_t: hierarchical %.6f +- %.6f (p_accept %.3f), direct %.6f +- %.6f\n", a->average(), a->error(),
                mc.get_sampler()->p_accept(), b->average(), b->error());
    ZEXPECT(a->average(), std::hypot(a->error(), b->error()), b->average(), 5, "rotor hierarchical chi_t");
    // a/m0 = 0.5 here: the O(a) formula is only a rough guide
    // (the O(a) formula of rotoraction.cc:92-95 is only a rough guide at a/m0 = 0.5: printed, not gated; the gate above is the
    // combined-error comparison of the two samplers)
    std::printf(" rotor chi_t O(a) formula: %.6f\n", act->chit_perturbative());
  }
  // ---- Schwinger 16 x 16, beta = 2, CoarsenAlternate: hierarchical sampler (16x16 -> 8x16 -> 8x8) and a
  //      3-level multilevel estimate of the average plaquette (I1(2)/I0(2) = 0.697775) -----------------------------
  {
    auto lat = std::make_shared<Lattice2D>(16, 16, CoarsenAlternate);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 2.0);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 1; hb.n_burnin = 100;
    auto hbfac = std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb);
    auto cfa = std::make_shared<QuenchedSchwingerConditionedFineActionFactory>();
    HierarchicalParameters hier;
    // the sampler starts cold (all links 0); the reference thermalises it with its 10 000 timing draws
    hier.n_max_level = 3; hier.n_meas = 2000;
    auto hfac = std::make_shared<HierarchicalSamplerFactory>(hbfac, cfa, hier);
    const double exact = 0.697775;
    {
      SingleLevelMCParameters mp;
      mp.n_burnin = 500; mp.n_samples = 10000; mp.n_autocorr_window = 50;
      MonteCarloSingleLevel mc(act, std::make_shared<QoIAvgPlaquette>(lat), hfac, mp);
      mc.evaluate();
      auto st = mc.get_statistics();
      std::printf(" Schwinger hierarchical sampler (3 levels): plaquette %.6f +- %.6f (exact %.6f), p_accept %.3f, variance %.3e, tau_int %.3f\n",
                  st->average(), st->error(), exact, mc.get_sampler()->p_accept(), st->variance(), st->tau_int());
      mc.get_sampler()->show_stats();
      { auto c = st->auto_corr(); std::printf("  C[k]/C[0]:"); for (unsigned k = 0; k < c.size(); k += 3) std::printf(" %.3f", c[k] / c[0]); std::printf("\n"); }
      // (the chain is sticky at p_accept ~ 0.35: the windowed tau_int of a 10^4-sample run is noisy, so the
      //  tolerance has a floor; 3 x 10^5 samples of this chain give 0.69770 +- 0.00015)
      ZEXPECT(st->average(), std::fmax(st->error(), 5e-4), exact, 5, "Schwinger hierarchical plaquette");
      EXPECT(mc.get_sampler()->p_accept() > 0.01, "Schwinger hierarchical acceptance");
    }
    MultiLevelMCParameters mlp;
    mlp.n_level = 3; mlp.n_burnin = 100; mlp.epsilon = 4e-3; mlp.n_min_samples_qoi = 200; mlp.n_meas = 200;
    MonteCarloMultiLevel mlmc(act, std::make_shared<QoIAvgPlaquetteFactory>(), hfac, cfa, mlp);
    mlmc.evaluate();
    mlmc.show_statistics();
    std::printf(" Schwinger MLMC plaquette = %.6f +- %.6f (exact %.6f)\n", mlmc.numerical_result(), mlmc.statistical_error(), exact);
    ZEXPECT(mlmc.numerical_result(), mlmc.statistical_error(), exact, 5, "Schwinger MLMC estimate");
  }
  // ---- Schwinger 8 x 8, beta = 1.5, CoarsenBoth (the reference template's default): Bessel-product fill-in -------
  {
    auto lat = std::make_shared<Lattice2D>(8, 8, CoarsenBoth);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.5);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 1; hb.n_burnin = 100;
    HierarchicalParameters hier;
    hier.n_max_level = 2; hier.n_meas = 1000;
    auto hfac = std::make_shared<HierarchicalSamplerFactory>(std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb),
                                                             std::make_shared<QuenchedSchwingerConditionedFineActionFactory>(), hier);
    SingleLevelMCParameters mp;
    mp.n_burnin = 500; mp.n_samples = 20000; mp.n_autocorr_window = 50;
    MonteCarloSingleLevel mc(act, std::make_shared<QoIAvgPlaquette>(lat), hfac, mp);
    mc.evaluate();
    auto st = mc.get_statistics();
    const double exact = 0.596133;  // I1(1.5) / I0(1.5)
    std::printf(" Schwinger CoarsenBoth hierarchical sampler: plaquette %.6f +- %.6f (exact %.6f), p_accept %.3f\n", st->average(),
                st->error(), exact, mc.get_sampler()->p_accept());
    ZEXPECT(st->average(), std::fmax(st->error(), 5e-4), exact, 5, "Schwinger CoarsenBoth plaquette");
    EXPECT(mc.get_sampler()->p_accept() > 0.5, "Schwinger CoarsenBoth acceptance");
  }
  // ---- GFF 16 x 16, CoarsenRotate: hierarchical sampler over the rotated coarse level (n_gibbs_smooth = 2, its own exact
  //      sampler on the coarsest level, Gaussian fill-in of the odd vertices) and a 2-level multilevel estimate ----------
  {
    auto lat = std::make_shared<Lattice2D>(16, 16, CoarsenRotate);
    auto act = std::make_shared<GFFAction>(lat, nullptr, 10.0);
    const double exact = 0.33804823;  // gff_phi_squared_analytical(10, 16, 16)
    auto cfa = std::make_shared<GFFConditionedFineActionFactory>();
    // 2 levels: 16 x 16 -> rotated 16 x 16 (128 vertices), whose smoothed action is two Gibbs sweeps away from the exact
    // marginal: nearly every proposal is accepted.  3 levels (-> 8 x 8): the fill-in of the rotated level is exact for the
    // PLAIN stencil there, while that level's action is the smoothed one -- in the reference as here -- so its two-level
    // step rejects often; the chain is still exact (delayed acceptance), only slower.
    for (unsigned int n_levels = 2; n_levels <= 3; ++n_levels) {
      HierarchicalParameters hier;
      hier.n_max_level = n_levels; hier.n_meas = 20;
      auto hfac = std::make_shared<HierarchicalSamplerFactory>(std::make_shared<ExactSamplerFactory>(), cfa, hier);
      SingleLevelMCParameters mp;
      mp.n_burnin = 100; mp.n_samples = n_levels == 2 ? 8000 : 20000; mp.n_autocorr_window = 20;
      MonteCarloSingleLevel mc(act, std::make_shared<QoI2DPhiSquared>(lat), hfac, mp);
      mc.evaluate();
      auto st = mc.get_statistics();
      std::printf(" GFF CoarsenRotate hierarchical sampler (%u levels): <phi^2> %.6f +- %.6f (exact %.6f), tau_int %.2f, p_accept %.3f\n",
                  n_levels, st->average(), st->error(), exact, st->tau_int(), mc.get_sampler()->p_accept());
      ZEXPECT(st->average(), st->error(), exact, 5, "GFF hierarchical <phi^2>");
      if (n_levels == 2) EXPECT(mc.get_sampler()->p_accept() > 0.9, "GFF 2-level acceptance %.3f", mc.get_sampler()->p_accept());
      // 3 levels: REPORT ONLY.  The reference's own 3-level acceptance is unknown (no number of it exists anywhere in the
      // reference); the 8 % seen here equals 0.987 x the plain-vs-marginal independence test on the 128-vertex rotated
      // level computed in numpy from the library's matrices (0.070 +- 0.003, the equality is asserted where both numbers are
      // computed for the same lattice: tests/test_gff_levels.py::test_three_level_acceptance_is_the_plain_vs_marginal_mismatch).
      // An error in the rotated-level fill-in would show as a difference THERE; a window here would only freeze whatever
      // the acceptance happens to be.
      if (n_levels == 3) std::printf(" (report only) GFF 3-level acceptance %.3f\n", mc.get_sampler()->p_accept());
    }
    // coarse action: copy_from_fine then evaluate on the rotated level equals evaluating Qhat there (consistency of the
    // vertex maps between the classes)
    auto coarse = std::dynamic_pointer_cast<GFFAction>(act->coarse_action());
    auto fine_state = std::make_shared<SampleState>(act->sample_size()), coarse_state = std::make_shared<SampleState>(coarse->sample_size());
    fill_sin(fine_state);
    coarse->copy_from_fine(fine_state, coarse_state);
    auto back = std::make_shared<SampleState>(act->sample_size());
    act->copy_from_coarse(coarse_state, back);
    double worst = 0;
    for (unsigned i = 0; i < 16; ++i)
      for (unsigned j = 0; j < 16; ++j)
        if ((i + j) % 2 == 0) worst = std::fmax(worst, std::fabs(back->data[16 * j + i] - fine_state->data[16 * j + i]));
    EXPECT(worst == 0.0 && coarse->sample_size() == 128 && coarse->evaluate(coarse_state) > 0.0, "GFF level transfer round trip");
  }
  // ---- OverrelaxedHeatBathSampler::draw without a copy: same chain as the plain C-ABI sweeps, lent samples stay intact ----
  {
    auto lat = std::make_shared<Lattice2D>(64, 64, CoarsenBoth);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
    act->set_seed(77, 3);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 5; hb.n_burnin = 0; hb.batch = 2;
    OverrelaxedHeatBathSampler s(act, hb);
    // the same chain through the in-place C entry point
    const unsigned n = act->sample_size();
    auto ref = std::make_shared<SampleState>(n, 2), scr = std::make_shared<SampleState>(n, 2);
    act->initialise_state(ref);
    auto a = std::make_shared<SampleState>(n, 2), b = std::make_shared<SampleState>(n, 2), keep = std::make_shared<SampleState>(n, 2);
    double worst = 0, kept_drift = 0;
    std::vector<double> kept;
    for (int d = 0; d < 7; ++d) {
      auto &out = (d & 1) ? b : a;
      s.draw(out);
      check(mlmcpi_lattice_sweep_draw(&act->abi_action(), ref->device_mutable(), scr->device_mutable(), 2, 5, 1, 77, 3, 6 * d, 0, nullptr), "sweep");
      for (size_t l = 0; l < out->data.size(); l += 97) worst = std::fmax(worst, std::fabs(out->data[l] - ref->data[l]));
      if (d == 1) {  // hold on to the second sample while the sampler moves on
        keep->share(*out);
        kept.assign(keep->data.data(), keep->data.data() + keep->data.size());
      }
    }
    for (size_t l = 0; l < kept.size(); ++l) kept_drift = std::fmax(kept_drift, std::fabs(keep->data[l] - kept[l]));
    EXPECT(worst == 0.0, "zero-copy draw differs from the in-place sweeps by %g", worst);
    EXPECT(kept_drift == 0.0, "a sample lent to the caller was overwritten (%g)", kept_drift);
    EXPECT(s.pool_size() <= 4, "sampler buffer pool grew to %zu", s.pool_size());
    std::printf(" zero-copy OverrelaxedHeatBathSampler: 7 draws identical to in-place sweeps, lent sample intact, pool %zu buffers\n", s.pool_size());
  }
  // ---- draw + QoI in one pass (Sampler::draw_with_qoi): same numbers as QoI::evaluate_device on the sample it returns -------
  {
    auto lat = std::make_shared<Lattice2D>(128, 128, CoarsenBoth);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 5; hb.n_burnin = 0; hb.batch = 3;
    auto check_fused = [&](std::shared_ptr<Action> act, std::shared_ptr<QoI> qoi, const char *what) {
      OverrelaxedHeatBathSampler s(act, hb);
      auto x = std::make_shared<SampleState>(act->sample_size(), 3);
      DeviceVector dq(3), dref(3), acc(15);  // acc: record_sample in the same call (per-chain moments n, sum q, sum q^2, ...)
      check(mlmcpi_memset(acc.ptr(), 0, 15 * sizeof(double), nullptr), "memset");
      double worst = 0.0, sum_q[3] = {0, 0, 0};
      for (int d = 0; d < 3; ++d) {
        const bool fused = s.draw_with_qoi(x, qoi->fused_kind(), (double *)dq.ptr(), (double *)acc.ptr());
        EXPECT(fused, "%s: draw_with_qoi refused", what);
        qoi->evaluate_device(x, (double *)dref.ptr());
        const auto a = dq.download<double>(), b = dref.download<double>();
        for (int c = 0; c < 3; ++c) {
          worst = std::fmax(worst, std::fabs(a[c] - b[c]) / std::fmax(1.0, std::fabs(b[c])));
          sum_q[c] += a[c];
        }
      }
      EXPECT(worst < 1e-10, "%s: fused QoI differs from evaluate_device by %g", what, worst);
      const auto m = acc.download<double>();
      for (int c = 0; c < 3; ++c)
        EXPECT(m[5 * c] == 3.0 && m[5 * c + 1] == sum_q[c], "%s: moments recorded inside the draw: n = %g, sum = %.17g vs %.17g", what,
               m[5 * c], m[5 * c + 1], sum_q[c]);
      std::printf(" draw_with_qoi (%s): 3 draws x 3 chains, largest relative difference to evaluate_device %.1e\n", what, worst);
    };
    auto schw = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
    schw->set_seed(91, 0);
    check_fused(schw, std::make_shared<QoIAvgPlaquette>(lat), "Schwinger, average plaquette");
    check_fused(schw, std::make_shared<QoI2DSusceptibility>(lat), "Schwinger, topological susceptibility");
    auto gff = std::make_shared<GFFAction>(lat, nullptr, 10.0);
    gff->set_seed(92, 0);
    check_fused(gff, std::make_shared<QoI2DPhiSquared>(lat), "GFF, phi^2");
  }
  // ---- cross-rank statistics: 2 ranks (threads of this process, one GPU) through MonteCarloSingleLevel -------------------------
  {
    const int W = 2;
    auto hub = std::make_shared<ThreadExchangeHub>(W);
    std::vector<double> avg(W), err(W);
    std::vector<unsigned> total(W), local(W), passes(W);
    std::vector<std::thread> ranks;
    for (int r = 0; r < W; ++r)
      ranks.emplace_back([&, r] {
        auto lat = std::make_shared<Lattice2D>(16, 16, CoarsenBoth);
        auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
        act->set_seed(2481317, (uint32_t)r);  // chain index = rank: independent Philox streams
        OverrelaxedHeatBathParameters hb;
        hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 1; hb.n_burnin = 50;
        SingleLevelMCParameters mp;
        mp.n_burnin = 100; mp.n_samples = 0; mp.epsilon = 2e-3; mp.n_min_samples_qoi = 200; mp.n_autocorr_window = 20;
        MonteCarloSingleLevel mc(act, std::make_shared<QoIAvgPlaquette>(lat), std::make_shared<OverrelaxedHeatBathSamplerFactory>(hb), mp,
                                 std::make_shared<ThreadExchange>(hub, r));
        mc.evaluate();
        auto st = mc.get_statistics();
        StatsSync sync(*st);
        avg[r] = st->average(); err[r] = st->error(); total[r] = st->samples(); local[r] = st->local_samples(); passes[r] = mc.passes();
      });
    for (auto &t : ranks) t.join();
    const double exact = 0.446390;  // I1(1) / I0(1)
    std::printf(" 2-rank single-level MC (thread ranks, one all-reduce per pass): plaquette %.6f +- %.6f (exact %.6f), %u samples "
                "(%u + %u), %u passes\n", avg[0], err[0], exact, total[0], local[0], local[1], passes[0]);
    EXPECT(avg[0] == avg[1] && err[0] == err[1] && total[0] == total[1] && passes[0] == passes[1], "ranks disagree on the reduced statistics");
    EXPECT(total[0] == local[0] + local[1] && local[0] >= local[1] && local[0] - local[1] <= 1 + total[0] / 2, "sample split");
    EXPECT(std::fabs(avg[0] - exact) < 4 * err[0] && err[0] < 2.5e-3, "2-rank plaquette %.6f +- %.6f", avg[0], err[0]);
  }
  // ---- site-at-a-time updates (action/action.hh:73-96) and the random_order loop of overrelaxedheatbathsampler.cc:8-31 ------
  {
    auto lat = std::make_shared<Lattice2D>(8, 8, CoarsenBoth);
    auto act = std::make_shared<QuenchedSchwingerAction>(lat, nullptr, RenormalisationNone, 1.0);
    auto a = std::make_shared<SampleState>(act->sample_size(), 3), b = std::make_shared<SampleState>(act->sample_size(), 3),
         scratch = std::make_shared<SampleState>(act->sample_size(), 3);
    act->initialise_state(a);
    b->data = a->data;
    // one heat-bath sweep (Philox step 7) = the four colour classes visited link by link with the same step
    act->sweep(a, scratch, 0, 1, 7);
    for (int colour = 0; colour < 4; ++colour)
      for (unsigned l = 0; l < act->sample_size(); ++l) {
        const unsigned mu = l & 1, v = l >> 1, j = v / 8, i = v % 8;
        if ((int)(mu == 0 ? (j & 1) : 2 + (i & 1)) != colour) continue;
        act->set_site_step(7);
        act->heatbath_update(b, l);
      }
    double worst = 0.0;
    for (unsigned k = 0; k < 3 * act->sample_size(); ++k) worst = std::fmax(worst, std::fabs(std::remainder(a->data[k] - b->data[k], 2 * M_PI)));
    std::printf(" site-at-a-time heat bath over the four colour classes vs one device sweep: max difference %.2e\n", worst);
    EXPECT(worst < 1e-12, "per-site updates do not reproduce the sweep: %.3e", worst);
    // overrelaxation of a single link conserves the action and is an involution
    const double S0 = act->evaluate(b);
    const double before = b->data[37];
    act->overrelaxation_update(b, 37);
    EXPECT(std::fabs(act->evaluate(b) - S0) < 1e-10 && std::fabs(b->data[37] - before) > 1e-6, "overrelaxation_update(37)");
    act->overrelaxation_update(b, 37);
    EXPECT(std::fabs(std::remainder(b->data[37] - before, 2 * M_PI)) < 1e-12, "overrelaxation_update twice is the identity");
    // random_order = true: the reference's loop (shuffled index set, one local update per index); plaquette of 16 x 16, beta = 1
    auto lat16 = std::make_shared<Lattice2D>(16, 16, CoarsenBoth);
    auto act16 = std::make_shared<QuenchedSchwingerAction>(lat16, nullptr, RenormalisationNone, 1.0);
    OverrelaxedHeatBathParameters hb;
    hb.n_sweep_heatbath = 1; hb.n_sweep_overrelax = 1; hb.n_burnin = 30; hb.random_order = true; hb.batch = 64;
    OverrelaxedHeatBathSampler sampler(act16, hb);
    QoIAvgPlaquette qoi(lat16);
    auto st = std::make_shared<SampleState>(act16->sample_size(), 64);
    double sum = 0.0, sum2 = 0.0;
    const int n_draws = 60;
    std::vector<double> chain_mean(64, 0.0);
    for (int d = 0; d < n_draws; ++d) {
      sampler.draw(st);
      const std::vector<double> q = qoi.evaluate_batch(st);
      for (int c = 0; c < 64; ++c) chain_mean[c] += q[c] / n_draws;
    }
    for (double m : chain_mean) { sum += m; sum2 += m * m; }
    const double mean = sum / 64, err = std::sqrt((sum2 / 64 - mean * mean) / 63);
    std::printf(" random_order sampler (shuffled site-at-a-time sweeps, 64 chains x %d draws): plaquette %.5f +- %.5f (I1/I0 = 0.44639)\n",
                n_draws, mean, err);
    EXPECT(std::fabs(mean - 0.446390) < 4 * err + 1e-3 && err < 3e-3, "random_order plaquette %.5f +- %.5f", mean, err);
  }
  // ---- RcclExchange (one rank here: communicator set-up, the all-reduce itself and the tear-down on real RCCL) ------------
  {
    RcclExchange ex(0, 1, "/tmp/mlmcpi_test_host_id", 0);
    std::vector<double> v = {1.5, -2.0, 3.25};
    ex.allreduce_sum(v.data(), v.size());
    EXPECT(ex.rank() == 0 && ex.size() == 1 && v[0] == 1.5 && v[1] == -2.0 && v[2] == 3.25, "RcclExchange with one rank");
    Statistics st("Q", 5, std::make_shared<RcclExchange>(0, 1, "/tmp/mlmcpi_test_host_id2", 0));
    for (int i = 0; i < 50; ++i) st.record_sample(std::sin(0.3 * i));
    Statistics local("Q", 5);
    for (int i = 0; i < 50; ++i) local.record_sample(std::sin(0.3 * i));
    EXPECT(st.variance() == local.variance() && st.tau_int() == local.tau_int() && st.samples() == 50, "Statistics over RcclExchange");
    std::printf(" RcclExchange: ncclCommInitRank / ncclAllReduce / ncclCommDestroy through libmlmcpi_rccl.so OK\n");
  }
  std::printf(failures ? "%d FAILURES\n" : "host layer: all checks passed\n", failures);
  return failures ? 1 : 0;
}
