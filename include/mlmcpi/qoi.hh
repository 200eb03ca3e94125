// qoi.hh -- QoI interface (qoi/quantityofinterest.hh:16-36) and the five observables of the sweep
// path as device reductions.  evaluate() returns chain 0; evaluate_batch() all chains of the state;
// evaluate_device() leaves the per-chain values on the device (asynchronous: no host round trip per sample, for
// loops that accumulate statistics on the device with mlmcpi_stats_accumulate).
#ifndef MLMCPI_QOI_HH
#define MLMCPI_QOI_HH
#include "action.hh"

namespace mlmcpi {

class QoI {
public:
  QoI() {}
  virtual ~QoI() {}
  const double virtual evaluate(const std::shared_ptr<SampleState> phi_state) { return evaluate_batch(phi_state)[0]; }
  virtual std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi_state) {
    if (!out || out_n != phi_state->batch()) {
      out = std::make_unique<DeviceVector>(phi_state->batch());  // allocated once per batch size, not per sample
      out_n = phi_state->batch();
    }
    evaluate_device(phi_state, (double *)out->ptr());
    return out->download<double>();
  }
  /** d_out[b] = Q(chain b), enqueued on the default stream */
  virtual void evaluate_device(const std::shared_ptr<SampleState> phi_state, double *d_out) = 0;
  /** selector of mlmcpi_lattice_sweep_draw_qoi for QoIs a sweep sampler can sum inside its last launch (0: none) */
  virtual int fused_kind() const { return 0; }

private:
  std::unique_ptr<DeviceVector> out;
  unsigned int out_n = 0;
};

class QoIFactory {
public:
  virtual ~QoIFactory() {}
  virtual std::shared_ptr<QoI> get(std::shared_ptr<Action> action) = 0;
};

/** qoi/qm/qoixsquared.cc:7-20 (the size check and its message, verbatim quirk included) */
class QoIXsquared : public QoI {
public:
  explicit QoIXsquared(const std::shared_ptr<Lattice1D> lattice) : M_lat(lattice->getM_lat()) {}
  void evaluate_device(const std::shared_ptr<SampleState> x, double *d_out) override {
    if (x->size() != M_lat) fatal("Evaluating QoISusceptibility on path of wrong size.");
    check(mlmcpi_qoi_xsquared(x->device(), M_lat, x->batch(), d_out, nullptr), "qoi_xsquared");
  }
private:
  const unsigned int M_lat;
};

/** qoi/qm/qoisusceptibility.cc:8-23 */
class QoISusceptibility : public QoI {
public:
  explicit QoISusceptibility(const std::shared_ptr<Lattice1D> lattice) : M_lat(lattice->getM_lat()), T_final(lattice->getT_final()) {}
  void evaluate_device(const std::shared_ptr<SampleState> x, double *d_out) override {
    if (x->size() != M_lat) fatal("Evaluating QoISusceptibility on path of wrong size.");
    check(mlmcpi_qoi_susceptibility(x->device(), M_lat, T_final, x->batch(), d_out, nullptr), "qoi_susceptibility");
  }
  int fused_kind() const override { return 4; }  // mlmcpi_path_sweep_draw_qoi (rotor sweeps)
private:
  const unsigned int M_lat;
  const double T_final;
};

/** qoi/qft/qoi2dsusceptibility.cc:8-27 */
class QoI2DSusceptibility : public QoI {
public:
  explicit QoI2DSusceptibility(const std::shared_ptr<Lattice2D> lattice) : Mt_lat(lattice->getMt_lat()), Mx_lat(lattice->getMx_lat()) {}
  int fused_kind() const override { return 2; }
  void evaluate_device(const std::shared_ptr<SampleState> phi, double *d_out) override {
    if (phi->size() != 2 * Mt_lat * Mx_lat) fatal("Evaluating QoI2DSusceptibility on state of wrong size.");
    check(mlmcpi_qoi_2d_susceptibility(phi->device(), Mt_lat, Mx_lat, phi->batch(), d_out, nullptr), "qoi_2d_susceptibility");
  }
private:
  const unsigned int Mt_lat, Mx_lat;
};

/** qoi/qft/qoiavgplaquette.cc:8-27 */
class QoIAvgPlaquette : public QoI {
public:
  explicit QoIAvgPlaquette(const std::shared_ptr<Lattice2D> lattice) : Mt_lat(lattice->getMt_lat()), Mx_lat(lattice->getMx_lat()) {}
  int fused_kind() const override { return 1; }
  void evaluate_device(const std::shared_ptr<SampleState> phi, double *d_out) override {
    if (phi->size() != 2 * Mt_lat * Mx_lat) fatal("Evaluating QoIAvgPlaquette on state of wrong size.");
    check(mlmcpi_qoi_avg_plaquette(phi->device(), Mt_lat, Mx_lat, phi->batch(), d_out, nullptr), "qoi_avg_plaquette");
  }
private:
  const unsigned int Mt_lat, Mx_lat;
};

/** qoi/qft/qoi2dphisquared.cc:8-15 */
class QoI2DPhiSquared : public QoI {
public:
  explicit QoI2DPhiSquared(const std::shared_ptr<Lattice2D> lattice) : M_lat(lattice->getNvertices()) {}
  int fused_kind() const override { return 3; }
  void evaluate_device(const std::shared_ptr<SampleState> phi, double *d_out) override {
    check(mlmcpi_qoi_phi_squared(phi->device(), M_lat, phi->batch(), d_out, nullptr), "qoi_phi_squared");
  }
private:
  const unsigned int M_lat;
};

}  // namespace mlmcpi
#endif
