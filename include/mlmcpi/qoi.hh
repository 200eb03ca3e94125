// qoi.hh -- QoI interface (qoi/quantityofinterest.hh:16-36) and the five observables of the sweep
// path as device reductions.  evaluate() returns chain 0; evaluate_batch() all chains of the state.
#ifndef MLMCPI_QOI_HH
#define MLMCPI_QOI_HH
#include "action.hh"

namespace mlmcpi {

class QoI {
public:
  QoI() {}
  virtual ~QoI() {}
  const double virtual evaluate(const std::shared_ptr<SampleState> phi_state) { return evaluate_batch(phi_state)[0]; }
  virtual std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi_state) = 0;
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
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> x) override {
    if (x->size() != M_lat) fatal("Evaluating QoISusceptibility on path of wrong size.");
    DeviceVector out(x->batch());
    check(mlmcpi_qoi_xsquared(x->device(), M_lat, x->batch(), (double *)out.ptr(), nullptr), "qoi_xsquared");
    return out.download<double>();
  }
private:
  const unsigned int M_lat;
};

/** qoi/qm/qoisusceptibility.cc:8-23 */
class QoISusceptibility : public QoI {
public:
  explicit QoISusceptibility(const std::shared_ptr<Lattice1D> lattice) : M_lat(lattice->getM_lat()), T_final(lattice->getT_final()) {}
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> x) override {
    if (x->size() != M_lat) fatal("Evaluating QoISusceptibility on path of wrong size.");
    DeviceVector out(x->batch());
    check(mlmcpi_qoi_susceptibility(x->device(), M_lat, T_final, x->batch(), (double *)out.ptr(), nullptr), "qoi_susceptibility");
    return out.download<double>();
  }
private:
  const unsigned int M_lat;
  const double T_final;
};

/** qoi/qft/qoi2dsusceptibility.cc:8-27 */
class QoI2DSusceptibility : public QoI {
public:
  explicit QoI2DSusceptibility(const std::shared_ptr<Lattice2D> lattice) : Mt_lat(lattice->getMt_lat()), Mx_lat(lattice->getMx_lat()) {}
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi) override {
    if (phi->size() != 2 * Mt_lat * Mx_lat) fatal("Evaluating QoI2DSusceptibility on state of wrong size.");
    DeviceVector out(phi->batch());
    check(mlmcpi_qoi_2d_susceptibility(phi->device(), Mt_lat, Mx_lat, phi->batch(), (double *)out.ptr(), nullptr), "qoi_2d_susceptibility");
    return out.download<double>();
  }
private:
  const unsigned int Mt_lat, Mx_lat;
};

/** qoi/qft/qoiavgplaquette.cc:8-27 */
class QoIAvgPlaquette : public QoI {
public:
  explicit QoIAvgPlaquette(const std::shared_ptr<Lattice2D> lattice) : Mt_lat(lattice->getMt_lat()), Mx_lat(lattice->getMx_lat()) {}
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi) override {
    if (phi->size() != 2 * Mt_lat * Mx_lat) fatal("Evaluating QoIAvgPlaquette on state of wrong size.");
    DeviceVector out(phi->batch());
    check(mlmcpi_qoi_avg_plaquette(phi->device(), Mt_lat, Mx_lat, phi->batch(), (double *)out.ptr(), nullptr), "qoi_avg_plaquette");
    return out.download<double>();
  }
private:
  const unsigned int Mt_lat, Mx_lat;
};

/** qoi/qft/qoi2dphisquared.cc:8-15 */
class QoI2DPhiSquared : public QoI {
public:
  explicit QoI2DPhiSquared(const std::shared_ptr<Lattice2D> lattice) : M_lat(lattice->getNvertices()) {}
  std::vector<double> evaluate_batch(const std::shared_ptr<SampleState> phi) override {
    DeviceVector out(phi->batch());
    check(mlmcpi_qoi_phi_squared(phi->device(), M_lat, phi->batch(), (double *)out.ptr(), nullptr), "qoi_phi_squared");
    return out.download<double>();
  }
private:
  const unsigned int M_lat;
};

}  // namespace mlmcpi
#endif
