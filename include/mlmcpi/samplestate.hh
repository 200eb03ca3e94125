// samplestate.hh -- host-side mirror of the reference's SampleState (common/samplestate.hh:19-53)
// backed by a device-resident batch of chains.
//
// The reference's state is an Eigen::VectorXd `data` that every caller indexes directly.  Here
// `data` is a small proxy with the same surface the sweep path uses (operator[], size(), data(),
// assignment): the authoritative copy lives in HBM and is mirrored to the host lazily, so that
// Sampler::draw / QoI::evaluate / Action::evaluate never move the state over PCIe unless host code
// actually reads or writes elements.  `batch` > 1 holds that many independent chains, chain-major.
#ifndef MLMCPI_SAMPLESTATE_HH
#define MLMCPI_SAMPLESTATE_HH
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../mlmcpi_hip.h"

namespace mlmcpi {

/** Error convention of the reference (mpi/mpi_wrapper.cc:174-177): message on cerr, then exit. */
[[noreturn]] inline void fatal(const std::string &msg) {
  std::cerr << "ERROR: " << msg << std::endl;
  std::exit(EXIT_FAILURE);
}

inline void check(int status, const char *what) {
  if (status != MLMCPI_OK) fatal(std::string(what) + ": " + mlmcpi_last_error());
}

class SampleState {
public:
  /** Host view with lazy synchronisation; mirrors the uses of Eigen::VectorXd on the sweep path. */
  class Data {
  public:
    double &operator[](size_t j) { owner->to_host(true); return owner->host[j]; }
    double operator[](size_t j) const { owner->to_host(false); return owner->host[j]; }
    size_t size() const { return owner->host.size(); }
    double *data() { owner->to_host(true); return owner->host.data(); }
    const double *data() const { owner->to_host(false); return owner->host.data(); }
    /** phi_state->data = other->data (hmcsampler.cc:17,31,66): device-to-device when possible */
    Data &operator=(const Data &other) {
      if (this == &other) return *this;
      if (size() != other.size()) fatal("SampleState size mismatch in assignment");
      if (other.owner->device_valid) {
        check(mlmcpi_copy_d2d(owner->dev, other.owner->dev, size() * sizeof(double), nullptr), "copy_d2d");
        owner->device_valid = true;
        owner->host_valid = false;
      } else {
        owner->host = other.owner->host;
        owner->host_valid = true;
        owner->device_valid = false;
      }
      return *this;
    }
    double squaredNorm() const {
      owner->to_host(false);
      double s = 0.0;
      for (double v : owner->host) s += v * v;
      return s;
    }

  private:
    friend class SampleState;
    SampleState *owner = nullptr;
  };

  /** M entries per chain, zero-initialised (samplestate.hh:23-26); `batch` chains. */
  explicit SampleState(const unsigned int M_, const unsigned int batch_ = 1)
      : M(M_), B(batch_), host((size_t)M_ * batch_, 0.0) {
    data.owner = this;
    check(mlmcpi_malloc((void **)&dev, bytes()), "mlmcpi_malloc");
    check(mlmcpi_memset(dev, 0, bytes(), nullptr), "mlmcpi_memset");
    host_valid = device_valid = true;
  }
  ~SampleState() { mlmcpi_free(dev); }
  SampleState(const SampleState &) = delete;
  SampleState &operator=(const SampleState &) = delete;

  /** Device pointer for reading (uploads pending host writes). */
  const double *device() const {
    const_cast<SampleState *>(this)->to_device();
    return dev;
  }
  /** Device pointer for kernels that modify the state. */
  double *device_mutable() {
    to_device();
    host_valid = false;
    return dev;
  }
  /** Exchange device buffers with a state of equal shape (ping-pong sweeps end in the scratch). */
  void swap_device(SampleState &other) {
    if (M != other.M || B != other.B) fatal("SampleState shape mismatch in swap_device");
    to_device();
    other.to_device();
    std::swap(dev, other.dev);
    host_valid = other.host_valid = false;
  }
  unsigned int size() const { return M; }
  unsigned int batch() const { return B; }
  size_t bytes() const { return host.size() * sizeof(double); }

  /** samplestate.cc:7-16: text dump, one chain per file section */
  void save_to_disk(const std::string filename) {
    to_host(false);
    std::ofstream file(filename.c_str());
    file << M << std::endl;
    for (size_t j = 0; j < host.size(); ++j) file << host[j] << " ";
    file << std::endl;
  }

  Data data;

private:
  void to_host(bool will_write) {
    if (!host_valid) {
      check(mlmcpi_copy_d2h(host.data(), dev, bytes(), nullptr), "copy_d2h");
      host_valid = true;
    }
    if (will_write) device_valid = false;
  }
  void to_device() {
    if (!device_valid) {
      check(mlmcpi_copy_h2d(dev, host.data(), bytes(), nullptr), "copy_h2d");
      device_valid = true;
    }
  }
  const unsigned int M, B;
  std::vector<double> host;
  double *dev = nullptr;
  bool host_valid = false, device_valid = false;
};

/** [batch] doubles on the device, for per-chain results (actions, QoIs, energies). */
class DeviceVector {
public:
  explicit DeviceVector(size_t n, size_t elem = sizeof(double)) : n_(n), elem_(elem) {
    check(mlmcpi_malloc(&p_, n * elem), "mlmcpi_malloc");
    check(mlmcpi_memset(p_, 0, n * elem, nullptr), "mlmcpi_memset");
  }
  ~DeviceVector() { mlmcpi_free(p_); }
  DeviceVector(const DeviceVector &) = delete;
  DeviceVector &operator=(const DeviceVector &) = delete;
  void *ptr() { return p_; }
  template <class T>
  std::vector<T> download() const {
    std::vector<T> h(n_);
    check(mlmcpi_copy_d2h(h.data(), p_, n_ * sizeof(T), nullptr), "copy_d2h");
    return h;
  }

private:
  void *p_ = nullptr;
  size_t n_, elem_;
};

}  // namespace mlmcpi
#endif
