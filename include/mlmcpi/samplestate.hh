// samplestate.hh -- host-side mirror of the reference's SampleState (common/samplestate.hh:19-53)
// backed by a device-resident batch of chains.
//
// The reference's state is an Eigen::VectorXd `data` that every caller indexes directly.  Here
// `data` is a small proxy with the same surface the sweep path uses (operator[], size(), data(),
// assignment): the authoritative copy lives in HBM and is mirrored to the host lazily, so that
// Sampler::draw / QoI::evaluate / Action::evaluate never move the state over PCIe unless host code
// actually reads or writes elements.  `batch` > 1 holds that many independent chains, chain-major.
//
// The device copy is a reference-counted buffer.  `a->data = b->data` copies, as in the reference; `a->share(*b)` makes
// `a` a second holder of b's buffer instead (copy on write).  The sweep sampler ends its draw with the latter: the
// reference's `phi_state->data = phi_state_cur->data` (overrelaxedheatbathsampler.cc:30) would be a full pass over HBM
// per sample -- as much traffic as two fused sweeps.  A state that is about to be modified in place while someone else
// still holds its buffer is detached first (device_mutable), so value semantics are kept.
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

/** reference-counted device allocation */
struct DeviceBuffer {
  explicit DeviceBuffer(size_t bytes_) : bytes(bytes_) { check(mlmcpi_malloc((void **)&p, bytes), "mlmcpi_malloc"); }
  ~DeviceBuffer() { mlmcpi_free(p); }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  double *p = nullptr;
  const size_t bytes;
};

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
        if (owner->dev != other.owner->dev)
          check(mlmcpi_copy_d2d(owner->device_overwrite(), other.owner->dev->p, size() * sizeof(double), nullptr), "copy_d2d");
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
    dev = std::make_shared<DeviceBuffer>(bytes());
    check(mlmcpi_memset(dev->p, 0, bytes(), nullptr), "mlmcpi_memset");
    host_valid = device_valid = true;
  }
  SampleState(const SampleState &) = delete;
  SampleState &operator=(const SampleState &) = delete;

  /** Device pointer for reading (uploads pending host writes). */
  const double *device() const {
    const_cast<SampleState *>(this)->to_device();
    return dev->p;
  }
  /** Device pointer for kernels that modify the state in place: a shared buffer is detached (copied) first. */
  double *device_mutable() {
    to_device();
    if (dev.use_count() > 1) {
      std::shared_ptr<DeviceBuffer> mine = std::make_shared<DeviceBuffer>(bytes());
      check(mlmcpi_copy_d2d(mine->p, dev->p, bytes(), nullptr), "copy_d2d");
      dev = mine;
    }
    host_valid = false;
    return dev->p;
  }
  /** Device pointer for kernels that overwrite the whole state: a shared buffer is dropped, not copied. */
  double *device_overwrite() {
    if (dev.use_count() > 1) dev = std::make_shared<DeviceBuffer>(bytes());
    host_valid = false;
    device_valid = true;
    return dev->p;
  }
  /** Exchange device buffers with a state of equal shape (ping-pong sweeps end in the scratch). */
  void swap_device(SampleState &other) {
    if (M != other.M || B != other.B) fatal("SampleState shape mismatch in swap_device");
    to_device();
    other.to_device();
    std::swap(dev, other.dev);
    host_valid = other.host_valid = false;
  }
  /** Become a second holder of `other`'s device buffer (no copy).  Both states keep value semantics: whoever modifies
   *  its state in place while the buffer is shared gets a private copy first (device_mutable). */
  void share(SampleState &other) {
    if (M != other.M || B != other.B) fatal("SampleState shape mismatch in share");
    other.to_device();
    dev = other.dev;
    device_valid = true;
    host_valid = false;
  }
  /** the buffer itself, for samplers that rotate a small pool of buffers instead of copying (sampler.hh) */
  std::shared_ptr<DeviceBuffer> buffer() {
    to_device();
    return dev;
  }
  void adopt(std::shared_ptr<DeviceBuffer> b) {
    if (b->bytes != bytes()) fatal("SampleState shape mismatch in adopt");
    dev = b;
    device_valid = true;
    host_valid = false;
  }
  bool device_shared() const { return dev.use_count() > 1; }
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
      check(mlmcpi_copy_d2h(host.data(), dev->p, bytes(), nullptr), "copy_d2h");
      host_valid = true;
    }
    if (will_write) device_valid = false;
  }
  void to_device() {
    if (!device_valid) {
      if (dev.use_count() > 1) dev = std::make_shared<DeviceBuffer>(bytes());  // never upload into a buffer someone else reads
      check(mlmcpi_copy_h2d(dev->p, host.data(), bytes(), nullptr), "copy_h2d");
      device_valid = true;
    }
  }
  const unsigned int M, B;
  std::vector<double> host;
  std::shared_ptr<DeviceBuffer> dev;
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
