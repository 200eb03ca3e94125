// exchange.hh -- the one collective of the path as an interface: element-wise sum of a small fp64 buffer over ranks.
//
// The reference combines statistics with scalar MPI_Allreduce calls (common/statistics.cc:29-95 through
// mpi/mpi_wrapper.cc:44-120) and ends its sampling loops on an all-reduced logical AND
// (montecarlo/montecarlosinglelevel.cc:84-86).  Here Statistics, MonteCarloSingleLevel and MonteCarloMultiLevel pack
// what they need into ONE buffer and call Exchange::allreduce_sum once per convergence check.  Implementations:
//   LocalExchange    one rank, no-op (the reference's build without USE_MPI)
//   RcclExchange     ncclAllReduce over xGMI through the C ABI of include/mlmcpi_comm.h (one process per GPU)
//   ThreadExchange   N ranks as threads of one process (tests on a machine without N GPUs)
// An MPI build supplies the same three lines around MPI_Allreduce.
#ifndef MLMCPI_EXCHANGE_HH
#define MLMCPI_EXCHANGE_HH
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../mlmcpi_comm.h"

namespace mlmcpi {

class Exchange {
public:
  virtual ~Exchange() {}
  virtual int rank() const = 0;
  virtual int size() const = 0;
  /** in place: buf[i] <- sum over ranks of buf[i]; collective (every rank calls it with the same n) */
  virtual void allreduce_sum(double *buf, size_t n) = 0;
};

class LocalExchange : public Exchange {
public:
  int rank() const override { return 0; }
  int size() const override { return 1; }
  void allreduce_sum(double *, size_t) override {}
};

/** mpi/mpi_wrapper.cc:187-202: share of `n` that falls to `rank` (remainder to the low ranks) */
inline unsigned int distribute_n(const unsigned int n, const int rank, const int size) {
  const unsigned int n_floor = n / (unsigned int)size, n_overflow = n - n_floor * (unsigned int)size;
  return (unsigned int)rank < n_overflow ? n_floor + 1 : n_floor;
}

/** RCCL over xGMI.  One process per GPU; the 128-byte rendezvous id travels by file (`id_path`, e.g. on /dev/shm) or is
 *  handed in by a host program that has its own channel. */
class RcclExchange : public Exchange {
public:
  RcclExchange(int rank_, int size_, const char *id_path, int device, double timeout_s = 120.0)
      : RcclExchange(rank_, size_, std::string(id_path), device, timeout_s) {}
  RcclExchange(int rank_, int size_, const std::string &id_path, int device, double timeout_s = 120.0) : comm(nullptr) {
    if (mlmcpi_comm_init_file(rank_, size_, id_path.c_str(), device, timeout_s, &comm)) die("mlmcpi_comm_init_file");
  }
  /** the rendezvous id itself (MLMCPI_COMM_ID_BYTES bytes from mlmcpi_comm_unique_id on rank 0, distributed by the
   *  caller).  A struct, not a pointer: a string literal must never be mistaken for an id. */
  struct Id { unsigned char bytes[MLMCPI_COMM_ID_BYTES]; };
  RcclExchange(int rank_, int size_, const Id &id, int device) : comm(nullptr) {
    if (mlmcpi_comm_init(rank_, size_, id.bytes, device, &comm)) die("mlmcpi_comm_init");
  }
  ~RcclExchange() override { mlmcpi_comm_destroy(comm); }
  int rank() const override { int r = 0; mlmcpi_comm_rank(comm, &r); return r; }
  int size() const override { int s = 1; mlmcpi_comm_size(comm, &s); return s; }
  void allreduce_sum(double *buf, size_t n) override {
    if (mlmcpi_comm_allreduce_sum_host_f64(comm, buf, n)) die("mlmcpi_comm_allreduce_sum_host_f64");
  }
  /** Proof that the group that formed is the one asked for, before any result depends on it: the communicator's own
   *  count (ncclCommCount) equals `expected` and one all-reduce of (1, rank) returns (N, N (N - 1) / 2).  Otherwise
   *  message + exit, like every other failure of this class: a run either has its N ranks or a non-zero status. */
  void verify(int expected) {
    const int n = size(), r = rank();
    double probe[2] = {1.0, (double)r};
    allreduce_sum(probe, 2);
    if (n != expected || probe[0] != (double)expected || probe[1] != 0.5 * expected * (expected - 1.0)) {
      std::fprintf(stderr, "ERROR: RcclExchange: asked for %d ranks, communicator reports %d, sum of ones %g, sum of ranks %g\n",
                   expected, n, probe[0], probe[1]);
      std::exit(EXIT_FAILURE);
    }
  }

private:
  static void die(const char *what) {  // the reference's convention: message + exit (mpi/mpi_wrapper.cc:174-177)
    std::fprintf(stderr, "ERROR: %s: %s\n", what, mlmcpi_comm_last_error());
    std::exit(EXIT_FAILURE);
  }
  mlmcpi_comm *comm;
};

/** N ranks as N threads of one process: a generation-counting barrier around per-rank contributions, summed in rank
 *  order by the last arrival (deterministic, like a fixed reduction tree).  For tests. */
class ThreadExchangeHub {
public:
  explicit ThreadExchangeHub(int size_) : size(size_), parts(size_), arrived(0), generation(0) {}
  void allreduce_sum(int rank, double *buf, size_t n) {
    std::unique_lock<std::mutex> lock(m);
    parts[rank].assign(buf, buf + n);
    const unsigned long gen = generation;
    if (++arrived == size) {
      result.assign(n, 0.0);
      for (int r = 0; r < size; ++r)
        for (size_t i = 0; i < n; ++i) result[i] += parts[r][i];
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != gen; });
    }
    // `result` stays valid until the next reduction completes, which needs this thread to arrive again
    for (size_t i = 0; i < n; ++i) buf[i] = result[i];
  }
  const int size;

private:
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::vector<double>> parts;
  std::vector<double> result;
  int arrived;
  unsigned long generation;
};

class ThreadExchange : public Exchange {
public:
  ThreadExchange(std::shared_ptr<ThreadExchangeHub> hub_, int rank_) : hub(hub_), my_rank(rank_) {}
  int rank() const override { return my_rank; }
  int size() const override { return hub->size; }
  void allreduce_sum(double *buf, size_t n) override { hub->allreduce_sum(my_rank, buf, n); }

private:
  std::shared_ptr<ThreadExchangeHub> hub;
  const int my_rank;
};

}  // namespace mlmcpi
#endif
