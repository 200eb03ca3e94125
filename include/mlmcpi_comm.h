/* mlmcpi_comm.h -- C ABI of the one collective on the path: the sum of a small packed fp64 statistics buffer over the
 * ranks of a node (one process per GPU), on RCCL over xGMI.
 *
 * Replaces the scalar MPI_Allreduce calls behind Statistics::variance / tau_int / samples / average
 * (reference src/common/statistics.cc:29-95 via src/mpi/mpi_wrapper.cc:44-120), the logical-AND termination test of
 * MonteCarloSingleLevel::evaluate (src/montecarlo/montecarlosinglelevel.cc:84-86) and the per-level exchange of
 * MonteCarloMultiLevel (src/montecarlo/montecarlomultilevel.cc:115-165): one ncclAllReduce(sum) of ~10^2 doubles per
 * convergence check instead of ~10 scalar round trips.  The C++ classes reach it through mlmcpi::RcclExchange
 * (include/mlmcpi/exchange.hh); bench.py's N > 1 runs call the same entry points through ctypes.
 *
 * Library: mlmcpathintegral_amd/libmlmcpi_rccl.so.  It does NOT link librccl: the RCCL runtime is opened at
 * mlmcpi_comm_load() time (dlopen), so a process that already carries an RCCL (a PyTorch process) can hand in that very
 * library and never holds two copies.  Conventions as in mlmcpi_hip.h: 0 on success, negative mlmcpi_status otherwise,
 * message via mlmcpi_comm_last_error(); d_* are device pointers; `stream` is a hipStream_t passed as void*.
 */
#ifndef MLMCPI_COMM_H
#define MLMCPI_COMM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLMCPI_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct mlmcpi_comm mlmcpi_comm;

const char *mlmcpi_comm_last_error(void);

/* Open the RCCL runtime.  path = NULL: $MLMCPI_RCCL_LIB, else librccl.so.1 on the loader path, else
 * /opt/rocm/lib/librccl.so.1.  Idempotent; the other entry points call it with NULL when it has not been called. */
int mlmcpi_comm_load(const char *path);

/* File the RCCL runtime in use was mapped from (dladdr of its ncclAllReduce); "" before mlmcpi_comm_load.  For run records. */
const char *mlmcpi_comm_runtime(void);

/* rank 0: a fresh 128-byte rendezvous id (ncclGetUniqueId), to be handed to every rank by whatever channel the host
 * program has (MPI_Bcast in the reference's tree, torch.distributed in bench.py, a file: mlmcpi_comm_init_file) */
int mlmcpi_comm_unique_id(void *id128);

/* Join the communicator of `world` ranks on HIP device `device` (ncclCommInitRank).  Collective over all ranks. */
int mlmcpi_comm_init(int rank, int world, const void *id128, int device, mlmcpi_comm **out);

/* The same with a file as the rendezvous channel: rank 0 removes whatever lies at `path`, then writes the id there
 * (atomically) together with its own pid and process start time; the others wait up to timeout_s seconds for a record
 * whose writer is still alive -- a file left behind by a run that died is never accepted.  `path` must be on a file
 * system all ranks of the node see (/dev/shm) and should be unique per launch (host/driver puts the launcher's pid in
 * the name).  For C++ hosts started as N processes without MPI (host/driver). */
int mlmcpi_comm_init_file(int rank, int world, const char *path, int device, double timeout_s, mlmcpi_comm **out);

/* rank and number of ranks as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount) -- not an echo of the
 * arguments of mlmcpi_comm_init: a caller can check that the group that formed is the one it asked for */
int mlmcpi_comm_rank(const mlmcpi_comm *c, int *rank);
int mlmcpi_comm_size(const mlmcpi_comm *c, int *size);

/* In-place sum over the ranks of n doubles in device memory, enqueued on `stream` (ncclAllReduce, ncclFloat64, ncclSum) */
int mlmcpi_comm_allreduce_sum_f64(mlmcpi_comm *c, double *d_buf, size_t n, void *stream);

/* The same for a HOST buffer: staged through a device buffer the communicator owns; returns when the sum is in h_buf.
 * This is what the Statistics / MonteCarlo classes call (their packed buffers live on the host). */
int mlmcpi_comm_allreduce_sum_host_f64(mlmcpi_comm *c, double *h_buf, size_t n);

int mlmcpi_comm_destroy(mlmcpi_comm *c);

#ifdef __cplusplus
}
#endif
#endif
