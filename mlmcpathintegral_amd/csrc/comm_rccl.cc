// comm_rccl.cc -- libmlmcpi_rccl.so: the statistics all-reduce on RCCL (include/mlmcpi_comm.h).
// The RCCL runtime is opened with dlopen; only its public header is used at build time.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE  // dladdr
#endif
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/mlmcpi_comm.h"

static_assert(MLMCPI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rendezvous id size");

namespace {

enum { OK = 0, ERR_INVALID = -1, ERR_HIP = -2, ERR_UNSUPPORTED = -3, ERR_NO_DEVICE = -4 };  // mlmcpi_status

thread_local std::string g_err;
int fail(int status, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return status;
}

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
std::mutex g_load_mutex;

int load(const char *path) {
  std::lock_guard<std::mutex> lock(g_load_mutex);
  if (g_rccl.handle) return OK;
  const char *env = getenv("MLMCPI_RCCL_LIB");
  const char *candidates[] = {path, env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void *h = nullptr;
  std::string tried;
  for (const char *c : candidates) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
    tried += std::string(c) + ": " + dlerror() + "; ";
  }
  if (!h) return fail(ERR_UNSUPPORTED, "cannot open the RCCL runtime (%s)", tried.c_str());
#define SYM(field, name)                                                          \
  *(void **)(&g_rccl.field) = dlsym(h, name);                                     \
  if (!g_rccl.field) { dlclose(h); return fail(ERR_UNSUPPORTED, "RCCL runtime lacks %s", name); }
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(AllReduce, "ncclAllReduce");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(CommUserRank, "ncclCommUserRank");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return OK;
}

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) return fail(ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));       \
  } while (0)
#define NCCL_TRY(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t r__ = (expr);                                                                  \
    if (r__ != ncclSuccess) return fail(ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r__));  \
  } while (0)

}  // namespace

struct mlmcpi_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;  // for the host-buffer form
  double *d_stage = nullptr;     // device staging buffer
  double *h_pinned = nullptr;    // pinned host mirror
  size_t cap = 0;                // doubles
};

extern "C" {

const char *mlmcpi_comm_last_error(void) { return g_err.c_str(); }

int mlmcpi_comm_load(const char *path) { return load(path); }

const char *mlmcpi_comm_runtime(void) {
  static thread_local std::string where;
  where.clear();
  Dl_info info;
  if (g_rccl.handle && g_rccl.AllReduce && dladdr((void *)g_rccl.AllReduce, &info) && info.dli_fname) where = info.dli_fname;
  return where.c_str();
}

int mlmcpi_comm_unique_id(void *id128) {
  if (!id128) return fail(ERR_INVALID, "id128 is NULL");
  if (int rc = load(nullptr)) return rc;
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return OK;
}

int mlmcpi_comm_init(int rank, int world, const void *id128, int device, mlmcpi_comm **out) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(ERR_INVALID, "bad arguments (rank %d of %d)", rank, world);
  if (int rc = load(nullptr)) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(ERR_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return fail(ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  mlmcpi_comm *c = new mlmcpi_comm;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    g_rccl.CommDestroy(c->comm);
    delete c;
    return fail(ERR_HIP, "hipStreamCreate failed");
  }
  *out = c;
  return OK;
}

// The rendezvous file.  A run that dies before rank 0 removes its file leaves it behind, and the next run's ranks > 0
// would read the old id at once and sit in ncclCommInitRank with mismatched ids for ever.  So the file
// names its writer -- pid and the start time of that pid (/proc/<pid>/stat, field 22) -- and a reader accepts it only
// while that very process is alive: a file left by a dead run is ignored until the live rank 0 replaces it (rename is
// atomic).  Rank 0 also removes whatever it finds at `path` before it writes.  Same node by construction (one process
// per GPU of one node, /dev/shm or any local path).
namespace {
struct IdFile {
  char magic[8];               // "MLMCPI1\0"
  int64_t pid;                 // the writer (rank 0)
  uint64_t start_ticks;        // its start time in clock ticks since boot
  char id[MLMCPI_COMM_ID_BYTES];
};
const char kIdMagic[8] = {'M', 'L', 'M', 'C', 'P', 'I', '1', 0};

// start time of process `pid` (0: no such process)
uint64_t process_start_ticks(int64_t pid) {
  char name[64], buf[1024];
  snprintf(name, sizeof name, "/proc/%lld/stat", (long long)pid);
  FILE *f = fopen(name, "r");
  if (!f) return 0;
  const size_t got = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[got] = 0;
  char *p = strrchr(buf, ')');  // the command name (field 2) may contain anything; the fields resume after its ')'
  if (!p) return 0;
  unsigned long long ticks = 0;
  int field = 2;
  for (char *tok = strtok(p + 1, " "); tok; tok = strtok(nullptr, " "))
    if (++field == 22) {  // starttime
      if (sscanf(tok, "%llu", &ticks) != 1) return 0;
      break;
    }
  if (field != 22) return 0;
  return ticks ? ticks : 1;
}
}  // namespace

int mlmcpi_comm_init_file(int rank, int world, const char *path, int device, double timeout_s, mlmcpi_comm **out) {
  if (!path || !*path) return fail(ERR_INVALID, "rendezvous path is empty");
  IdFile rec;
  if (rank == 0) {
    unlink(path);  // whatever an earlier run left here
    memset(&rec, 0, sizeof rec);
    memcpy(rec.magic, kIdMagic, sizeof rec.magic);
    rec.pid = (int64_t)getpid();
    rec.start_ticks = process_start_ticks(rec.pid);
    if (int rc = mlmcpi_comm_unique_id(rec.id)) return rc;
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)rec.pid);
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(&rec, 1, sizeof rec, f) != sizeof rec) {
      if (f) fclose(f);
      return fail(ERR_INVALID, "cannot write the rendezvous id to %s", tmp.c_str());
    }
    fclose(f);
    if (rename(tmp.c_str(), path) != 0) return fail(ERR_INVALID, "cannot rename %s to %s", tmp.c_str(), path);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    std::string why = "no file";
    for (;;) {
      FILE *f = fopen(path, "rb");
      if (f) {
        const size_t got = fread(&rec, 1, sizeof rec, f);
        fclose(f);
        if (got != sizeof rec || memcmp(rec.magic, kIdMagic, sizeof rec.magic) != 0) why = "not a rendezvous record";
        else if (rec.start_ticks == 0 || process_start_ticks(rec.pid) != rec.start_ticks) why = "stale: its writer is gone";
        else break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        return fail(ERR_INVALID, "rank %d: no usable rendezvous id at %s after %.0f s (%s)", rank, path, timeout_s, why.c_str());
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  const int rc = mlmcpi_comm_init(rank, world, rec.id, device, out);
  if (rc == OK && rank == 0) unlink(path);  // every rank has joined once ncclCommInitRank returns on rank 0
  return rc;
}

// What the COMMUNICATOR says (ncclCommUserRank / ncclCommCount), not what the caller passed to mlmcpi_comm_init
int mlmcpi_comm_rank(const mlmcpi_comm *c, int *rank) {
  if (!c || !rank) return fail(ERR_INVALID, "bad arguments");
  NCCL_TRY(g_rccl.CommUserRank(c->comm, rank));
  return OK;
}

int mlmcpi_comm_size(const mlmcpi_comm *c, int *size) {
  if (!c || !size) return fail(ERR_INVALID, "bad arguments");
  NCCL_TRY(g_rccl.CommCount(c->comm, size));
  return OK;
}

int mlmcpi_comm_allreduce_sum_f64(mlmcpi_comm *c, double *d_buf, size_t n, void *stream) {
  if (!c || !d_buf || n == 0) return fail(ERR_INVALID, "bad arguments");
  NCCL_TRY(g_rccl.AllReduce(d_buf, d_buf, n, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream));
  return OK;
}

int mlmcpi_comm_allreduce_sum_host_f64(mlmcpi_comm *c, double *h_buf, size_t n) {
  if (!c || !h_buf || n == 0) return fail(ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  if (n > c->cap) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->d_stage = nullptr; c->h_pinned = nullptr; c->cap = 0;
    const size_t cap = n < 256 ? 256 : n;
    HIP_TRY(hipMalloc((void **)&c->d_stage, cap * sizeof(double)));
    HIP_TRY(hipHostMalloc((void **)&c->h_pinned, cap * sizeof(double), hipHostMallocDefault));
    c->cap = cap;
  }
  memcpy(c->h_pinned, h_buf, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_pinned, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.AllReduce(c->d_stage, c->d_stage, n, ncclFloat64, ncclSum, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_pinned, c->d_stage, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(h_buf, c->h_pinned, n * sizeof(double));
  return OK;
}

int mlmcpi_comm_destroy(mlmcpi_comm *c) {
  if (!c) return OK;
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
  return OK;
}

}  // extern "C"
