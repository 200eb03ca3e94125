// internal.hpp -- host-side plumbing shared by the translation units of libmlmcpi_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mlmcpi_hip.h"
#include "device_common.hpp"

namespace mlmcpi {

// error channel (thread local)
int fail(int status, const char *fmt, ...);
int fail_hip(hipError_t e, const char *what);

#define MLMCPI_HIP_TRY(expr)                              \
  do {                                                    \
    hipError_t e__ = (expr);                              \
    if (e__ != hipSuccess) return fail_hip(e__, #expr);   \
  } while (0)

// launch check: kernels report configuration errors through hipGetLastError
#define MLMCPI_LAUNCH_CHECK(name)                         \
  do {                                                    \
    hipError_t e__ = hipGetLastError();                   \
    if (e__ != hipSuccess) return fail_hip(e__, name);    \
  } while (0)

#define MLMCPI_REQUIRE(cond, ...)                         \
  do {                                                    \
    if (!(cond)) return fail(MLMCPI_ERR_INVALID, __VA_ARGS__); \
  } while (0)

inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

// Library-owned scratch for reduction partials: one buffer per (host thread, device, stream), grown on demand (the
// first call of a given size allocates; steady state does not).  `stream` = the stream the caller launches on.
int scratch(size_t bytes, void **d_ptr, hipStream_t stream);

inline RngKey make_key(uint64_t seed, uint32_t chain0, uint32_t step) {
  return RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), chain0, step};
}

// Tuning knobs (they never change results): read from the environment ONCE, at the first sweep call of the process, and
// settable afterwards through mlmcpi_set_option (tests flip them between calls).
struct Tuning {
  uint32_t tile_w = 0, tile_h = 0, tile_nt = 0;  // MLMCPI_SWEEP_TILE=TWxTHxNT (0: default geometry)
  bool or_lds = false;                           // MLMCPI_OR_KERNEL=lds: LDS-resident instead of register-tiled overrelaxation
  bool or_patch = false;                         // MLMCPI_OR_KERNEL=patch: 2 x 2 register blocks on 64 x 32 tiles instead of 4 x 4 on 64 x 64
  bool or_block = false;                         // MLMCPI_OR_KERNEL=block: 4 x 4 register blocks sweep by sweep instead of the closed form (Schwinger)
  uint32_t or_threads = 0;                       // MLMCPI_OR_THREADS (LDS-resident kernel's workgroup size; 0: default)
  bool or_heat_split = false;                    // MLMCPI_OR_HEAT=split: the heat-bath sweep behind the last overrelaxation launch gets a launch of its own
  int or_heat_wide = 0;                          // MLMCPI_OR_HEAT=wide|narrow: 1024-thread workgroups for the fused launch (+1 / -1; 0: by the number of tiles)
};
Tuning tuning();  // a copy taken under the lock: callers snapshot it once per call

// Table of the step-envelope heat-bath sampler for an action of the given scale (2 beta, 2 m0 / a), in the memory of the
// current device: built and uploaded on first use (device_common.hpp, runtime.hip).
int vs_table_device(double scale, const uint32_t **d_table);

constexpr uint32_t kComputeUnits = 256;  // MI355X
constexpr uint32_t kMaxFuse = 16;  // max sweeps fused in one launch (kinds travel in a bitmask)

}  // namespace mlmcpi
