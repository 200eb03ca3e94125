// valu_dep_bench.hip -- how much of a SIMD's issue rate can W waves reach when every wave runs C independent dependency
// chains?  (The heat-bath cell of the fused Schwinger launch is one long chain per lane -- Philox rounds, polynomial,
// exponential -- and a CU holds 4 waves per SIMD: is the vector pipe idle because no wave has a ready instruction?)
// Per (instruction kind, chains per wave, waves per SIMD): cycles per wave-instruction seen by one SIMD = elapsed SIMD
// cycles / (instructions per wave x waves per SIMD).  Blocks of 256 threads (one wave per SIMD), W blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_dep_bench.hip -o tools/build/valu_dep_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

enum Op { FMA32, ADD64, FMA64, MAD64, BITOP3, PHILOX, EXP32, PKFMA32 };

template <int OP>
__device__ __forceinline__ void step(uint32_t &lo, uint32_t &hi, uint32_t k) {
  if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(lo) : "v"(k));
  if (OP == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(lo));
  if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo) : "v"(k), "v"(hi));
  if (OP == ADD64 || OP == FMA64 || OP == MAD64 || OP == PHILOX || OP == PKFMA32) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"((uint64_t)k << 32));
    if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v) : "v"((uint64_t)k << 32));
    if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(((uint64_t)k << 32) | k));
    if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(v) : "v"(lo), "v"(k) : "vcc");
    if (OP == PHILOX) {  // one half round: product, then hi ^ counter ^ key (two instructions)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(v) : "v"(lo), "v"(k) : "vcc");
      uint32_t h = (uint32_t)(v >> 32), l = (uint32_t)v;
      asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(h) : "v"(l), "v"(k));
      v = ((uint64_t)l << 32) | h;
    }
    lo = (uint32_t)v;
    hi = (uint32_t)(v >> 32);
  }
}

template <int OP, int C>
__global__ void __launch_bounds__(256) dep_kernel(uint32_t *out, int iters) {
  uint32_t lo[C], hi[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    lo[c] = threadIdx.x * 7u + c + 0x3F800000u;
    hi[c] = 0x3FF00000u + c;
  }
  uint32_t k;
  asm volatile("v_mov_b32 %0, 0x3F7FFFFF" : "=v"(k));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) step<OP>(lo[c], hi[c], k);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += lo[c] ^ hi[c];
  if (s == 0x12345u) out[0] = s;
}

template <int OP, int C>
static void run(const char *name, uint32_t *d_out, int n_cu) {
  const int insts_per_step = OP == PHILOX ? 2 : 1;
  for (int W : {1, 2, 4, 8}) {
    const int iters = 4096 / C;   // the same number of instructions per wave whatever C
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((dep_kernel<OP, C>), dim3(n_cu * W), dim3(256), 0, 0, d_out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((dep_kernel<OP, C>), dim3(n_cu * W), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 16 * C * insts_per_step;
    const double cyc = ms * 1e-3 * 2.4e9 / (insts * W);
    printf("%-8s chains/wave %d  waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction on the SIMD  (%5.2f per wave)\n", name, C, W, ms,
           cyc, cyc * W);
  }
}

int main() {
  uint32_t *d_out;
  hipMalloc(&d_out, 4096);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs; a wave's own cycles per instruction = (cycles on the SIMD) x (waves per SIMD)\n", prop.name, n_cu);
#define ALL(OP) run<OP, 1>(#OP, d_out, n_cu); run<OP, 2>(#OP, d_out, n_cu); run<OP, 4>(#OP, d_out, n_cu);
  ALL(PKFMA32) ALL(FMA32) ALL(BITOP3) ALL(MAD64) ALL(PHILOX) ALL(ADD64) ALL(FMA64) ALL(EXP32)
  return 0;
}
