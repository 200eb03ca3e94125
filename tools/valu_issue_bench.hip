// valu_issue_bench.hip -- issue cost of the vector instructions the heat-bath kernel is made of, relative to v_add_f32.
// Every kernel runs the same loop of independent instructions (8 register chains) on every SIMD of the chip with 8
// waves per SIMD; time / (instructions per wave x waves per SIMD) is the issue cost per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o tools/build/valu_issue_bench (build/ is git-ignored but travels to the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, DECL, BODY, SINK)                                         \
  __global__ void __launch_bounds__(256) NAME(double *out, int iters) {        \
    DECL                                                                       \
    for (int it = 0; it < iters; ++it) {                                       \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(BODY) }             \
    }                                                                          \
    SINK                                                                       \
  }

#define DDECL double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; double b = 1.0000001, c = 1e-9;
#define DSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;
#define FDECL float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; float b = 1.0000001f, c = 1e-9f;
#define UDECL uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t b = 0xD2511F53u;
#define USINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = a0;
#define LDECL uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t b = 0xD2511F53u;

#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
#define B_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
#define B_FLOOR64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(a##i));
#define B_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define B_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define B_LOG32(i) asm volatile("v_log_f32 %0, %0" : "+v"(a##i));
#define B_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a##i) : "v"(b), "v"((uint32_t)i + 3u) : "vcc");
#define B_CVT(i) { double t; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t) : "v"(a##i)); asm volatile("" :: "v"(t)); }
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define B_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a##i), "v"(c) : "vcc");
#define B_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(a##i) : "v"(b));
#define B_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a##i) : "v"(1));
#define B_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_LSHR(i) asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(a##i));
#define B_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
#define B_FMA64S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(c));
#define B_CVT3264(i) { float t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(a##i)); asm volatile("" :: "v"(t)); }
#define B_CND64(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(m));
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a##i));
#define B_FRACT64(i) asm volatile("v_fract_f64 %0, %0" : "+v"(a##i));
#define B_RNDNE64(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a##i));

#define B_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##i) : "v"(b), "v"((uint32_t)i + 3u));
#define B_BITOP3S(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##i) : "v"(b), "s"(sk));
#define B_XORS(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a##i) : "s"(sk));
#define B_CVTF32U32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
#define B_CVTUB0(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a##i));
#define B_EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a##i) : "v"(b));
#define B_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
#define B_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3c088889" : "+v"(a##i) : "v"(b));
#define B_CMPF32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a##i), "v"(c) : "vcc");
#define B_CVTU32F64(i) { uint32_t t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(a##i)); asm volatile("" :: "v"(t)); }
#define UDECLS UDECL uint32_t sk = __builtin_amdgcn_readfirstlane(blockIdx.x * 7u + 3u);
#define B_MAD64S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a##i) : "s"(sk), "v"((uint32_t)i + 3u) : "vcc");
#define LDECLS LDECL uint32_t sk = __builtin_amdgcn_readfirstlane(blockIdx.x * 7u + 3u);
KERNEL(k_mad_u64_u32_sgpr, LDECLS, B_MAD64S, USINK)
#define B_ADDINL(i) asm volatile("v_add_u32 %0, 64, %0" : "+v"(a##i));
KERNEL(k_add_u32_inline, UDECL, B_ADDINL, USINK)
#define B_ADDLIT(i) asm volatile("v_add_u32 %0, 0x9E3779B9, %0" : "+v"(a##i));
KERNEL(k_add_u32_literal, UDECL, B_ADDLIT, USINK)
#define B_FMA32S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "s"(sf), "v"(c));
#define FDECLS FDECL float sf = __builtin_amdgcn_readfirstlane((int)blockIdx.x) * 1e-9f + 1.0f;
KERNEL(k_fma32_sgpr, FDECLS, B_FMA32S, DSINK)
#define B_ADD64S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "s"(sd));
#define DDECLS DDECL double sd = (double)__builtin_amdgcn_readfirstlane((int)blockIdx.x) * 1e-9;
KERNEL(k_add64_sgpr, DDECLS, B_ADD64S, DSINK)
KERNEL(k_bitop3, UDECL, B_BITOP3, USINK)
KERNEL(k_bitop3_sgpr, UDECLS, B_BITOP3S, USINK)
KERNEL(k_xor_sgpr, UDECLS, B_XORS, USINK)
KERNEL(k_cvt_f32_u32, UDECL, B_CVTF32U32, USINK)
KERNEL(k_cvt_f32_ubyte0, UDECL, B_CVTUB0, USINK)
KERNEL(k_exp32, FDECL, B_EXP32, DSINK)
KERNEL(k_lshl_add, UDECL, B_LSHLADD, USINK)
KERNEL(k_add3, UDECL, B_ADD3, USINK)
KERNEL(k_min_u32, UDECL, B_MINU, USINK)
KERNEL(k_fmaak32, FDECL, B_FMAAK, DSINK)
KERNEL(k_cmp32, FDECL, B_CMPF32, DSINK)
KERNEL(k_cvt_u32_f64, DDECL, B_CVTU32F64, DSINK)
KERNEL(k_fma64, DDECL, B_FMA64, DSINK)
KERNEL(k_add64, DDECL, B_ADD64, DSINK)
KERNEL(k_mul64, DDECL, B_MUL64, DSINK)
KERNEL(k_rcp64, DDECL, B_RCP64, DSINK)
KERNEL(k_rsq64, DDECL, B_RSQ64, DSINK)
KERNEL(k_floor64, DDECL, B_FLOOR64, DSINK)
KERNEL(k_ldexp64, DDECL, B_LDEXP, DSINK)
KERNEL(k_cmp64, DDECL, B_CMP64, DSINK)
KERNEL(k_add32, FDECL, B_ADD32, DSINK)
KERNEL(k_fma32, FDECL, B_FMA32, DSINK)
KERNEL(k_log32, FDECL, B_LOG32, DSINK)
KERNEL(k_xor, UDECL, B_XOR, USINK)
KERNEL(k_mullo, UDECL, B_MULLO, USINK)
KERNEL(k_mulhi, UDECL, B_MULHI, USINK)
KERNEL(k_cvt_f64_u32, UDECL, B_CVT, USINK)
KERNEL(k_cndmask, UDECL, B_CNDMASK, USINK)
KERNEL(k_alignbit, UDECL, B_ALIGN, USINK)
KERNEL(k_mad_u64_u32, LDECL, B_MAD64, USINK)
KERNEL(k_fract64, DDECL, B_FRACT64, DSINK)
KERNEL(k_rndne64, DDECL, B_RNDNE64, DSINK)
KERNEL(k_fma64_sgpr, DDECL, B_FMA64S, DSINK)
KERNEL(k_cvt_f32_f64, DDECL, B_CVT3264, DSINK)
KERNEL(k_add_u32, UDECL, B_ADDU, USINK)
KERNEL(k_lshr, UDECL, B_LSHR, USINK)
KERNEL(k_and_or, UDECL, B_ANDOR, USINK)
KERNEL(k_bfe, UDECL, B_BFE, USINK)
#define UDECLM UDECL uint64_t m = __ballot(threadIdx.x & 1);
KERNEL(k_cndmask_sgpr, UDECLM, B_CND64, USINK)

#define RUN(NAME) run(#NAME, NAME)
static double g_base = 0;
template <class K>
static void run(const char *name, K kern) {
  double *d;
  hipMalloc(&d, 8);
  const int iters = 2000, grid = 256 * 4 * 2;  // 256-thread workgroups = 4 waves: 8 workgroups per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 64 * 8;  // per wave x 8 waves per SIMD
  const double ns = ms * 1e6 / insts_per_simd;
  if (g_base == 0) g_base = ns;
  printf("%-16s %8.3f ms  %6.3f ns/wave-inst  %5.2f x v_add_f32  (%.1f cycles at 2.4 GHz)\n", name, ms, ns, ns / g_base, ns * 2.4);
  hipFree(d);
}

int main() {
  RUN(k_add32); RUN(k_fma32); RUN(k_log32); RUN(k_fma64); RUN(k_fma64_sgpr); RUN(k_add64); RUN(k_mul64); RUN(k_rcp64);
  RUN(k_rsq64); RUN(k_floor64); RUN(k_fract64); RUN(k_rndne64); RUN(k_ldexp64); RUN(k_cmp64); RUN(k_cvt_f32_f64);
  RUN(k_cvt_f64_u32); RUN(k_xor); RUN(k_add_u32); RUN(k_lshr); RUN(k_and_or); RUN(k_bfe); RUN(k_alignbit); RUN(k_mullo);
  RUN(k_mulhi); RUN(k_mad_u64_u32); RUN(k_cndmask_sgpr);
  RUN(k_bitop3); RUN(k_bitop3_sgpr); RUN(k_xor_sgpr); RUN(k_cvt_f32_u32); RUN(k_cvt_f32_ubyte0); RUN(k_exp32); RUN(k_lshl_add);
  RUN(k_mad_u64_u32_sgpr); RUN(k_add_u32_inline); RUN(k_add_u32_literal); RUN(k_fma32_sgpr); RUN(k_add64_sgpr); RUN(k_add3); RUN(k_min_u32); RUN(k_fmaak32); RUN(k_cmp32); RUN(k_cvt_u32_f64);
  return 0;
}
