// valu_cost.hip -- issue cost of the VALU instructions the ray-pool kernels lean on, relative to v_fma_f32, on
// this GPU: 4 waves per SIMD (1024-thread workgroups, one per CU), 8 independent chains per lane so that latency
// never shows.  Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 40000
#define DEF(name, decl, body, fold)                                                            \
  __global__ __launch_bounds__(1024) void k_##name(float* out, float seed, uint32_t useed) {   \
    decl;                                                                                      \
    _Pragma("unroll 1") for (int i = 0; i < N_IT; i++) { body; }                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = fold;                                         \
  }
#define F8 float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; float m = seed * 0.999f, c = seed * 0.5f
#define FSUM (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define REP8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)
#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define OP_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define OP_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define OP_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "s"(msk));
#define OP_DIVSCALE(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(m) : "vcc");
#define OP_DIVFIXUP(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
DEF(fma, F8, REP8(OP_FMA), FSUM)
DEF(mul, F8, REP8(OP_MUL), FSUM)
DEF(max3, F8, REP8(OP_MAX3), FSUM)
DEF(rcp, F8, REP8(OP_RCP), FSUM)
DEF(sqrt, F8, REP8(OP_SQRT), FSUM)
DEF(cndmask, F8; unsigned long long msk = __builtin_amdgcn_ballot_w64(threadIdx.x & 1), REP8(OP_CNDMASK), FSUM)
DEF(divscale, F8, REP8(OP_DIVSCALE), FSUM)
DEF(divfixup, F8, REP8(OP_DIVFIXUP), FSUM)
typedef float f2 __attribute__((ext_vector_type(2)));
#define P8 f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; f2 m = {seed * 0.999f, seed}, c = {seed * 0.5f, seed}
#define PSUM ((a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7).x)
#define OP_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
DEF(pk_mul, P8, REP8(OP_PKMUL), PSUM)
DEF(pk_add, P8, REP8(OP_PKADD), PSUM)
DEF(pk_fma, P8, REP8(OP_PKFMA), PSUM)
#define U8 uint32_t a0 = useed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t m = useed | 1u
#define USUM ((float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7))
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(m));
#define OP_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(m));
DEF(mul_lo_u32, U8, REP8(OP_MULLO), USUM)
DEF(mul_hi_u32, U8, REP8(OP_MULHI), USUM)
DEF(mul_u32_u24, U8, REP8(OP_MUL24), USUM)
DEF(xor, U8, REP8(OP_XOR), USUM)
DEF(add_u32, U8, REP8(OP_ADDU), USUM)
typedef unsigned long long u64;
#define Q8 u64 a0 = useed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t m = useed | 1u; uint32_t lo
#define QSUM ((float)(uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7))
#define OP_MAD64(x) lo = (uint32_t)x; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(lo), "v"(m) : "vcc");
DEF(mad_u64_u32, Q8, REP8(OP_MAD64), QSUM)
typedef double d1;
#define D8 double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; double m = seed * 0.999, c = seed * 0.5
#define DSUM ((float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
#define OP_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define OP_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(m));
DEF(fma_f64, D8, REP8(OP_FMA64), DSUM)
DEF(mul_f64, D8, REP8(OP_MUL64), DSUM)

#define CH8(op) op(a0) op(a0) op(a0) op(a0) op(a0) op(a0) op(a0) op(a0)
DEF(lat_fma, F8, CH8(OP_FMA), FSUM)
DEF(lat_mul, F8, CH8(OP_MUL), FSUM)
DEF(lat_max3, F8, CH8(OP_MAX3), FSUM)
DEF(lat_add_u32, U8, CH8(OP_ADDU), USUM)
DEF(lat_pk_mul, P8, CH8(OP_PKMUL), PSUM)
DEF(lat_pk_add, P8, CH8(OP_PKADD), PSUM)
DEF(lat_cndmask, F8; unsigned long long msk = __builtin_amdgcn_ballot_w64(threadIdx.x & 1), CH8(OP_CNDMASK), FSUM)
DEF(lat_rcp, F8, CH8(OP_RCP), FSUM)
DEF(lat_mad_u64_u32, Q8, CH8(OP_MAD64), QSUM)
template <typename K>
static double run(K k, float* out, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, out, 1.0f, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, out, 1.0f, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
  const double base = run(k_fma, out, cus);
  // 4 waves per SIMD x 8 ops x N_IT iterations per wave
  const double clk = p.clockRate * 1e3;  // Hz
  printf("%-14s %8s %10s %12s\n", "instruction", "ms", "vs fma", "cyc/wave-op");
#define ROW(name) { double ms = run(k_##name, out, cus); printf("%-14s %8.3f %10.2f %12.2f\n", #name, ms, ms / base, ms * 1e-3 * clk / (4.0 * 8 * N_IT)); }
  ROW(fma) ROW(mul) ROW(max3) ROW(cndmask) ROW(add_u32) ROW(xor) ROW(mul_u32_u24) ROW(pk_mul) ROW(pk_add) ROW(pk_fma) ROW(rcp) ROW(sqrt)
  printf("-- one dependent chain per lane (4 waves per SIMD): cycles per wave-op = what a SIMD spends per op when every wave is latency-bound\n");
  ROW(lat_fma) ROW(lat_mul) ROW(lat_max3) ROW(lat_add_u32) ROW(lat_pk_mul) ROW(lat_pk_add) ROW(lat_cndmask) ROW(lat_rcp) ROW(lat_mad_u64_u32)
  ROW(divscale) ROW(divfixup) ROW(mul_lo_u32) ROW(mul_hi_u32) ROW(mad_u64_u32) ROW(fma_f64) ROW(mul_f64)
  return 0;
}
