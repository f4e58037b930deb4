// issue_rate.hip -- how many shader cycles one SIMD of gfx950 spends per wave64 VALU instruction, measured with
// s_memtime INSIDE the kernel (so the answer does not depend on the clock the chip happens to run at), plus the
// shader clock itself (s_memtime ticks per wall-clock second).  This is the "peak" of bench.py's VALU roofline.
//
//   one workgroup per CU, W waves per SIMD (W = 1, 2, 4: blocks of 256 / 512 / 1024 threads), every lane runs
//   UNROLL independent chains of ONE instruction for N_IT iterations; cycles per wave-instruction per SIMD
//   = elapsed ticks / (W * UNROLL * N_IT).  "dep" rows run ONE dependent chain (issue + latency).
//
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 4096
#define UNROLL 16

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7) op(8) op(9) op(10) op(11) op(12) op(13) op(14) op(15)

// every kernel writes (ticks of wave 0 of the block) to out[blockIdx.x]
#define KERNEL(name, type, init, op, fold)                                                              \
  __global__ __launch_bounds__(1024) void k_##name(unsigned long long* out, float* sink, float seed) { \
    type a[UNROLL];                                                                                     \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) a[i] = init;                                    \
    type m, c;                                                                                          \
    { const int i = 17; m = init; }                                                                     \
    { const int i = 23; c = init; }                                                                     \
    (void)m, (void)c;                                                                                   \
    __syncthreads();                                                                                    \
    const unsigned long long w0 = wall_clock64();                                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    _Pragma("unroll 1") for (int it = 0; it < N_IT; it++) { REP16(op) }                                 \
    __syncthreads();                                                                                    \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    const unsigned long long w1 = wall_clock64();                                                       \
    if (threadIdx.x == 0) out[2 * blockIdx.x] = t1 - t0, out[2 * blockIdx.x + 1] = w1 - w0;             \
    float s = 0;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) s += fold;                                       \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                    \
  }

#define FINIT (seed + (float)threadIdx.x * 1e-3f + (float)i)
#define PINIT (f2{seed + (float)threadIdx.x * 1e-3f + (float)i, seed})
#define UINIT ((uint32_t)(seed * 977.f) + threadIdx.x + (uint32_t)i)

#define OP_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
#define OP_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_SUB(k) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[k]) : "v"(m));
#define OP_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_MAX(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
#define OP_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m) : );
#define OP_CMP(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(m) : "vcc");
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
#define OP_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define OP_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define OP_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(m));
#define OP_DEP_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
#define OP_DEP_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(m));
#define OP_DEP_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));

KERNEL(fma, float, FINIT, OP_FMA, a[i])
KERNEL(add, float, FINIT, OP_ADD, a[i])
KERNEL(sub, float, FINIT, OP_SUB, a[i])
KERNEL(mul, float, FINIT, OP_MUL, a[i])
KERNEL(max, float, FINIT, OP_MAX, a[i])
KERNEL(max3, float, FINIT, OP_MAX3, a[i])
KERNEL(cndmask, float, FINIT, OP_CND, a[i])
KERNEL(cmp, float, FINIT, OP_CMP, a[i])
KERNEL(mov, float, FINIT, OP_MOV, a[i])
KERNEL(pk_mul, f2, PINIT, OP_PKMUL, a[i].x)
KERNEL(pk_add, f2, PINIT, OP_PKADD, a[i].x)
KERNEL(pk_fma, f2, PINIT, OP_PKFMA, a[i].x)
KERNEL(add_u32, uint32_t, UINIT, OP_ADDU, (float)a[i])
KERNEL(xor, uint32_t, UINIT, OP_XOR, (float)a[i])
KERNEL(mul_lo, uint32_t, UINIT, OP_MULLO, (float)a[i])
KERNEL(mul_hi, uint32_t, UINIT, OP_MULHI, (float)a[i])
KERNEL(rcp, float, FINIT, OP_RCP, a[i])
KERNEL(sqrt, float, FINIT, OP_SQRT, a[i])
KERNEL(dep_fma, float, FINIT, OP_DEP_FMA, a[i])
KERNEL(dep_add, float, FINIT, OP_DEP_ADD, a[i])
KERNEL(dep_max3, float, FINIT, OP_DEP_MAX3, a[i])

// the box step of rt_pool.h as an instruction mix: 6 sub, 6 mul, 2 max3/min3, 2 max/min, cmp, cndmask, 3 add_u32, cmp
#define OP_BOX(k)                                                                      \
  asm volatile(                                                                        \
      "v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n"         \
      "v_sub_f32 %3, %5, %3\n v_sub_f32 %0, %5, %0\n v_sub_f32 %1, %5, %1\n"         \
      "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n"         \
      "v_mul_f32 %3, %3, %5\n v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n"         \
      "v_max3_f32 %2, %0, %1, %2\n v_min3_f32 %3, %0, %1, %3\n"                       \
      "v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %5\n"                                 \
      "v_cmp_gt_f32 vcc, %3, %2\n v_cndmask_b32 %0, %0, %1, vcc\n"                    \
      "v_add_u32 %1, %1, %0\n v_add_u32 %2, %2, %0\n v_add_u32 %3, %3, %0\n"         \
      "v_cmp_gt_i32 vcc, 0, %0\n"                                                      \
      : "+v"(a[(4 * k) & 15]), "+v"(a[(4 * k + 1) & 15]), "+v"(a[(4 * k + 2) & 15]), "+v"(a[(4 * k + 3) & 15]) \
      : "v"(m), "v"(c)                                                                  \
      : "vcc");
KERNEL(box_mix22, float, FINIT, OP_BOX, a[i])

template <typename K>
static void run(const char* name, K k, int cus, unsigned long long* d_out, float* d_sink, double insts_per_rep, double wall_hz) {
  printf("%-10s", name);
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, d_out, d_sink, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, d_out, d_sink, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * cus);
    hipMemcpy(h.data(), d_out, 2 * cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0, wsum = 0;
    for (int i = 0; i < cus; i++) sum += (double)h[2 * i], wsum += (double)h[2 * i + 1];
    const double ticks = sum / cus, wall_s = wsum / cus / wall_hz;
    // per SIMD: wps waves, each issuing UNROLL * N_IT * insts_per_rep instructions
    printf("  %dw/SIMD %6.2f cyc/inst (%4.0f MHz)", wps, ticks / (wps * (double)UNROLL * N_IT * insts_per_rep), ticks / wall_s / 1e6);
  }
  printf("\n");
}

__global__ void k_clock(unsigned long long* out, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long w0 = wall_clock64();
  unsigned long long t1 = t0;
  while ((long long)(wall_clock64() - w0) < spin) t1 = __builtin_amdgcn_s_memtime();
  out[0] = t1 - t0;
  out[1] = wall_clock64() - w0;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, (size_t)cus * 2 * sizeof(unsigned long long));
  hipMalloc(&d_sink, (size_t)cus * 1024 * sizeof(float));
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, d_out, wall_khz * 20);  // ~20 ms
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  const double memtime_hz = (double)h[0] / ((double)h[1] / (wall_khz * 1e3));
  printf("device %s: %d CUs, clockRate %d kHz, wall clock %d kHz, s_memtime ticks at %.1f MHz (idle chip, one lane)\n",
         p.name, cus, p.clockRate, wall_khz, memtime_hz / 1e6);
  printf("(MHz) = s_memtime ticks per wall-clock second while that kernel runs on every CU\n");
  printf("cycles per wave64 instruction per SIMD = s_memtime ticks / (waves per SIMD x instructions per wave)\n");
#define ROW(name, n) run(#name, k_##name, cus, d_out, d_sink, n, wall_khz * 1e3);
  ROW(fma, 1) ROW(add, 1) ROW(sub, 1) ROW(mul, 1) ROW(max, 1) ROW(max3, 1) ROW(cmp, 1) ROW(cndmask, 1) ROW(mov, 1)
  ROW(pk_mul, 1) ROW(pk_add, 1) ROW(pk_fma, 1) ROW(add_u32, 1) ROW(xor, 1) ROW(mul_lo, 1) ROW(mul_hi, 1) ROW(rcp, 1) ROW(sqrt, 1)
  ROW(dep_fma, 1) ROW(dep_add, 1) ROW(dep_max3, 1)
  ROW(box_mix22, 22)
  return 0;
}
