// exec_width.hip -- does a wave64 VALU instruction get cheaper when only part of the wave is active?  (If gfx950 skipped
// the 32-lane half -- or 16-lane quarter -- of a wave whose EXEC bits are all zero, compacting the active lanes of a divergent
// pass into the low lanes would pay.)  Same method as issue_rate.hip: s_memtime inside the kernel, one 1024-thread workgroup
// per CU (4 waves per SIMD), 16 independent chains of the box step's instruction mix, executed by lanes [0, LIM) / a strided
// set of LIM lanes.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 exec_width.hip -o exec_width && ./exec_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4096
template <int LIM, int STRIDE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + (float)threadIdx.x * 1e-3f + (float)i;
  const float m = seed * 1.0001f, c = seed * 0.5f;
  const int lane = threadIdx.x & 63;
  const bool on = (lane % STRIDE) == 0 && (lane / STRIDE) < LIM;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++)
        asm volatile(
            "v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %5, %3\n v_sub_f32 %0, %5, %0\n v_sub_f32 %1, %5, %1\n"
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5\n v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n"
            "v_max3_f32 %2, %0, %1, %2\n v_min3_f32 %3, %0, %1, %3\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %5\n"
            "v_cmp_gt_f32 vcc, %3, %2\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %1, %1, %0\n v_add_u32 %2, %2, %0\n v_add_u32 %3, %3, %0\n v_cmp_gt_i32 vcc, 0, %0\n"
            : "+v"(a[4 * k4]), "+v"(a[4 * k4 + 1]), "+v"(a[4 * k4 + 2]), "+v"(a[4 * k4 + 3]) : "v"(m), "v"(c) : "vcc");
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int LIM, int STRIDE>
static void run(const char* what, int cus, unsigned long long* d_out, float* d_sink) {
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((k<LIM, STRIDE>), dim3(cus), dim3(1024), 0, 0, d_out, d_sink, 1.0f);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(cus);
  hipMemcpy(h.data(), d_out, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < cus; i++) sum += (double)h[i];
  printf("%-44s %6.2f cycles per wave instruction (4 waves per SIMD, box-step mix)\n", what, sum / cus / (4.0 * 4 * 22 * N_IT));
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, cus * sizeof(unsigned long long));
  hipMalloc(&d_sink, (size_t)cus * 1024 * sizeof(float));
  run<64, 1>("all 64 lanes", cus, d_out, d_sink);
  run<48, 1>("lanes 0-47", cus, d_out, d_sink);
  run<32, 1>("lanes 0-31 (one half empty)", cus, d_out, d_sink);
  run<16, 1>("lanes 0-15 (three quarters empty)", cus, d_out, d_sink);
  run<8, 1>("lanes 0-7", cus, d_out, d_sink);
  run<1, 1>("lane 0", cus, d_out, d_sink);
  run<32, 2>("32 lanes, every other one", cus, d_out, d_sink);
  run<16, 4>("16 lanes, every fourth", cus, d_out, d_sink);
  return 0;
}
