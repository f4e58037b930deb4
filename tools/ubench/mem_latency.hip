// mem_latency.hip -- what a wave of gfx950 waits for when it hands data to itself through global memory:
//   (a) a dependent load from a line it read before (L2 hit), from a per-wave region of `footprint` bytes;
//   (b) `n_st` coalesced 256-byte stores followed by one load of the LAST stored row and s_waitcnt vmcnt(0)
//       (vmcnt counts loads and stores in order on gfx9: the load's wait is also the stores' acknowledgement);
//   (c) the same stores with NO wait (issue cost only).
// All with 16 waves per CU on every CU (the render kernels' occupancy), ticks of s_memtime per iteration, averaged.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 mem_latency.hip -o mem_latency && ./mem_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 500
constexpr uint32_t ROW = 160;  // dwords per row (the stacks' FPOOL)

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* mem, unsigned long long* out, uint32_t rows_per_wave, uint32_t n_st, int work) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  uint32_t* base = mem + (size_t)gwave * rows_per_wave * ROW;
  uint32_t acc = 0, row = 0;
  unsigned long long total = 0;
  for (int it = 0; it < N_IT; it++) {
    // some unrelated work between iterations so that earlier stores drain
    for (int j = 0; j < work; j++) acc = acc * 1664525u + 1013904223u;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {  // one dependent load
      acc += base[row * ROW + ((lane + acc) & 63u)];
    } else {
      for (uint32_t s = 0; s < n_st; s++) base[((row + s) & (rows_per_wave - 1u)) * ROW + lane] = acc + s;  // rows_per_wave: a power of two
      if (MODE == 1) acc += base[((row + n_st - 1) & (rows_per_wave - 1u)) * ROW + (63u - lane)];
    }
    if (MODE != 2) __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    total += t1 - t0;
    row = (row + n_st + (acc & 1u)) & (rows_per_wave - 1u);
  }
  if (lane == 0) out[gwave] = total / N_IT;
  if (acc == 0x12345u) mem[0] = acc;
}

// the render kernels' own access shape: one buffer resource per wave, the row a constant offset, N stores unrolled
template <int N, bool WAIT, int LANES = 64>
__global__ __launch_bounds__(1024) void kb(uint32_t* mem, unsigned long long* out, int work) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  uint32_t* base = mem + (size_t)gwave * 128 * ROW;
  const uint64_t bv = (uint64_t)base;
  uint32_t* ub = (uint32_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(bv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)bv));
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 128 * ROW * 4, 0x00020000);
  uint32_t acc = lane;
  unsigned long long total = 0;
  for (int it = 0; it < N_IT; it++) {
    for (int j = 0; j < work; j++) acc = acc * 1664525u + 1013904223u;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // LANES of the 64 lanes store, to LANES consecutive positions somewhere in the row (a compacted push)
    const uint32_t pos = __builtin_amdgcn_readfirstlane((acc >> 8) % 96u) + lane / (64 / LANES);
    if (lane % (64 / LANES) == 0) {
#pragma unroll
      for (int k = 0; k < N; k++) __builtin_amdgcn_raw_buffer_store_b32(acc + k, r, pos * 4u + ((k * ROW * 4u) & 4095u), (k * ROW * 4u) & ~4095u, 0);
    }
    if (WAIT) {
      acc += __builtin_amdgcn_raw_buffer_load_b32(r, (pos + 63u - 2u * lane) * 4u + (((N - 1) * ROW * 4u) & 4095u), ((N - 1) * ROW * 4u) & ~4095u, 0);
      __builtin_amdgcn_s_waitcnt(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    total += t1 - t0;
  }
  if (lane == 0) out[gwave] = total / N_IT;
  if (acc == 0x12345u) mem[0] = acc;
}

// store bursts of 4 waves per CU (one per SIMD) while the other 12 waves run BG: 0 = nothing, 1 = a VALU loop, 2 = random
// ds_read_b128 pairs (the render kernels' program fetches)
template <int N, int BG>
__global__ __launch_bounds__(1024) void kc(uint32_t* mem, unsigned long long* out, volatile int* stop) {
  __shared__ uint4 lds[4096];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(i * 7u, i * 13u, i, 1u);
  __syncthreads();
  const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * 16 + wave);
  uint32_t acc = lane * 2654435761u + gwave;
  if (wave >= 4) {  // background waves: run until the measuring waves are done
    if (BG == 0) return;
    for (int it = 0; it < 400000; it++) {
      if (BG == 1) {
#pragma unroll
        for (int j = 0; j < 32; j++) acc = acc * 1664525u + 1013904223u;
      } else {
        const uint4 a = lds[acc & 4095u], b = lds[(acc >> 12) & 4095u];
        acc = acc * 1664525u + (a.x ^ b.y) + 1013904223u;
      }
      if ((it & 255) == 0 && *stop) break;
    }
    if (acc == 0x12345u) mem[1] = acc;
    return;
  }
  uint32_t* base = mem + (size_t)gwave * 128 * ROW;
  const uint64_t bv = (uint64_t)base;
  uint32_t* ub = (uint32_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(bv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)bv));
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 128 * ROW * 4, 0x00020000);
  unsigned long long total = 0;
  for (int it = 0; it < N_IT; it++) {
    for (int j = 0; j < 1000; j++) acc = acc * 1664525u + 1013904223u;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const uint32_t pos = __builtin_amdgcn_readfirstlane((acc >> 8) % 96u) + lane / 4;
    if (lane % 4 == 0) {
#pragma unroll
      for (int k = 0; k < N; k++) __builtin_amdgcn_raw_buffer_store_b32(acc + k, r, pos * 4u + ((k * ROW * 4u) & 4095u), (k * ROW * 4u) & ~4095u, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    total += t1 - t0;
  }
  if (lane == 0) out[gwave] = total / N_IT;
  if (acc == 0x12345u) mem[0] = acc;
  if (threadIdx.x == 0) atomicAdd((int*)stop, 1);
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const uint32_t waves = (uint32_t)cus * 16;
  unsigned long long* d_out;
  hipMalloc(&d_out, waves * sizeof(unsigned long long));
  std::vector<unsigned long long> h(waves);
  auto run = [&](int mode, uint32_t rows, uint32_t n_st, int work = 64, int threads = 1024) {
    uint32_t* mem;
    hipMalloc(&mem, (size_t)waves * rows * ROW * 4);
    hipMemset(mem, 0, (size_t)waves * rows * ROW * 4);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(threads), 0, 0, mem, d_out, rows, n_st, work);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(threads), 0, 0, mem, d_out, rows, n_st, work);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(threads), 0, 0, mem, d_out, rows, n_st, work);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    const uint32_t used = (uint32_t)cus * (threads / 64);
    for (uint32_t i = 0; i < used; i++) s += (double)h[i];
    hipFree(mem);
    return s / used;
  };
  for (uint32_t rows : {16u, 128u, 512u})
    printf("load only, %5.1f KB per wave (%6.1f MB in all): %8.0f ticks\n", rows * ROW * 4 / 1024., (double)waves * rows * ROW * 4 / 1e6, run(0, rows, 1));
  for (uint32_t n_st : {1u, 4u, 13u, 23u, 41u}) {
    printf("%2u stores + load + wait: %8.0f ticks   stores only, no wait: %6.0f ticks\n", n_st, run(1, 128, n_st), run(2, 128, n_st));
  }
  // the render kernels' duty cycle: thousands of cycles of other work between two bursts
  for (int work : {64, 1000, 4000})
    for (uint32_t n_st : {13u, 41u})
      printf("work %4d LCG steps, %2u stores + load + wait: %8.0f ticks   stores only, no wait: %6.0f ticks   load only: %6.0f ticks\n", work, n_st,
             run(1, 128, n_st, work), run(2, 128, n_st, work), run(0, 128, n_st, work));
  // how much of that is the CU's store path shared by 16 waves that burst together: 1, 4, 16 waves per CU
  for (int threads : {64, 256, 1024})
    for (uint32_t n_st : {1u, 13u, 41u})
      printf("%2d waves per CU, %2u stores + load + wait: %8.0f ticks   stores only, no wait: %6.0f ticks   load only: %6.0f ticks\n", threads / 64, n_st,
             run(1, 128, n_st, 1000, threads), run(2, 128, n_st, 1000, threads), run(0, 128, n_st, 1000, threads));
  auto runb = [&](auto kern, int threads) {
    uint32_t* mem;
    hipMalloc(&mem, (size_t)waves * 128 * ROW * 4);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, mem, d_out, 1000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    const uint32_t used = (uint32_t)cus * (threads / 64);
    for (uint32_t i = 0; i < used; i++) s += (double)h[i];
    hipFree(mem);
    return s / used;
  };
  for (int threads : {64, 256, 1024}) {
    printf("unrolled buffer stores, %2d waves per CU:  1 store %5.0f (+load+wait %5.0f)   13 stores %5.0f (%5.0f)   41 stores %5.0f (%5.0f) ticks\n", threads / 64,
           runb(kb<1, false>, threads), runb(kb<1, true>, threads), runb(kb<13, false>, threads), runb(kb<13, true>, threads),
           runb(kb<41, false>, threads), runb(kb<41, true>, threads));
  }
  printf("16 of 64 lanes store (64 contiguous bytes per row), 16 waves per CU: 13 stores %5.0f (+load+wait %5.0f)   41 stores %5.0f (%5.0f) ticks\n",
         runb(kb<13, false, 16>, 1024), runb(kb<13, true, 16>, 1024), runb(kb<41, false, 16>, 1024), runb(kb<41, true, 16>, 1024));
  printf(" 4 of 64 lanes store (16 contiguous bytes per row), 16 waves per CU: 13 stores %5.0f (+load+wait %5.0f)   41 stores %5.0f (%5.0f) ticks\n",
         runb(kb<13, false, 4>, 1024), runb(kb<13, true, 4>, 1024), runb(kb<41, false, 4>, 1024), runb(kb<41, true, 4>, 1024));
  {
    int* d_stop;
    hipMalloc(&d_stop, 4);
    auto runc = [&](auto kern) {
      uint32_t* mem;
      hipMalloc(&mem, (size_t)waves * 128 * ROW * 4);
      hipMemset(d_stop, 0, 4);
      hipMemset(d_out, 0, waves * sizeof(unsigned long long));
      hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 0, 0, mem, d_out, d_stop);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double s = 0;
      uint32_t n = 0;
      for (uint32_t i = 0; i < waves; i++)
        if ((i & 15u) < 4u) s += (double)h[i], n++;
      hipFree(mem);
      return s / n;
    };
    printf("4 storing waves per CU, 18 stores (16 lanes), no wait: alone %5.0f   beside 12 VALU waves %5.0f   beside 12 LDS-reading waves %5.0f ticks\n",
           runc(kc<18, 0>), runc(kc<18, 1>), runc(kc<18, 2>));
  }
  return 0;
}
