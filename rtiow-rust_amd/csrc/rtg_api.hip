// rtg_api.hip -- kernels + the C ABI of include/rtiow_gpu.h (librtiow_gpu.so).
//
// There is no CPU fallback anywhere in this file: without a HIP device every compute entry point
// returns RTG_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: librccl is dlopen()ed by rtg_par_cast_multi, never linked

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rtiow_gpu.h"
#include "rt_pool.h"
#include "rt_pool_full.h"
#include "rt_sync_full.h"
#include "rt_trace.h"
#include "scene_builder.h"

using namespace rtg;

// ===================================================================================================
// Kernels
// ===================================================================================================
constexpr uint32_t FEAT_ALL = FEAT_XFORM | FEAT_MEDIUM | FEAT_RECT | FEAT_TEXTURE;

__device__ __forceinline__ void flush_counts(const Counts& c, uint32_t draws, unsigned long long* g) {
  // one atomic per counter per wave would be nicer; the instrumented variant is not the timed one
  atomicAdd(&g[0], (unsigned long long)c.aabb);
  atomicAdd(&g[1], (unsigned long long)c.prim);
  atomicAdd(&g[2], (unsigned long long)c.shaded);
  atomicAdd(&g[3], (unsigned long long)c.rays);
  atomicAdd(&g[4], (unsigned long long)draws);
}

// par_cast (lib.rs:363-376): one lane owns one pixel and folds its ns samples IN ORDER
// (iter::Sum is a left fold from (0,0,0), vec3.rs:195-203), then divides by ns.
// Block = 16x16 pixels, each wave an 8x8 sub-tile (primary rays of a wave stay coherent).
template <uint32_t FEAT, bool COUNT>
__global__ __launch_bounds__(256) void render_kernel(DevScene sc, DevCamera cam, DevParams P, float* out,
                                                     unsigned long long* counters) {
  const uint32_t nbx = (P.nx + 15u) / 16u;
  const uint32_t bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
  const uint32_t tiles_x = (P.nx + P.tile_w - 1u) / P.tile_w;
  const uint32_t tile = ((by * 16u) / P.tile_h) * tiles_x + (bx * 16u) / P.tile_w;
  if (tile % P.nranks != P.rank) return;
  const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
  const uint32_t x = bx * 16u + (w & 1u) * 8u + (l & 7u);
  const uint32_t row = by * 16u + (w >> 1) * 8u + (l >> 3);
  if (x >= P.nx || row >= P.ny) return;
  const uint32_t y = P.ny - 1u - row;  // lib.rs:328: row 0 is y = ny-1
  Counts cnt = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  V3 col = mk(0.f, 0.f, 0.f);
  for (uint32_t s = 0; s < P.ns; s++) {
    uint32_t bounces, draws;
    V3 c = sample_color<FEAT, COUNT>(sc, cam, P, x, y, s, cnt, bounces, draws);
    col = vadd(col, c);
    if (COUNT) total_draws += draws;
  }
  col = sdiv(col, (float)P.ns);  // lib.rs:374
  float* o = out + 3ull * ((size_t)row * P.nx + x);
  o[0] = col.x, o[1] = col.y, o[2] = col.z;
  if (COUNT) flush_counts(cnt, total_draws, counters);
}

template <uint32_t FEAT>
__global__ void debug_hit_top_kernel(DevScene sc, uint32_t n, const float* rays, uint32_t seed_lo, uint32_t seed_hi,
                                     float t_near, float* out, uint32_t* out_mat) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + 7ull * i;
  SampleRng rng;
  rng.init(((uint64_t)seed_hi << 32) | seed_lo, i, 0);
  rng.set_event(1);
  HitRec h;
  Counts cnt = {0, 0, 0, 0};
  bool hit = hit_top<FEAT, false>(sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), r[6], t_near, rng, h, cnt);
  float* o = out + 8ull * i;
  o[0] = hit ? 1.f : 0.f;
  o[1] = hit ? h.t : 0.f;
  o[2] = hit ? h.p.x : 0.f, o[3] = hit ? h.p.y : 0.f, o[4] = hit ? h.p.z : 0.f;
  o[5] = hit ? h.n.x : 0.f, o[6] = hit ? h.n.y : 0.f, o[7] = hit ? h.n.z : 0.f;
  out_mat[i] = hit ? h.mat : 0xffffffffu;
}

template <uint32_t FEAT>
__global__ void debug_samples_kernel(DevScene sc, DevCamera cam, DevParams P, uint32_t n, const uint32_t* xs,
                                     const uint32_t* ys, const uint32_t* ss, float* out_rgb, uint32_t* out_info) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counts cnt = {0, 0, 0, 0};
  uint32_t bounces = 0, draws = 0;
  V3 c = sample_color<FEAT, true>(sc, cam, P, xs[i], ys[i], ss[i], cnt, bounces, draws);
  out_rgb[3 * i] = c.x, out_rgb[3 * i + 1] = c.y, out_rgb[3 * i + 2] = c.z;
  out_info[4 * i] = bounces, out_info[4 * i + 1] = draws, out_info[4 * i + 2] = cnt.aabb, out_info[4 * i + 3] = cnt.prim;
}

// print_ppm's to_u8 (lib.rs:348-352): sqrt, * 255.99, `as i32` (saturating, NaN -> 0), clamp 0..=255
__global__ void tonemap_kernel(size_t n, const float* __restrict__ rgb, uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 255.99f * __builtin_sqrtf(rgb[i]);
  int32_t q = f32_as_i32(v);
  out[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

__global__ void debug_math_kernel(int op, size_t n, const float* in, const float* in2, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i];
  float r;
  switch (op) {
    case 0: r = rt_logf(x); break;
    case 1: r = rt_pow5f(x); break;
    case 2: r = rt_sinf(x); break;
    case 3: r = __builtin_sqrtf(x); break;
    case 4: r = 1.f / x; break;
    default: r = x / in2[i]; break;
  }
  out[i] = r;
}

// ===================================================================================================
// Host side
// ===================================================================================================
namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(RTG_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)
}  // namespace

struct rtg_builder {
  SceneBuilder sb;
};

struct rtg_scene {
  int device = 0;
  DevScene dev{};
  uint32_t features = 0;
  uint32_t n_prog = 0, n_mat = 0, n_tex = 0;
  uint64_t bytes = 0;
  void* buffers[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long* d_counters = nullptr;  // [0..4] N/P/H/rays/draws, [7] = work-queue head (persistent kernel)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t own_stream = nullptr;  // rtg_par_cast_multi: this scene's launch stream (created on first use)
  int num_cus = 0;
  float* d_frame = nullptr;     // rtg_par_cast: device staging frame for host framebuffers
  size_t frame_bytes = 0;
  float* d_stack = nullptr;     // full-feature pool kernel: per-wave transform stacks
  size_t stack_bytes = 0;
  uint32_t* d_slots = nullptr;  // ray-pool path slots when they live in global memory
  size_t slots_bytes = 0;
  float* d_scratch = nullptr;  // chunk-mode per-sample colours
  size_t scratch_bytes = 0;
  int force_chunks = 0;        // RTG_CHUNKS: 0 = automatic
  uint64_t scratch_limit = 0;  // option scratch_mb: budget of the per-sample colour scratch in bytes, 0 = half of the free HBM
  bool whole_scratch = false;  // the kernel trace reads every sample colour back: one pass regardless of the budget
  LaunchConsts* d_consts = nullptr;  // camera / frame parameters / chunk description of the launch (rt_pool.h), written on the launch stream
  uint32_t* d_lpt = nullptr;   // cost-ordered work queue (rt_pool.h LptQueue)
  size_t lpt_bytes = 0;
  LptQueue lpt_desc{};         // descriptor of the last launch (RTG_VERBOSE histogram)
  int last_kernel = 0;         // what launch_render chose last: 1 = baseline, 3 = lean ray pools, 4 = full-feature ray pools
  uint32_t last_pix_work = 0;  // pixel work items of that launch (tiles x tile area), 0 when it kept no per-sample scratch
  int verbose = 0;
  int window = -1;             // full-feature kernel: records of the program staged in LDS (-1 = as many as fit)
  int ray_lds = 1;             // 0: all slot fields in global memory
  int force_rccl = 0;          // rtg_par_cast_multi: run the RCCL reduce even over ONE distinct device (a clique of one)
  int bvh4 = 0;                // 1: traverse the 4-wide collapse of the Bvh (same image, other counters; needs wide_bytes)
  uint32_t wide_bytes = 0;     // size of the 4-wide image in buffers[7], 0 = the scene has none
  int sync_full = -1;          // full-feature scenes on the pool-free lock-step kernel (rt_sync_full.h): -1 = when the program holds no BOX record, 0 / 1 = never / always
  uint32_t n_box = 0;          // BOX records of the flat program
  int lpt = 2;                 // RTG_LPT=0: natural order throughout; 1 / 2 = LptQueue::mode
  int lpt_deep = 4;            // RTG_LPT_DEEP: scatter events at bounce >= this make up a block's cost
  int lpt_shift = 0;           // RTG_LPT_SHIFT: merge cost classes in groups of 1 << shift
  int lpt_phase1 = 0;          // RTG_LPT_PHASE1: chunks in natural order, 0 = n_chunks / 8 clamped to [2, 8]
  // 3 = ray-pool kernels (rt_pool.h / rt_pool_full.h), 1 = one-lane-per-pixel baseline (rt_trace.h) for every scene
  int kernel_version = 3;
  PoolTuning pool_tune{40, 16, 24, 16, 16, 40};  // lean ray-pool kernel (refill_min, sphere_min, box_leave: profiles/r02_e_final/lean_knob_sweep.txt, middle of round 2)
  PoolTuning full_tune{20, 24, 32, 16, 24, 40};  // full-feature pool kernel (a service there also has hit records to move)
  PoolTuning sync_tune{20, 16, 32, 16, 16, 40};  // lock-step kernel (only run_ahead / run_ahead_min / gather_min matter there)
  int wg_per_cu = 0;                       // 0 = ask the occupancy API
  int full_threads = 0;                    // full-feature pool kernel: 0 = the variant's maximum (RTG_BLOCK overrides)
  int pool_threads = 1024;                 // lean ray-pool kernel: ONE 16-wave workgroup per CU shares one LDS copy of the program
};

// The per-sample colour scratch may take up to half of the free HBM (288 GB per MI355X).
static uint64_t scratch_cap() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 16ull << 30;
  return (uint64_t)free_b / 2;
}

// Sample passes.  The pool kernels park every sample colour in a [sample][pixel work index] scratch (12 B each) that the
// ordered fold consumes; a frame whose scratch would exceed the budget (option scratch_mb, default half of the free HBM)
// or whose work items would overflow the 32-bit queue counter is rendered in several passes over consecutive sample
// ranges, the fold kernel carrying the running per-pixel sum from pass to pass (rt_pool.h ChunkMode::s_begin) -- the
// same left fold, bit for bit, with O(budget) instead of O(spp) memory.  Returns the samples per pass (>= 1).
static uint32_t samples_per_pass(const rtg_scene* s, uint64_t pix_work, uint32_t ns) {
  const uint64_t per_sample = pix_work * 3 * sizeof(float);
  uint64_t budget = s->scratch_limit ? s->scratch_limit : std::max<uint64_t>(scratch_cap(), s->scratch_bytes);
  if (s->whole_scratch) budget = ~0ull;
  uint64_t k = std::max<uint64_t>(1, budget / std::max<uint64_t>(per_sample, 1));
  k = std::min<uint64_t>(k, 0xfffffffeull / std::max<uint64_t>(pix_work, 1));  // work items of a pass: one 32-bit counter
  k = std::max<uint64_t>(1, std::min<uint64_t>(k, ns));
  const uint64_t n_pass = (ns + k - 1) / k;
  return (uint32_t)((ns + n_pass - 1) / n_pass);  // balanced passes
}

static uint64_t owned_pixels(const DevParams& d);
static hipError_t grow(void** buf, size_t* have, size_t need) {
  if (need <= *have) return hipSuccess;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr, *have = 0;
  hipError_t e = hipMalloc(buf, need);
  if (e == hipSuccess) *have = need;
  return e;
}

static hipError_t setup_lpt(rtg_scene* s, ChunkMode& cm, uint64_t capacity, hipStream_t stream);

// Lean scenes, ray-pool kernel (rt_pool.h): one persistent 1024-thread workgroup per CU.
template <bool COUNT>
static hipError_t launch_pool(rtg_scene* s, const DevCamera& cam, const DevParams& d, float* d_out,
                              hipStream_t stream) {
  uint32_t tiles_x = (d.nx + d.tile_w - 1) / d.tile_w, tiles_y = (d.ny + d.tile_h - 1) / d.tile_h;
  uint32_t tiles = tiles_x * tiles_y;
  uint32_t owned = tiles > d.rank ? (tiles - d.rank + d.nranks - 1) / d.nranks : 0;
  const uint64_t pix_work = (uint64_t)owned * d.tile_w * d.tile_h;
  if (pix_work == 0) return hipSuccess;  // this rank owns no tile
  if (pix_work > 0xfffffffeull) return hipErrorInvalidValue;
  const int bt = s->pool_threads;
  const uint32_t waves = (uint32_t)bt / 64;
  // Sample-chunk mode (see rt_pool.h).  Default: one sample per work item.  Work items are then ~100x more numerous than
  // path slots, so the end-of-frame tail (slots finishing their last item while the queue is empty) is negligible;
  // measured on C2: 40.6 ms with one pixel (50 samples) per item, 23.5 ms with one sample per item.
  uint64_t n_chunks = d.ns;
  if (s->force_chunks > 0) n_chunks = (uint64_t)s->force_chunks;
  if (n_chunks > d.ns) n_chunks = d.ns;
  const bool use_scratch = n_chunks > 1;  // else (ns = 1, or option chunks = 1): a slot folds its pixel's samples itself
  uint32_t per_pass = d.ns, chunk = d.ns;
  if (use_scratch) {
    per_pass = samples_per_pass(s, pix_work, d.ns);
    chunk = (uint32_t)((d.ns + n_chunks - 1) / n_chunks);
    if (per_pass < d.ns) chunk = 1u;  // several passes: one sample per work item
    hipError_t ea = grow((void**)&s->d_scratch, &s->scratch_bytes, pix_work * per_pass * 3 * sizeof(float));
    if (ea != hipSuccess) return ea;
  }
  s->last_pix_work = use_scratch && per_pass == d.ns ? (uint32_t)pix_work : 0u;
  uint32_t* queue = (uint32_t*)(s->d_counters + 7);
  const size_t lds_limit = 160 * 1024;
  const bool wide = s->bvh4 && s->wide_bytes != 0;
  const uint32_t image = wide ? s->wide_bytes : s->dev.lds_image_bytes;
  DevScene dev = s->dev;
  if (wide) dev.lds_off = (const uint32_t*)s->buffers[7], dev.lds_image_bytes = s->wide_bytes;  // the WIDE kernel's reading of these two
  bool use_lds = image != 0 && pool_lds_bytes(image, s->n_mat, waves, true, false) <= lds_limit;
  if (wide && !use_lds) return hipErrorNotSupported;  // (rtg_scene_set_option refuses bvh4 for images that do not fit)
  // (hot slot fields in LDS keep best_pc as 16 bits = 14 bits of image offset / 8 + 2 bits of scatter tries: images < 128 KB)
  bool ray_lds = s->ray_lds && pool_lds_bytes(image, s->n_mat, waves, use_lds, true) <= lds_limit && (!use_lds || image < (1u << 17));
  size_t lds = pool_lds_bytes(image, s->n_mat, waves, use_lds, ray_lds);
  void (*kernel)(DevScene, const LaunchConsts*, float*, uint32_t, uint32_t*, unsigned long long*, PoolTuning, uint32_t*);
  if (wide) kernel = ray_lds ? render_lean_pool<true, COUNT, true, true> : render_lean_pool<true, COUNT, false, true>;
  else if (ray_lds) kernel = use_lds ? render_lean_pool<true, COUNT, true> : render_lean_pool<false, COUNT, true>;
  else kernel = use_lds ? render_lean_pool<true, COUNT, false> : render_lean_pool<false, COUNT, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int per_cu = s->wg_per_cu;
  if (per_cu <= 0) {
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, bt, lds);
    if (e != hipSuccess) return e;
  }
  if (per_cu < 1) per_cu = 1;
  for (uint32_t s0 = 0; s0 < d.ns; s0 += per_pass) {  // ONE pass unless the scratch budget is smaller than the frame's sample colours
    DevParams dp = d;
    dp.ns = std::min(d.ns, s0 + per_pass);  // the pass renders samples [s0, dp.ns)
    ChunkMode cm{};
    cm.scratch = nullptr, cm.chunk = d.ns, cm.n_chunks = 1, cm.pix_work = (uint32_t)pix_work, cm.s_begin = s0;
    if (use_scratch) {
      cm.chunk = chunk;
      cm.n_chunks = (dp.ns - s0 + chunk - 1) / chunk;
      cm.scratch = s->d_scratch - 3ull * s0 * pix_work;  // biased: sample s of work index w at scratch[3 * (s * pix_work + w)]
    }
    const uint64_t total_work = pix_work * cm.n_chunks;
    if (total_work > 0xfffffffeull) return hipErrorInvalidValue;
    e = hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream);  // re-initialise every pass
    if (e != hipSuccess) return e;
    // a wave keeps POOL paths in flight; do not launch more waves than there is work for
    uint64_t want = (total_work + (uint64_t)waves * POOL - 1) / ((uint64_t)waves * POOL);
    uint32_t grid = (uint32_t)std::min<uint64_t>(want ? want : 1, (uint64_t)s->num_cus * per_cu);
    e = setup_lpt(s, cm, (uint64_t)grid * waves * POOL, stream);
    if (e != hipSuccess) return e;
    if (s->verbose)
      fprintf(stderr, "[rtg] pool: samples [%u, %u) of %u: grid %u x %d threads, %d WG/CU, lds %zu B (program staged: %d, hot slot fields in LDS: %d), %u chunk(s) of %u samples, cost-ordered queue after %u chunk(s)\n",
              s0, dp.ns, d.ns, grid, bt, per_cu, lds, (int)use_lds, (int)ray_lds, cm.n_chunks, cm.chunk, (cm.lpt_samples - (cm.lpt_samples ? s0 : 0u)) / (cm.chunk ? cm.chunk : 1u));
    e = grow((void**)&s->d_slots, &s->slots_bytes, (size_t)grid * waves * POOL * POOL_FIELDS * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(write_launch_consts, dim3(1), dim3(1), 0, stream, s->d_consts, LaunchConsts{cam, dp, cm, make_pixmap(dp)});
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(bt), lds, stream, dev, (const LaunchConsts*)s->d_consts, d_out, (uint32_t)total_work, queue,
                       s->d_counters, s->pool_tune, s->d_slots);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (cm.scratch) {
      hipLaunchKernelGGL(fold_samples_kernel, dim3((uint32_t)((pix_work + 255) / 256)), dim3(256), 0, stream, dp, cm, make_pixmap(dp), d_out, d.ns);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// The 4-wide image of a lean program that is ONE Bvh over spheres (rt_pool.h WIDE; option `bvh4`): every node record holds
// the boxes of up to four GRANDCHILDREN of a node of the reference's tree (bvh.rs:22-81), in the reference's left-to-right
// order, so that a traversal step tests four boxes at once and the leaves are still met in the reference's order.  Returns
// false when the program has another shape (list-level objects, bare leaves, more than 8 levels).
static bool build_wide_image(const Packet* lo, const Packet* hi, size_t n, std::vector<uint32_t>& out) {
  struct T { uint32_t box; int left, right; uint32_t sphere; };  // left < 0: a leaf (box + sphere)
  std::vector<T> nodes;
  auto op_of = [&](size_t i) { return hi[i].w[3] & 0xffu; };
  if (n < 4 || op_of(n - 1) != OP_END || op_of(0) != OP_BOX || hi[0].w[2] != n - 1) return false;
  bool ok = true;
  std::function<int(size_t, size_t&)> parse = [&](size_t i, size_t& end) -> int {
    if (i >= n || op_of(i) != OP_BOX) { ok = false; end = i + 1; return -1; }
    end = hi[i].w[2];
    const int id = (int)nodes.size();
    nodes.push_back(T{(uint32_t)i, -1, -1, 0});
    if (op_of(i + 1) == OP_SPHERE) {
      if (end != i + 2) ok = false;
      nodes[id].sphere = (uint32_t)(i + 1);
      return id;
    }
    size_t e1 = 0, e2 = 0;
    const int l = parse(i + 1, e1);
    if (!ok || e1 >= end) { ok = false; return id; }
    const int r = parse(e1, e2);
    if (e2 != end) ok = false;
    nodes[id].left = l, nodes[id].right = r;
    return id;
  };
  size_t end0 = 0;
  const int root = parse(0, end0);
  if (!ok || root < 0 || nodes[root].left < 0) return false;
  out.clear();
  auto alloc = [&](uint32_t bytes) { const uint32_t at = (uint32_t)out.size() * 4u; out.resize(out.size() + bytes / 4u, 0u); return at; };
  std::function<uint32_t(int, uint32_t, uint32_t)> emit = [&](int t, uint32_t parent, uint32_t level) -> uint32_t {
    if (level > 7u) { ok = false; return 0; }
    const uint32_t at = alloc(WIDE_NODE_BYTES);
    int entry[4];
    uint32_t n_ch = 0;
    for (int x : {nodes[t].left, nodes[t].right}) {
      if (nodes[x].left < 0) entry[n_ch++] = x;
      else entry[n_ch++] = nodes[x].left, entry[n_ch++] = nodes[x].right;
    }
    uint32_t child[4] = {0, 0, 0, 0}, leafmask = 0;
    for (uint32_t k = 0; k < n_ch; k++) {
      const T e = nodes[entry[k]];
      const Packet bl = lo[e.box], bh = hi[e.box];  // (min.x, max.x, min.y, max.y) (min.z, max.z, ..)
      uint32_t* b = out.data() + at / 4u + 4u + 12u * k;  // (min, max) pairs, then (max, min) pairs: rt_pool.h
      b[0] = bl.w[0], b[1] = bl.w[1], b[2] = bl.w[2], b[3] = bl.w[3], b[4] = bh.w[0], b[5] = bh.w[1];
      b[6] = bl.w[1], b[7] = bl.w[0], b[8] = bl.w[3], b[9] = bl.w[2], b[10] = bh.w[1], b[11] = bh.w[0];
      if (e.left < 0) {
        const uint32_t sp = alloc(WIDE_SPHERE_BYTES);
        uint32_t* r = out.data() + sp / 4u;
        r[0] = hi[e.sphere].w[2], r[1] = hi[e.sphere].w[3];
        r[2] = lo[e.sphere].w[0], r[3] = lo[e.sphere].w[1], r[4] = lo[e.sphere].w[2], r[5] = lo[e.sphere].w[3];
        r[6] = at, r[7] = 0u;
        child[k] = sp, leafmask |= 1u << k;
      } else {
        child[k] = emit(entry[k], at, level + 1u);
      }
    }
    uint32_t* h = out.data() + at / 4u;
    h[0] = parent, h[1] = LDS_BOX_BIT | OP_BOX | (level << 8) | (n_ch << 12) | (leafmask << 16);
    h[2] = (child[0] >> 3) | ((child[1] >> 3) << 16), h[3] = (child[2] >> 3) | ((child[3] >> 3) << 16);
    return at;
  };
  emit(root, WIDE_NO_PARENT, 0u);
  return ok && out.size() * 4u < 512u * 1024u;
}

// Cost-ordered work queue (rt_pool.h, ChunkMode): enabled when the frame has enough chunks for a measuring
// phase and enough blocks to order; the buffers are re-zeroed on the launch stream every call.
static hipError_t setup_lpt(rtg_scene* s, ChunkMode& cm, uint64_t capacity, hipStream_t stream) {
  cm.lpt = nullptr, cm.lpt_samples = 0, cm.lpt_deep = 0;
  const uint32_t n_blocks = cm.pix_work / LPT_BLOCK;
  if (!s->lpt || !cm.scratch || cm.n_chunks < 6 || n_blocks < 64 || n_blocks > 65536 || cm.pix_work % LPT_BLOCK) return hipSuccess;
  // Phase 1 must outlast the first fill of the pools (`capacity` paths in flight) by enough for the
  // counts to mean something when the blocks are filed; it may take up to a third of the frame.
  uint32_t phase1 = std::max<uint32_t>(std::min(8u, std::max(2u, cm.n_chunks / 8u)), (uint32_t)((2 * capacity + cm.pix_work - 1) / cm.pix_work));
  if (s->lpt_phase1 > 0) phase1 = (uint32_t)s->lpt_phase1;
  if (phase1 < 1 || phase1 > cm.n_chunks / 3) return hipSuccess;
  // layout: [descriptor, 64 B] [cost n] [ctl LPT_CTL] [list LPT_CLASSES x n]
  const size_t words = 16 + (size_t)n_blocks * (1 + LPT_CLASSES) + LPT_CTL;
  hipError_t e = grow((void**)&s->d_lpt, &s->lpt_bytes, words * sizeof(uint32_t));
  if (e != hipSuccess) return e;
  LptQueue q;
  memset(&q, 0, sizeof(q));
  static_assert(sizeof(LptQueue) <= 64, "descriptor slot");
  q.cost = s->d_lpt + 16, q.ctl = q.cost + n_blocks, q.list = q.ctl + LPT_CTL;
  q.n_blocks = n_blocks, q.phase1 = phase1;
  q.phase2_base = phase1 * cm.pix_work;
  q.span = LPT_BLOCK * (cm.n_chunks - phase1);
  q.mode = (uint32_t)s->lpt, q.shift = (uint32_t)s->lpt_shift;
  // descriptor and zeroed counters travel on the launch stream: ordered with the render kernels before and after
  hipLaunchKernelGGL(write_lpt_descriptor, dim3(1), dim3(1), 0, stream, reinterpret_cast<LptQueue*>(s->d_lpt), q);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  s->lpt_desc = q;
  e = hipMemsetAsync(q.cost, 0, ((size_t)n_blocks + LPT_CTL) * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  cm.lpt = reinterpret_cast<const LptQueue*>(s->d_lpt);
  cm.lpt_samples = cm.s_begin + phase1 * cm.chunk;  // samples [s_begin, lpt_samples) of every pixel: phase 1 of this pass
  cm.lpt_deep = (uint32_t)s->lpt_deep;
  return hipSuccess;
}

// Full-feature scenes, ray-pool kernel (rt_pool_full.h) or, for list worlds without a Bvh, the lock-step kernel
// (rt_sync_full.h): always one sample per work item + ordered fold, in as many sample passes as the scratch budget asks for.
template <bool COUNT>
static hipError_t launch_full_pool(rtg_scene* s, const DevCamera& cam, const DevParams& d, float* d_out,
                                   hipStream_t stream) {
  uint32_t tiles_x = (d.nx + d.tile_w - 1) / d.tile_w, tiles_y = (d.ny + d.tile_h - 1) / d.tile_h;
  uint32_t tiles = tiles_x * tiles_y;
  uint32_t owned = tiles > d.rank ? (tiles - d.rank + d.nranks - 1) / d.nranks : 0;
  const uint64_t pix_work = (uint64_t)owned * d.tile_w * d.tile_h;
  if (pix_work == 0) return hipSuccess;
  if (pix_work > 0xfffffffeull) return hipErrorInvalidValue;
  const uint32_t per_pass = samples_per_pass(s, pix_work, d.ns);
  const bool tex = (s->features & FEAT_TEXTURE) != 0;
  // ONE 16-wave workgroup per CU shares one LDS copy of the program (128 VGPRs per lane).
  const int bt_max = tex ? RT_FULL_TEX_THREADS : 1024;
  const int bt = s->full_threads > 0 && s->full_threads <= bt_max ? s->full_threads : bt_max;
  const uint32_t waves = (uint32_t)bt / 64;
  hipError_t e = grow((void**)&s->d_scratch, &s->scratch_bytes, pix_work * per_pass * 3 * sizeof(float));
  if (e != hipSuccess) return e;
  s->last_pix_work = per_pass == d.ns ? (uint32_t)pix_work : 0u;
  uint32_t* queue = (uint32_t*)(s->d_counters + 7);
  // Program placement: the whole program in LDS when it fits, else a leading window
  // (depth-first order: the window holds whole leading subtrees) and global memory for the rest.
  const size_t list_bytes = full_pool_lds_bytes(0, waves);
  const size_t budget = 160 * 1024;  // all of a CU's LDS: one workgroup per CU
  uint32_t window = s->n_prog;
  int prog = 1;
  if ((size_t)window * 32 + list_bytes > budget) window = (uint32_t)((budget - list_bytes) / 32), prog = 2;
  if (s->window >= 0) {
    window = std::min<uint32_t>(s->n_prog, (uint32_t)s->window);
    prog = window == 0 ? 0 : (window == s->n_prog ? 1 : 2);
  }
  const size_t lds = full_pool_lds_bytes(window, waves);
  void (*kernel)(DevScene, const LaunchConsts*, float*, uint32_t, uint32_t*, unsigned long long*, PoolTuning, uint32_t*, float*,
                 uint32_t);
  const bool genb = (s->features & FEAT_BOUNDARY) != 0;  // a medium bounded by an object graph: the nested-walk variant
  if (genb) {
    if (prog == 0) kernel = tex ? render_full_pool<0, true, COUNT, true> : render_full_pool<0, false, COUNT, true>;
    else if (prog == 1) kernel = tex ? render_full_pool<1, true, COUNT, true> : render_full_pool<1, false, COUNT, true>;
    else kernel = tex ? render_full_pool<2, true, COUNT, true> : render_full_pool<2, false, COUNT, true>;
  } else if (prog == 0) kernel = tex ? render_full_pool<0, true, COUNT> : render_full_pool<0, false, COUNT>;
  else if (prog == 1) kernel = tex ? render_full_pool<1, true, COUNT> : render_full_pool<1, false, COUNT>;
  else kernel = tex ? render_full_pool<2, true, COUNT> : render_full_pool<2, false, COUNT>;
  e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int per_cu = s->wg_per_cu;
  if (per_cu <= 0) {
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, bt, lds);
    if (e != hipSuccess) return e;
  }
  if (per_cu < 1) per_cu = 1;
  const bool lock_step = (s->sync_full > 0 || (s->sync_full < 0 && s->n_box == 0)) && prog == 1;  // list world without a Bvh: one path per lane (rt_sync_full.h)
  void (*k2)(DevScene, const LaunchConsts*, float*, uint32_t, uint32_t*, unsigned long long*, PoolTuning, float*, uint32_t) = nullptr;
  const size_t lds2 = (size_t)window * 32;
  if (lock_step) {
    if (genb) k2 = tex ? render_full_sync<1, true, COUNT, true> : render_full_sync<1, false, COUNT, true>;
    else k2 = tex ? render_full_sync<1, true, COUNT, false> : render_full_sync<1, false, COUNT, false>;
    e = hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
  }
  for (uint32_t s0 = 0; s0 < d.ns; s0 += per_pass) {  // ONE pass unless the scratch budget is smaller than the frame's sample colours
    DevParams dp = d;
    dp.ns = std::min(d.ns, s0 + per_pass);
    ChunkMode cm{};
    cm.scratch = s->d_scratch - 3ull * s0 * pix_work, cm.chunk = 1u, cm.n_chunks = dp.ns - s0, cm.pix_work = (uint32_t)pix_work, cm.s_begin = s0;
    const uint64_t total_work = pix_work * cm.n_chunks;
    if (total_work > 0xfffffffeull) return hipErrorInvalidValue;
    e = hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    uint64_t want = (total_work + (uint64_t)waves * FPOOL - 1) / ((uint64_t)waves * FPOOL);
    uint32_t grid = (uint32_t)std::min<uint64_t>(want ? want : 1, (uint64_t)s->num_cus * per_cu);
    e = setup_lpt(s, cm, (uint64_t)grid * waves * FPOOL, stream);
    if (e != hipSuccess) return e;
    e = grow((void**)&s->d_slots, &s->slots_bytes, (size_t)grid * waves * FPOOL * FPOOL_FIELDS * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    e = grow((void**)&s->d_stack, &s->stack_bytes, (size_t)grid * waves * (genb ? 2 : 1) * MAX_XFORM_DEPTH * 6 * 64 * sizeof(float));
    if (e != hipSuccess) return e;
    if (s->verbose)
      fprintf(stderr, "[rtg] full pool: samples [%u, %u) of %u: grid %u x %d threads, %d WG/CU, lds %zu B (program window: %u of %u records), cost-ordered queue after %u chunk(s)\n",
              s0, dp.ns, d.ns, grid, bt, per_cu, lds, window, s->n_prog, cm.lpt_samples - (cm.lpt_samples ? s0 : 0u));
    hipLaunchKernelGGL(write_launch_consts, dim3(1), dim3(1), 0, stream, s->d_consts, LaunchConsts{cam, dp, cm, make_pixmap(dp)});
    if (lock_step)
      hipLaunchKernelGGL(k2, dim3(grid), dim3(bt), lds2, stream, s->dev, (const LaunchConsts*)s->d_consts, d_out, (uint32_t)total_work, queue,
                         s->d_counters, s->sync_tune, s->d_stack, window);
    else
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(bt), lds, stream, s->dev, (const LaunchConsts*)s->d_consts, d_out, (uint32_t)total_work, queue,
                         s->d_counters, s->full_tune, s->d_slots, s->d_stack, window);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_samples_kernel, dim3((uint32_t)((pix_work + 255) / 256)), dim3(256), 0, stream, dp, cm, make_pixmap(dp), d_out, d.ns);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <bool COUNT>
static hipError_t launch_render(rtg_scene* s, const DevCamera& cam, const DevParams& d, float* d_out,
                                hipStream_t stream) {
  // geometry / texture features pick the kernel; the albedo-range bits only say whether the pool kernels' "accum
  // is +0" argument holds (rt_pool.h PoolField)
  const uint32_t geom = s->features & (FEAT_ALL | FEAT_BOUNDARY);
  const bool accum_zero = !(s->features & FEAT_WIDE_ALBEDO) && (!(s->features & FEAT_BRIGHT_ALBEDO) || d.max_bounces <= 63u);
  // FEAT_DEEP: graph shapes only the general walk of the baseline kernel handles (flat_scene.h)
  const bool pool_ok = accum_zero && s->kernel_version >= 3 && d.nx <= 0xffffu && d.ny <= 0xffffu && !(s->features & FEAT_DEEP);
  s->last_kernel = 1;
  if (geom != 0 && pool_ok) {
    s->last_kernel = 4;
    return launch_full_pool<COUNT>(s, cam, d, d_out, stream);
  }
  if (geom == 0 && pool_ok) {
    s->last_kernel = 3;
    return launch_pool<COUNT>(s, cam, d, d_out, stream);
  }
  uint32_t nbx = (d.nx + 15) / 16, nby = (d.ny + 15) / 16;
  dim3 grid(nbx * nby), block(256);
  if (s->features & FEAT_DEEP)
    hipLaunchKernelGGL((render_kernel<FEAT_ALL | FEAT_DEEP, COUNT>), grid, block, 0, stream, s->dev, cam, d, d_out, s->d_counters);
  else if (geom == 0)
    hipLaunchKernelGGL((render_kernel<0u, COUNT>), grid, block, 0, stream, s->dev, cam, d, d_out, s->d_counters);
  else
    hipLaunchKernelGGL((render_kernel<FEAT_ALL, COUNT>), grid, block, 0, stream, s->dev, cam, d, d_out, s->d_counters);
  return hipGetLastError();
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

extern "C" {

const char* rtg_version(void) { return "rtiow-rust_amd 0.1 (gfx950 HIP; flat-program ray-pool kernels)"; }
const char* rtg_last_error(void) { return g_err.c_str(); }

int rtg_device_count(int* n) {
  if (!n) return fail(RTG_ERR_INVALID, "null argument");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  *n = c;
  return RTG_OK;
}

int rtg_builder_create(rtg_builder** out) {
  if (!out) return fail(RTG_ERR_INVALID, "null argument");
  *out = new rtg_builder();
  return RTG_OK;
}
void rtg_builder_destroy(rtg_builder* b) { delete b; }

// ---- textures ------------------------------------------------------------------------------------
static rtg_id bad(const char* msg) {
  fail(RTG_ERR_INVALID, msg);
  return RTG_INVALID_ID;
}

rtg_id rtg_texture_constant(rtg_builder* b, const float rgb[3]) {
  if (!b || !rgb) return bad("null argument");
  HostTexture t;
  t.kind = TEX_CONSTANT;
  t.rgb[0] = rgb[0], t.rgb[1] = rgb[1], t.rgb[2] = rgb[2];
  b->sb.textures.push_back(t);
  return (rtg_id)b->sb.textures.size() - 1;
}
rtg_id rtg_texture_checker(rtg_builder* b, rtg_id t0, rtg_id t1) {
  if (!b) return bad("null argument");
  if (t0 >= b->sb.textures.size() || t1 >= b->sb.textures.size()) return bad("checker: bad texture handle");
  HostTexture t;
  t.kind = TEX_CHECKER;
  t.t0 = t0, t.t1 = t1;
  b->sb.textures.push_back(t);
  return (rtg_id)b->sb.textures.size() - 1;
}
rtg_id rtg_texture_perlin(rtg_builder* b, float scale) {
  if (!b) return bad("null argument");
  if (!b->sb.has_perlin) return bad("perlin: call rtg_builder_set_perlin_tables first");
  HostTexture t;
  t.kind = TEX_PERLIN;
  t.scale = scale;
  b->sb.textures.push_back(t);
  return (rtg_id)b->sb.textures.size() - 1;
}
int rtg_builder_set_perlin_tables(rtg_builder* b, const float vecs[768], const uint8_t px[256],
                                  const uint8_t py[256], const uint8_t pz[256]) {
  if (!b || !vecs || !px || !py || !pz) return fail(RTG_ERR_INVALID, "null argument");
  for (int i = 0; i < 256; i++) {
    b->sb.perlin_vecs[4 * i] = vecs[3 * i], b->sb.perlin_vecs[4 * i + 1] = vecs[3 * i + 1];
    b->sb.perlin_vecs[4 * i + 2] = vecs[3 * i + 2], b->sb.perlin_vecs[4 * i + 3] = 0.f;
    b->sb.perlin_perm[i] = px[i], b->sb.perlin_perm[256 + i] = py[i], b->sb.perlin_perm[512 + i] = pz[i];
  }
  b->sb.has_perlin = true;
  return RTG_OK;
}

// ---- materials -----------------------------------------------------------------------------------
static rtg_id push_material(rtg_builder* b, uint32_t kind, rtg_id tex, const float* albedo, float param) {
  if (!b) return bad("null argument");
  HostMaterial m;
  m.kind = kind;
  if (tex != RTG_INVALID_ID) {
    if (tex >= b->sb.textures.size()) return bad("material: bad texture handle");
    m.tex = tex;
  }
  if (albedo) m.albedo[0] = albedo[0], m.albedo[1] = albedo[1], m.albedo[2] = albedo[2];
  m.param = param;
  b->sb.materials.push_back(m);
  return (rtg_id)b->sb.materials.size() - 1;
}
rtg_id rtg_material_lambertian(rtg_builder* b, rtg_id albedo) {
  if (albedo == RTG_INVALID_ID) return bad("material: bad texture handle");
  return push_material(b, MAT_LAMBERTIAN, albedo, nullptr, 0.f);
}
rtg_id rtg_material_metal(rtg_builder* b, const float albedo[3], float fuzz) {
  if (!albedo) return bad("null argument");
  return push_material(b, MAT_METAL, RTG_INVALID_ID, albedo, fuzz);
}
rtg_id rtg_material_dielectric(rtg_builder* b, float ref_idx) {
  return push_material(b, MAT_DIELECTRIC, RTG_INVALID_ID, nullptr, ref_idx);
}
rtg_id rtg_material_diffuse_light(rtg_builder* b, rtg_id emission, float brightness) {
  if (emission == RTG_INVALID_ID) return bad("material: bad texture handle");
  return push_material(b, MAT_DIFFUSE_LIGHT, emission, nullptr, brightness);
}
rtg_id rtg_material_isotropic(rtg_builder* b, rtg_id albedo) {
  if (albedo == RTG_INVALID_ID) return bad("material: bad texture handle");
  return push_material(b, MAT_ISOTROPIC, albedo, nullptr, 0.f);
}

// ---- objects -------------------------------------------------------------------------------------
static rtg_id push_object(rtg_builder* b, const HostObject& o) {
  if (!b) return bad("null argument");
  try {
    return b->sb.add_object(o);
  } catch (const BuildError& e) {
    fail(e.code, e.msg);
    return RTG_INVALID_ID;
  }
}
rtg_id rtg_object_sphere(rtg_builder* b, float radius, rtg_id material) {
  HostObject o;
  o.kind = HostObject::SPHERE;
  o.f[0] = radius;
  o.mat = material;
  return push_object(b, o);
}
rtg_id rtg_object_rect(rtg_builder* b, int axis, float r0s, float r0e, float r1s, float r1e, float k, rtg_id material) {
  HostObject o;
  o.kind = HostObject::RECT;
  o.axis = axis;
  o.f[0] = k, o.f[1] = r0s, o.f[2] = r0e, o.f[3] = r1s, o.f[4] = r1e;
  o.mat = material;
  return push_object(b, o);
}
static rtg_id wrapper(rtg_builder* b, HostObject::Kind kind, const float* v, rtg_id child) {
  HostObject o;
  o.kind = kind;
  if (v) o.f[0] = v[0], o.f[1] = v[1], o.f[2] = v[2];
  o.a = child;
  return push_object(b, o);
}
rtg_id rtg_object_flip_normals(rtg_builder* b, rtg_id object) { return wrapper(b, HostObject::FLIP, nullptr, object); }
rtg_id rtg_object_translate(rtg_builder* b, const float offset[3], rtg_id object) {
  if (!offset) return bad("null argument");
  return wrapper(b, HostObject::TRANSLATE, offset, object);
}
rtg_id rtg_object_scale(rtg_builder* b, const float factor[3], rtg_id object) {
  if (!factor) return bad("null argument");
  return wrapper(b, HostObject::SCALE, factor, object);
}
rtg_id rtg_object_rotate_y(rtg_builder* b, float degrees, rtg_id object) {
  // object.rs:477-484: radians = degrees * PI / 180; sin/cos via the platform libm (host-side setup)
  float radians = degrees * 3.14159265358979323846f / 180.f;
  float sc[3] = {sinf(radians), cosf(radians), 0.f};
  return wrapper(b, HostObject::ROTATE_Y, sc, object);
}
rtg_id rtg_object_and(rtg_builder* b, rtg_id o0, rtg_id o1) {
  HostObject o;
  o.kind = HostObject::AND;
  o.a = o0, o.b = o1;
  return push_object(b, o);
}
rtg_id rtg_object_rect_prism(rtg_builder* b, const float p0[3], const float p1[3], rtg_id m) {
  // object.rs:420-473: And(And(+Z, And(+Y, +X)), And(Flip(-Z), And(Flip(-Y), Flip(-X))))
  if (!b || !p0 || !p1) return bad("null argument");
  rtg_id zp = rtg_object_rect(b, 2, p0[0], p1[0], p0[1], p1[1], p1[2], m);
  rtg_id yp = rtg_object_rect(b, 1, p0[0], p1[0], p0[2], p1[2], p1[1], m);
  rtg_id xp = rtg_object_rect(b, 0, p0[1], p1[1], p0[2], p1[2], p1[0], m);
  if (zp == RTG_INVALID_ID || yp == RTG_INVALID_ID || xp == RTG_INVALID_ID) return RTG_INVALID_ID;
  rtg_id zn = rtg_object_flip_normals(b, rtg_object_rect(b, 2, p0[0], p1[0], p0[1], p1[1], p0[2], m));
  rtg_id yn = rtg_object_flip_normals(b, rtg_object_rect(b, 1, p0[0], p1[0], p0[2], p1[2], p0[1], m));
  rtg_id xn = rtg_object_flip_normals(b, rtg_object_rect(b, 0, p0[1], p1[1], p0[2], p1[2], p0[0], m));
  return rtg_object_and(b, rtg_object_and(b, zp, rtg_object_and(b, yp, xp)),
                        rtg_object_and(b, zn, rtg_object_and(b, yn, xn)));
}
rtg_id rtg_object_linear_move(rtg_builder* b, rtg_id object, const float motion[3]) {
  if (!motion) return bad("null argument");
  return wrapper(b, HostObject::MOVE, motion, object);
}
rtg_id rtg_object_constant_medium(rtg_builder* b, rtg_id boundary, float density, rtg_id material) {
  HostObject o;
  o.kind = HostObject::MEDIUM;
  o.f[0] = density;
  o.a = boundary;
  o.mat = material;
  return push_object(b, o);
}
rtg_id rtg_object_bvh(rtg_builder* b, const rtg_id* objects, size_t n, float e0, float e1) {
  if (!b || (!objects && n)) return bad("null argument");
  try {
    return b->sb.add_bvh(objects, n, e0, e1);
  } catch (const BuildError& e) {
    fail(e.code, e.msg);
    return RTG_INVALID_ID;
  }
}

rtg_id rtg_object_bvh_sah(rtg_builder* b, const rtg_id* objects, size_t n, float e0, float e1) {
  if (!b || (!objects && n)) return bad("null argument");
  try {
    return b->sb.add_bvh(objects, n, e0, e1, true);
  } catch (const BuildError& e) {
    fail(e.code, e.msg);
    return RTG_INVALID_ID;
  }
}

// ---- camera (camera.rs:18-50; host-side setup, tan from the platform libm) ---------------------
int rtg_camera_look(const float from[3], const float at[3], const float up[3], float fov, float aspect,
                    float aperture, float focus_dist, float e0, float e1, rtg_camera* out) {
  if (!from || !at || !up || !out) return fail(RTG_ERR_INVALID, "null argument");
  auto dot3 = [](const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
  auto unit = [&](float* v) {
    float len = sqrtf(dot3(v, v));
    v[0] = v[0] / len, v[1] = v[1] / len, v[2] = v[2] / len;
  };
  auto cross3 = [](const float* a, const float* b, float* r) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = -(a[0] * b[2] - a[2] * b[0]);
    r[2] = a[0] * b[1] - a[1] * b[0];
  };
  float lens_radius = aperture / 2.f;
  float theta = fov * 3.14159265358979323846f / 180.f;
  float half_height = tanf(theta / 2.f);
  float half_width = aspect * half_height;
  float w[3] = {from[0] - at[0], from[1] - at[1], from[2] - at[2]};
  unit(w);
  float u[3], v[3];
  cross3(up, w, u);
  unit(u);
  cross3(w, u, v);
  float hw_fd = half_width * focus_dist, hh_fd = half_height * focus_dist;
  float h2 = (2.f * half_width) * focus_dist, v2 = (2.f * half_height) * focus_dist;
  for (int i = 0; i < 3; i++) {
    out->origin[i] = from[i];
    out->lower_left_corner[i] = ((from[i] - hw_fd * u[i]) - hh_fd * v[i]) - focus_dist * w[i];
    out->horizontal[i] = h2 * u[i];
    out->vertical[i] = v2 * v[i];
    out->u[i] = u[i];
    out->v[i] = v[i];
  }
  out->lens_radius = lens_radius;
  out->exposure_start = e0;
  out->exposure_end = e1;
  return RTG_OK;
}

// ---- scene ---------------------------------------------------------------------------------------
static int upload(void** dst, const void* src, size_t bytes, uint64_t* total) {
  size_t alloc = bytes ? bytes : 16;
  HIP_TRY(hipMalloc(dst, alloc));
  if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  *total += alloc;
  return RTG_OK;
}

void rtg_scene_destroy(rtg_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (void* p : s->buffers)
    if (p) (void)hipFree(p);
  if (s->d_counters) (void)hipFree(s->d_counters);
  if (s->d_scratch) (void)hipFree(s->d_scratch);
  if (s->d_slots) (void)hipFree(s->d_slots);
  if (s->d_stack) (void)hipFree(s->d_stack);
  if (s->d_lpt) (void)hipFree(s->d_lpt);
  if (s->d_consts) (void)hipFree(s->d_consts);
  if (s->d_frame) (void)hipFree(s->d_frame);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  delete s;
}

int rtg_scene_create(rtg_builder* b, const rtg_id* world, size_t n, int device, rtg_scene** out) {
  if (!b || !out || (!world && n)) return fail(RTG_ERR_INVALID, "null argument");
  FlatScene fs;
  try {
    b->sb.flatten(world, n, &fs);
  } catch (const BuildError& e) {
    return fail(e.code, e.msg);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(RTG_ERR_DEVICE, "no HIP device: the rtiow hot path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(RTG_ERR_INVALID, "bad device index");
  HIP_TRY(hipSetDevice(device));
  rtg_scene* s = new rtg_scene();
  s->device = device;
  s->features = fs.features;
  s->n_prog = (uint32_t)fs.lo.size();
  s->n_mat = (uint32_t)fs.mat.size() / 2;
  s->n_tex = (uint32_t)fs.tex.size() / 2;
  int rc;
  if ((rc = upload(&s->buffers[0], fs.lo.data(), fs.lo.size() * 16, &s->bytes)) ||
      (rc = upload(&s->buffers[1], fs.hi.data(), fs.hi.size() * 16, &s->bytes)) ||
      (rc = upload(&s->buffers[2], fs.mat.data(), fs.mat.size() * 16, &s->bytes)) ||
      (rc = upload(&s->buffers[3], fs.tex.data(), fs.tex.size() * 16, &s->bytes)) ||
      (rc = upload(&s->buffers[4], fs.perlin_vecs.data(), sizeof(float) * 1024, &s->bytes)) ||
      (rc = upload(&s->buffers[5], fs.perlin_perm.data(), 768, &s->bytes))) {
    rtg_scene_destroy(s);
    return rc;
  }
  s->dev.lo = (const uint4*)s->buffers[0];
  s->dev.hi = (const uint4*)s->buffers[1];
  s->dev.mat = (const uint4*)s->buffers[2];
  s->dev.tex = (const uint4*)s->buffers[3];
  s->dev.perlin_vecs = (const float4*)s->buffers[4];
  s->dev.perlin_perm = (const uint8_t*)s->buffers[5];
  s->dev.n_prog = s->n_prog;
  s->dev.n_mat = s->n_mat;
  for (const Packet& h : fs.hi) s->n_box += (h.w[3] & 0xffu) == OP_BOX ? 1u : 0u;
  if ((fs.features & (FEAT_ALL | FEAT_BOUNDARY)) == 0) {  // lean program (BOX / SPHERE / END): layout of its LDS image (rt_pool.h)
    std::vector<uint32_t> ops(fs.hi.size()), off(fs.hi.size());
    for (size_t i = 0; i < fs.hi.size(); i++) ops[i] = fs.hi[i].w[3];
    s->dev.lds_image_bytes = lds_image_offsets(ops.data(), ops.size(), off.data());
    if ((rc = upload(&s->buffers[6], off.data(), off.size() * sizeof(uint32_t), &s->bytes))) {
      rtg_scene_destroy(s);
      return rc;
    }
    s->dev.lds_off = (const uint32_t*)s->buffers[6];
    std::vector<uint32_t> wide;
    if (build_wide_image(fs.lo.data(), fs.hi.data(), fs.hi.size(), wide)) {
      if ((rc = upload(&s->buffers[7], wide.data(), wide.size() * sizeof(uint32_t), &s->bytes))) {
        rtg_scene_destroy(s);
        return rc;
      }
      s->wide_bytes = (uint32_t)(wide.size() * sizeof(uint32_t));
    }
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount;
  if (s->num_cus <= 0) s->num_cus = 256;
  if (hipMalloc((void**)&s->d_counters, 32 * sizeof(unsigned long long)) != hipSuccess || hipMalloc((void**)&s->d_consts, sizeof(LaunchConsts)) != hipSuccess ||
      hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
    rtg_scene_destroy(s);
    return fail(RTG_ERR_DEVICE, "scene: counter/event allocation failed");
  }
  *out = s;
  return RTG_OK;
}


// Scheduling / measurement switches of ONE scene handle (none of them changes a bit of the result; every setting is
// covered by test_every_kernel_variant_and_schedule_gives_the_same_bits).  The library itself reads no environment
// variable for these: sweep tools and the test-suite set them through this call (capi.py forwards RTG_* variables).
int rtg_scene_set_option(rtg_scene* s, const char* name, int value) {
  if (!s || !name) return fail(RTG_ERR_INVALID, "null argument");
  const std::string k(name);
  const uint32_t u = (uint32_t)value;
  if (k == "kernel") s->kernel_version = value;                 // 3 = ray pools (default), 1 = one lane per pixel
  else if (k == "chunks") s->force_chunks = value;              // lean pool kernel: sample chunks per pixel, 0 = one sample per work item
  else if (k == "scratch_mb") s->scratch_limit = value > 0 ? (uint64_t)value << 20 : 0;  // budget of the per-sample colour scratch (0 = half of the free HBM): larger frames render in sample passes
  else if (k == "lpt") s->lpt = value;                          // cost-ordered queue: 0 off, 1 block-major, 2 class-major (default)
  else if (k == "lpt_phase1") s->lpt_phase1 = value;
  else if (k == "lpt_deep") s->lpt_deep = value;
  else if (k == "lpt_shift") s->lpt_shift = std::min(6, std::max(0, value));
  else if (k == "ray_lds") s->ray_lds = value;
  else if (k == "bvh4") {
    if (value && !s->wide_bytes) return fail(RTG_ERR_INVALID, "bvh4: the scene is not one Bvh of spheres (no 4-wide image)");
    if (value && pool_lds_bytes(s->wide_bytes, s->n_mat, (uint32_t)s->pool_threads / 64u, true, false) > 160 * 1024)
      return fail(RTG_ERR_INVALID, "bvh4: the 4-wide image does not fit a CU's 160 KB of LDS (the 4-wide walk only exists LDS-staged)");
    s->bvh4 = value;
  }
  else if (k == "force_rccl") s->force_rccl = value;
  else if (k == "sync") s->sync_full = value;
  else if (k == "block") {
    if (value < 64 || value > 1024 || value % 64) return fail(RTG_ERR_INVALID, "block: a multiple of 64 in [64, 1024]");
    s->pool_threads = s->full_threads = value;
  }
  else if (k == "wg_per_cu") s->wg_per_cu = value;
  else if (k == "verbose") s->verbose = value;                  // print launch geometry / schedule statistics to stderr
  else if (k == "window") s->window = value;                    // full-feature kernel: records staged in LDS, -1 = automatic
  else if (k == "box_leave") s->pool_tune.box_leave = s->full_tune.box_leave = s->sync_tune.box_leave = u;
  else if (k == "refill_min") s->pool_tune.refill_min = s->full_tune.refill_min = s->sync_tune.refill_min = u;
  else if (k == "gather_min") s->pool_tune.gather_min = s->full_tune.gather_min = s->sync_tune.gather_min = u;
  else if (k == "run_ahead") s->pool_tune.run_ahead = s->full_tune.run_ahead = s->sync_tune.run_ahead = u;
  else if (k == "run_ahead_min") s->pool_tune.run_ahead_min = s->full_tune.run_ahead_min = s->sync_tune.run_ahead_min = u;
  else if (k == "sphere_min") s->pool_tune.sphere_min = s->full_tune.sphere_min = s->sync_tune.sphere_min = u;
  else return fail(RTG_ERR_INVALID, "rtg_scene_set_option: unknown option '" + k + "'");
  return RTG_OK;
}

int rtg_scene_info(const rtg_scene* s, uint32_t* n_instructions, uint32_t* n_materials, uint32_t* n_textures,
                   uint64_t* hbm_bytes) {
  if (!s) return fail(RTG_ERR_INVALID, "null argument");
  if (n_instructions) *n_instructions = s->n_prog;
  if (n_materials) *n_materials = s->n_mat;
  if (n_textures) *n_textures = s->n_tex;
  if (hbm_bytes) *hbm_bytes = s->bytes;
  return RTG_OK;
}

// ---- render --------------------------------------------------------------------------------------
static DevCamera to_dev(const rtg_camera* c) {
  DevCamera d;
  auto v = [](const float* p) { return V3{p[0], p[1], p[2]}; };
  d.origin = v(c->origin), d.llc = v(c->lower_left_corner), d.horizontal = v(c->horizontal);
  d.vertical = v(c->vertical), d.u = v(c->u), d.v = v(c->v);
  d.lens_radius = c->lens_radius, d.e0 = c->exposure_start, d.e1 = c->exposure_end;
  return d;
}

static int check_params(const rtg_scene* s, const rtg_camera* camera, const rtg_params* p, DevParams* out) {
  if (!s || !camera || !p) return fail(RTG_ERR_INVALID, "null argument");
  if (p->struct_size != sizeof(rtg_params)) return fail(RTG_ERR_INVALID, "rtg_params.struct_size mismatch");
  if (p->nx == 0 || p->ny == 0 || p->ns == 0) return fail(RTG_ERR_INVALID, "nx, ny, ns must be > 0");
  if ((uint64_t)p->nx * p->ny > 0xffffffffull) return fail(RTG_ERR_INVALID, "image too large for 32-bit pixel index");
  if (!(camera->exposure_start < camera->exposure_end))  // camera.rs:55 / rand assert
    return fail(RTG_ERR_RANGE, "Uniform::sample_single called with low >= high");
  // rand 0.6.5's sample_single panics on non-finite bounds once its scale is not finite ("non-finite boundaries");
  // with scale = inf the retry loop of SampleRng::gen_range would never accept a value (a hung GPU)
  if (!std::isfinite(camera->exposure_start) || !std::isfinite(camera->exposure_end))
    return fail(RTG_ERR_RANGE, "Uniform::sample_single called with non-finite boundaries");
  if (!std::isfinite(camera->exposure_end - camera->exposure_start))  // rand would shrink the scale here; not restated
    return fail(RTG_ERR_RANGE, "exposure range wider than f32::MAX is not supported");
  DevParams d;
  d.nx = p->nx, d.ny = p->ny, d.ns = p->ns, d.max_bounces = p->max_bounces;
  d.t_near = p->t_near;
  d.seed_lo = (uint32_t)p->seed, d.seed_hi = (uint32_t)(p->seed >> 32);
  d.tile_w = p->tile_w ? p->tile_w : 16u;
  d.tile_h = p->tile_h ? p->tile_h : 16u;
  if (d.tile_w % 16u || d.tile_h % 16u) return fail(RTG_ERR_INVALID, "tile_w / tile_h must be multiples of 16");
  d.nranks = p->nranks ? p->nranks : 1u;
  d.rank = p->rank;
  if (d.rank >= d.nranks) return fail(RTG_ERR_INVALID, "rank >= nranks");
  *out = d;
  return RTG_OK;
}

static uint64_t owned_pixels(const DevParams& d) {
  uint64_t px = 0;
  uint32_t tiles_x = (d.nx + d.tile_w - 1) / d.tile_w, tiles_y = (d.ny + d.tile_h - 1) / d.tile_h;
  for (uint32_t ty = 0; ty < tiles_y; ty++)
    for (uint32_t tx = 0; tx < tiles_x; tx++) {
      if ((ty * tiles_x + tx) % d.nranks != d.rank) continue;
      uint32_t w = std::min(d.tile_w, d.nx - tx * d.tile_w), h = std::min(d.tile_h, d.ny - ty * d.tile_h);
      px += (uint64_t)w * h;
    }
  return px;
}

int rtg_par_cast_device(rtg_scene* s, const rtg_camera* camera, const rtg_params* params, float* d_out,
                        void* hip_stream, rtg_stats* stats) {
  DevParams d;
  int rc = check_params(s, camera, params, &d);
  if (rc) return rc;
  if (!d_out) return fail(RTG_ERR_INVALID, "null output");
  if (stats && stats->struct_size != sizeof(rtg_stats)) return fail(RTG_ERR_INVALID, "rtg_stats.struct_size mismatch");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  DevCamera cam = to_dev(camera);
  bool count = stats && (params->flags & RTG_FLAG_COUNTERS);
  if (count) {
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, 7 * sizeof(unsigned long long), stream));
    HIP_TRY(hipMemsetAsync(s->d_counters + 8, 0, 24 * sizeof(unsigned long long), stream));
  }
  if (stats) HIP_TRY(hipEventRecord(s->ev0, stream));
  HIP_TRY(count ? launch_render<true>(s, cam, d, d_out, stream) : launch_render<false>(s, cam, d, d_out, stream));
  if (stats) {
    HIP_TRY(hipEventRecord(s->ev1, stream));
    HIP_TRY(hipEventSynchronize(s->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->samples = owned_pixels(d) * d.ns;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (count) HIP_TRY(hipMemcpy(h, s->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    stats->aabb_tests = h[0], stats->prim_tests = h[1], stats->shaded_hits = h[2], stats->rays = h[3], stats->draws = h[4];
    if (count && s->verbose && s->lpt_desc.n_blocks && s->d_lpt) {  // cost classes of the last frame (class 0 = deepest)
      std::vector<uint32_t> ctl(LPT_CTL);
      HIP_TRY(hipMemcpy(ctl.data(), s->lpt_desc.ctl, LPT_CTL * sizeof(uint32_t), hipMemcpyDeviceToHost));
      std::string line;
      for (uint32_t c = 0; c < LPT_CLASSES; c++)
        if (ctl[c]) line += " " + std::to_string(c) + ":" + std::to_string(ctl[c]);
      fprintf(stderr, "[rtg] cost-ordered queue: %u of %u blocks filed, class:blocks%s\n", ctl[LPT_CLASSES], s->lpt_desc.n_blocks, line.c_str());
    }
    if (count && s->verbose) {
      unsigned long long q[24];
      HIP_TRY(hipMemcpy(q, s->d_counters + 8, sizeof(q), hipMemcpyDeviceToHost));
      if (q[18]) {  // lean pool kernel: per-wave timeline, scaled so that the longest wave = the measured kernel time
        const double us = (double)ms * 1000. / (double)q[16], n = (double)q[18];
        fprintf(stderr, "[rtg] wave timeline (us from its start): sees the work queue empty at min %.0f / mean %.0f / max %.0f; done at mean %.0f / max %.0f\n",
                (double)((1ull << 62) - q[19]) * us, (double)q[20] / n * us, (double)q[21] * us, (double)q[17] / n * us, (double)q[16] * us);
      }
      double tt = (double)(q[8] + q[9] + q[10] + q[11]);  // (q[9] = service minus shade)
      fprintf(stderr, "[rtg] wave-time shares (s_memtime, instrumented variant): shade %.1f%% gen+pull|service %.1f%% box %.1f%% sphere %.1f%%; "
              "per pass: shade %.0f, gen|service %.0f, box %.0f, sphere %.0f ticks\n", 100 * q[8] / tt, 100 * q[9] / tt, 100 * q[10] / tt,
              100 * q[11] / tt, q[4] ? (double)q[8] / q[4] : 0., q[4] ? (double)q[9] / q[4] : 0., q[0] ? (double)q[10] / q[0] : 0.,
              q[2] ? (double)q[11] / q[2] : 0.);
      if (q[15]) {  // full-feature pool kernel: the steps of a service
        unsigned long long f[2];
        HIP_TRY(hipMemcpy(f, s->d_counters + 5, sizeof(f), hipMemcpyDeviceToHost));
        fprintf(stderr, "[rtg] services %llu: finish step %.1f%% of wave time (%.0f ticks each), refill step %.1f%% (%.0f ticks per refill)\n", f[0],
                100 * f[1] / tt, f[0] ? (double)f[1] / f[0] : 0., 100 * q[15] / tt, q[6] ? (double)q[15] / q[6] : 0.);
      }
      fprintf(stderr, "[rtg] pool schedule: box steps %llu (avg %.1f lanes), sphere passes %llu (avg %.1f lanes), shade passes %llu "
                "(avg %.1f lanes), end / camera-ray passes %llu (avg %.1f lanes), refills %llu (avg %.1f lanes)\n", q[0], q[0] ? (double)q[1] / q[0] : 0.0, q[2],
                q[2] ? (double)q[3] / q[2] : 0.0, q[4], q[4] ? (double)q[5] / q[4] : 0.0, q[12], q[12] ? (double)q[13] / q[12] : 0.0, q[6],
                q[6] && q[7] ? (double)q[7] / q[6] : 0.0);
    }
  }
  return RTG_OK;
}

int rtg_par_cast(rtg_scene* s, const rtg_camera* camera, const rtg_params* params, float* out_rgb, rtg_stats* stats) {
  if (!s || !params || !out_rgb) return fail(RTG_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  size_t bytes = (size_t)params->nx * params->ny * 3 * sizeof(float);
  // the staging frame lives with the scene handle (no hipMalloc / hipFree per call)
  hipError_t e = grow((void**)&s->d_frame, &s->frame_bytes, bytes ? bytes : 16);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(framebuffer)");
  float* d_out = s->d_frame;
  // pixels of other ranks stay as the caller left them; a single rank overwrites every pixel
  const bool partial = params->nranks > 1;
  if (partial) e = hipMemcpy(d_out, out_rgb, bytes, hipMemcpyHostToDevice);
  int rc = (e == hipSuccess) ? rtg_par_cast_device(s, camera, params, d_out, nullptr, stats) : hip_fail(e, "hipMemcpy");
  if (rc == RTG_OK) {
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail(e, "render / copy back");
  }
  return rc;
}

// ---- single-process multi-GPU par_cast ---------------------------------------------------
// RCCL is dlopen()ed on first use (more than one distinct device, or the `force_rccl` option), so librtiow_gpu.so carries
// no link-time dependency on it and one-GPU hosts never load it.
namespace {
struct Rccl {
  void* lib = nullptr;
  std::string path;  // rtg_multi_reset: the library to load instead of the default search ("" = default)
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::map<std::vector<int>, std::vector<ncclComm_t>> comms;  // one clique per device list, created once, destroyed by rtg_multi_reset
  uint64_t n_reduces = 0;                                      // ncclReduce calls issued (rtg_multi_reset reports and clears it)
  std::mutex mu;
};
Rccl g_rccl;

// (g_rccl.mu held)
bool rccl_load(std::string* why) {
  if (g_rccl.lib) return true;
  void* h = nullptr;
  std::string tried;
  auto attempt = [&](const char* n, int flags) {
    if (h) return;
    (void)dlerror();
    h = dlopen(n, flags);
    if (!h && !(flags & RTLD_NOLOAD)) {
      const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
      tried += std::string(tried.empty() ? "" : "; ") + (e ? e : n);
    }
  };
  if (!g_rccl.path.empty()) {
    attempt(g_rccl.path.c_str(), RTLD_NOW | RTLD_LOCAL);
  } else {
    // an already-loaded librccl (e.g. the one a host framework ships) first, then the ROCm installation's
    for (const char* n : {"librccl.so", "librccl.so.1"}) attempt(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) attempt(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) {
    *why = "librccl not loadable: " + (tried.empty() ? std::string("?") : tried);
    return false;
  }
  auto sym = [&](const char* n) { return dlsym(h, n); };
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Reduce || !g_rccl.GetErrorString) {
    *why = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclReduce / ncclGetErrorString";
    dlclose(h);
    return false;
  }
  g_rccl.lib = h;
  return true;
}

// (g_rccl.mu held) destroy the cached cliques
void rccl_drop_comms() {
  if (g_rccl.CommDestroy)
    for (auto& kv : g_rccl.comms)
      for (ncclComm_t c : kv.second)
        if (c) (void)g_rccl.CommDestroy(c);
  g_rccl.comms.clear();
}

// ONE collective over the distinct devices: reduce(sum) of the float3 framebuffers heads[k]->d_frame (device devs[k],
// stream heads[k]->own_stream) to heads[0]'s.  On any failure the group is still closed and the communicators of this
// device list are dropped (their state is unknown); the caller synchronizes the streams.
int rccl_reduce_frames(const std::vector<int>& devs, const std::vector<rtg_scene*>& heads, size_t n_floats) {
  std::lock_guard<std::mutex> lock(g_rccl.mu);
  std::string why;
  if (!rccl_load(&why)) return fail(RTG_ERR_DEVICE, why);
  auto it = g_rccl.comms.find(devs);
  if (it == g_rccl.comms.end()) {
    std::vector<ncclComm_t> c(devs.size(), nullptr);
    ncclResult_t r = g_rccl.CommInitAll(c.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) return fail(RTG_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    it = g_rccl.comms.emplace(devs, std::move(c)).first;
  }
  std::string err;
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) {
    err = std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r);
  } else {
    for (size_t k = 0; k < devs.size() && err.empty(); k++) {
      hipError_t he = hipSetDevice(devs[k]);
      if (he != hipSuccess) {
        err = std::string("hipSetDevice: ") + hipGetErrorString(he);
        break;
      }
      r = g_rccl.Reduce(heads[k]->d_frame, heads[k]->d_frame, n_floats, ncclFloat, ncclSum, 0, it->second[k], heads[k]->own_stream);
      if (r != ncclSuccess) err = std::string("ncclReduce: ") + g_rccl.GetErrorString(r);
      else g_rccl.n_reduces++;
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();  // always: a group left open would swallow every later RCCL call of the process
    if (r2 != ncclSuccess && err.empty()) err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r2);
  }
  if (!err.empty()) {
    for (ncclComm_t c : it->second)
      if (c) (void)g_rccl.CommDestroy(c);
    g_rccl.comms.erase(it);
    return fail(RTG_ERR_DEVICE, err);
  }
  return RTG_OK;
}

__global__ void add_frames_kernel(size_t n, float* __restrict__ dst, const float* __restrict__ src) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = dst[i] + src[i];  // every pixel has ONE non-zero contributor: x + 0 is exact
}

int par_cast_multi_body(rtg_scene* const* scenes, int n_scenes, const rtg_camera* camera, const rtg_params* params, float* out_rgb,
                        rtg_stats* stats) {
  const size_t n_floats = (size_t)params->nx * params->ny * 3;
  const size_t bytes = n_floats * sizeof(float);
  const bool count = stats && (params->flags & RTG_FLAG_COUNTERS);
  // (1) every scene renders ITS tiles (tile % n_scenes == i) into its own zero-filled full frame, on its own stream
  for (int i = 0; i < n_scenes; i++) {
    rtg_scene* s = scenes[i];
    HIP_TRY(hipSetDevice(s->device));
    if (!s->own_stream) HIP_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    hipError_t e = grow((void**)&s->d_frame, &s->frame_bytes, bytes ? bytes : 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(framebuffer)");
    HIP_TRY(hipMemsetAsync(s->d_frame, 0, bytes, s->own_stream));
    rtg_params p = *params;
    p.rank = (uint32_t)i, p.nranks = (uint32_t)n_scenes;
    DevParams d;
    int rc = check_params(s, camera, &p, &d);
    if (rc) return rc;
    if (count) {
      HIP_TRY(hipMemsetAsync(s->d_counters, 0, 7 * sizeof(unsigned long long), s->own_stream));
      HIP_TRY(hipMemsetAsync(s->d_counters + 8, 0, 24 * sizeof(unsigned long long), s->own_stream));
    }
    HIP_TRY(hipEventRecord(s->ev0, s->own_stream));
    const DevCamera cam = to_dev(camera);
    HIP_TRY(count ? launch_render<true>(s, cam, d, s->d_frame, s->own_stream) : launch_render<false>(s, cam, d, s->d_frame, s->own_stream));
    HIP_TRY(hipEventRecord(s->ev1, s->own_stream));
  }
  // (2) scenes that share a device with an earlier one are summed there; one frame per DISTINCT device remains
  std::vector<int> devs;          // distinct devices in order of first appearance
  std::vector<rtg_scene*> heads;  // the scene holding each device's partial frame
  bool force_rccl = false;
  for (int i = 0; i < n_scenes; i++) {
    rtg_scene* s = scenes[i];
    force_rccl = force_rccl || s->force_rccl != 0;
    size_t k = 0;
    while (k < devs.size() && devs[k] != s->device) k++;
    if (k == devs.size()) {
      devs.push_back(s->device), heads.push_back(s);
      continue;
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->own_stream));
    hipLaunchKernelGGL(add_frames_kernel, dim3((uint32_t)((n_floats + 255) / 256)), dim3(256), 0, heads[k]->own_stream, n_floats,
                       heads[k]->d_frame, s->d_frame);
    HIP_TRY(hipGetLastError());
  }
  // (3) ONE collective over the distinct devices: reduce(sum) of the float3 framebuffer to the first device (xGMI).
  // `force_rccl`: also with ONE distinct device (a clique of one) -- the same dlopen / ncclCommInitAll / grouped in-place
  // ncclReduce code as on a node, which one-GPU test boxes could otherwise never execute.
  if (devs.size() > 1 || force_rccl) {
    int rc = rccl_reduce_frames(devs, heads, n_floats);
    if (rc) return rc;
  }
  // (4) wait, copy the assembled frame out, gather stats (kernel time = the slowest shard)
  for (rtg_scene* h : heads) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->own_stream));
  }
  HIP_TRY(hipSetDevice(heads[0]->device));
  HIP_TRY(hipMemcpy(out_rgb, heads[0]->d_frame, bytes, hipMemcpyDeviceToHost));
  if (stats) {
    rtg_stats total{};
    total.struct_size = sizeof(rtg_stats);
    for (int i = 0; i < n_scenes; i++) {
      rtg_scene* s = scenes[i];
      HIP_TRY(hipSetDevice(s->device));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
      total.kernel_ms = std::max(total.kernel_ms, ms);
      rtg_params p = *params;
      p.rank = (uint32_t)i, p.nranks = (uint32_t)n_scenes;
      DevParams d;
      (void)check_params(s, camera, &p, &d);
      total.samples += owned_pixels(d) * d.ns;
      if (count) {
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpy(h, s->d_counters, sizeof(h), hipMemcpyDeviceToHost));
        total.aabb_tests += h[0], total.prim_tests += h[1], total.shaded_hits += h[2], total.rays += h[3], total.draws += h[4];
      }
    }
    *stats = total;
  }
  return RTG_OK;
}
}  // namespace

int rtg_par_cast_multi(rtg_scene* const* scenes, int n_scenes, const rtg_camera* camera, const rtg_params* params,
                       float* out_rgb, rtg_stats* stats) {
  if (!scenes || n_scenes <= 0 || !camera || !params || !out_rgb) return fail(RTG_ERR_INVALID, "null argument");
  if (params->struct_size != sizeof(rtg_params)) return fail(RTG_ERR_INVALID, "rtg_params.struct_size mismatch");
  if (params->nranks > 1u) return fail(RTG_ERR_INVALID, "rtg_par_cast_multi shards by itself: params.rank / nranks must be 0 / 0|1");
  if (stats && stats->struct_size != sizeof(rtg_stats)) return fail(RTG_ERR_INVALID, "rtg_stats.struct_size mismatch");
  for (int i = 0; i < n_scenes; i++) {
    if (!scenes[i]) return fail(RTG_ERR_INVALID, "null scene handle");
    // one handle = one frame, one work queue, one stream: the same handle twice would wipe its own tiles
    for (int k = 0; k < i; k++)
      if (scenes[k] == scenes[i]) return fail(RTG_ERR_INVALID, "rtg_par_cast_multi: the same scene handle appears twice (one handle per shard)");
  }
  const int rc = par_cast_multi_body(scenes, n_scenes, camera, params, out_rgb, stats);
  if (rc != RTG_OK) {
    // whatever was queued before the failure must not outlive the call (the caller may free out_rgb, destroy the handles
    // or call again): drain every stream that may hold work, keeping the first error message
    const std::string first = g_err;
    for (int i = 0; i < n_scenes; i++) {
      if (!scenes[i]->own_stream) continue;
      if (hipSetDevice(scenes[i]->device) == hipSuccess) (void)hipStreamSynchronize(scenes[i]->own_stream);
    }
    g_err = first;
  }
  return rc;
}

int rtg_multi_reset(const char* rccl_library_or_null, uint64_t* n_reduces_or_null) {
  std::lock_guard<std::mutex> lock(g_rccl.mu);
  if (n_reduces_or_null) *n_reduces_or_null = g_rccl.n_reduces;
  g_rccl.n_reduces = 0;
  rccl_drop_comms();
  if (g_rccl.lib) dlclose(g_rccl.lib);
  g_rccl.lib = nullptr;
  g_rccl.CommInitAll = nullptr, g_rccl.CommDestroy = nullptr, g_rccl.GroupStart = nullptr, g_rccl.GroupEnd = nullptr;
  g_rccl.Reduce = nullptr, g_rccl.GetErrorString = nullptr;
  g_rccl.path = rccl_library_or_null ? rccl_library_or_null : "";
  return RTG_OK;
}

// ---- probes --------------------------------------------------------------------------------------
int rtg_debug_hit_top(rtg_scene* s, size_t n, const float* rays, uint64_t seed, float t_near, float* out,
                      uint32_t* out_material) {
  if (!s || !rays || !out || !out_material) return fail(RTG_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  DevBuf<float> d_rays, d_out;
  DevBuf<uint32_t> d_mat;
  HIP_TRY(d_rays.alloc(7 * n));
  HIP_TRY(d_out.alloc(8 * n));
  HIP_TRY(d_mat.alloc(n));
  HIP_TRY(hipMemcpy(d_rays.p, rays, 7 * n * sizeof(float), hipMemcpyHostToDevice));
  dim3 grid((uint32_t)((n + 63) / 64)), block(64);
  if (n) {
    if (s->features & FEAT_DEEP)
      hipLaunchKernelGGL((debug_hit_top_kernel<FEAT_ALL | FEAT_DEEP>), grid, block, 0, 0, s->dev, (uint32_t)n, d_rays.p, (uint32_t)seed,
                         (uint32_t)(seed >> 32), t_near, d_out.p, d_mat.p);
    else if (s->features == 0)
      hipLaunchKernelGGL((debug_hit_top_kernel<0u>), grid, block, 0, 0, s->dev, (uint32_t)n, d_rays.p, (uint32_t)seed,
                         (uint32_t)(seed >> 32), t_near, d_out.p, d_mat.p);
    else
      hipLaunchKernelGGL((debug_hit_top_kernel<FEAT_ALL>), grid, block, 0, 0, s->dev, (uint32_t)n, d_rays.p,
                         (uint32_t)seed, (uint32_t)(seed >> 32), t_near, d_out.p, d_mat.p);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_out.p, 8 * n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_material, d_mat.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RTG_OK;
}

int rtg_debug_samples(rtg_scene* s, const rtg_camera* camera, const rtg_params* params, size_t n, const uint32_t* xs,
                      const uint32_t* ys, const uint32_t* samples, float* out_rgb, uint32_t* out_info) {
  DevParams d;
  int rc = check_params(s, camera, params, &d);
  if (rc) return rc;
  if (!xs || !ys || !samples || !out_rgb || !out_info) return fail(RTG_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (params->flags & RTG_FLAG_TRACE_KERNEL) {
    // Trace the PRODUCTION kernel: render the whole frame with the instrumented variant of whatever kernel par_cast
    // uses for this scene, with its per-sample trace table switched on, then pick the requested keys out of the table
    // (colour: the per-sample scratch the ordered fold reads).
    const uint32_t tiles_x = (d.nx + d.tile_w - 1) / d.tile_w, tiles_y = (d.ny + d.tile_h - 1) / d.tile_h;
    const uint32_t tiles = tiles_x * tiles_y;
    const uint64_t owned = tiles > d.rank ? (tiles - d.rank + d.nranks - 1) / d.nranks : 0;
    const uint64_t pix_work = owned * d.tile_w * d.tile_h;
    // per-slot accumulators, indexed by (workgroup * waves + wave): a CU holds at most 32 waves of these kernels, and the
    // `wg_per_cu` option may ask for more (queued) workgroups of up to 16 waves each
    const size_t trace_waves = (size_t)s->num_cus * std::max(32, s->wg_per_cu > 0 ? s->wg_per_cu * 16 : 0);
    const size_t table = (size_t)4 * d.ns * pix_work, slots = trace_waves * std::max(FPOOL, POOL) * 3;
    DevBuf<uint32_t> d_trace;
    HIP_TRY(d_trace.alloc(table + slots));
    HIP_TRY(hipMemset(d_trace.p, 0, (table + slots) * sizeof(uint32_t)));
    HIP_TRY(grow((void**)&s->d_frame, &s->frame_bytes, (size_t)d.nx * d.ny * 3 * sizeof(float)));
    HIP_TRY(hipMemset(s->d_counters, 0, 32 * sizeof(unsigned long long)));
    const unsigned long long ptrs[2] = {(unsigned long long)(uintptr_t)d_trace.p, (unsigned long long)(uintptr_t)(d_trace.p + table)};
    HIP_TRY(hipMemcpy(s->d_counters + 30, ptrs, sizeof(ptrs), hipMemcpyHostToDevice));
    const DevCamera cam = to_dev(camera);
    s->whole_scratch = true;  // the trace reads every sample colour back: one sample pass
    hipError_t e = launch_render<true>(s, cam, d, s->d_frame, nullptr);
    s->whole_scratch = false;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipMemset(s->d_counters + 30, 0, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "trace launch");
    if (s->last_kernel < 3 || s->last_pix_work == 0)
      return fail(RTG_ERR_UNSUPPORTED, "this scene / frame runs on the baseline kernel (or without the per-sample scratch): nothing to trace");
    std::vector<uint32_t> h_trace(table);
    std::vector<float> h_col((size_t)3 * d.ns * pix_work);
    HIP_TRY(hipMemcpy(h_trace.data(), d_trace.p, table * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_col.data(), s->d_scratch, h_col.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
      const uint32_t x = xs[i], row = d.ny - 1u - ys[i], sm = samples[i];
      const uint32_t tx = x / d.tile_w, ty = row / d.tile_h, tile = ty * tiles_x + tx;
      if (x >= d.nx || ys[i] >= d.ny || sm >= d.ns || tile % d.nranks != d.rank) return fail(RTG_ERR_INVALID, "trace key outside this rank's frame");
      const uint32_t k = tile / d.nranks, lx = x - tx * d.tile_w, ly = row - ty * d.tile_h;  // = pixel_to_work (rt_pool.h)
      const uint32_t b = (ly >> 3) * (d.tile_w >> 3) + (lx >> 3);
      const size_t w = (size_t)k * d.tile_w * d.tile_h + b * 64u + (ly & 7u) * 8u + (lx & 7u);
      const size_t at = (size_t)sm * pix_work + w;
      for (int c = 0; c < 3; c++) out_rgb[3 * i + c] = h_col[3 * at + c];
      for (int c = 0; c < 4; c++) out_info[4 * i + c] = h_trace[4 * at + c];
    }
    return RTG_OK;
  }
  DevBuf<uint32_t> dx, dy, ds, dinfo;
  DevBuf<float> drgb;
  HIP_TRY(dx.alloc(n));
  HIP_TRY(dy.alloc(n));
  HIP_TRY(ds.alloc(n));
  HIP_TRY(dinfo.alloc(4 * n));
  HIP_TRY(drgb.alloc(3 * n));
  HIP_TRY(hipMemcpy(dx.p, xs, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dy.p, ys, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds.p, samples, n * 4, hipMemcpyHostToDevice));
  DevCamera cam = to_dev(camera);
  dim3 grid((uint32_t)((n + 63) / 64)), block(64);
  if (n) {
    if (s->features & FEAT_DEEP)
      hipLaunchKernelGGL((debug_samples_kernel<FEAT_ALL | FEAT_DEEP>), grid, block, 0, 0, s->dev, cam, d, (uint32_t)n, dx.p, dy.p, ds.p,
                         drgb.p, dinfo.p);
    else if (s->features == 0)
      hipLaunchKernelGGL((debug_samples_kernel<0u>), grid, block, 0, 0, s->dev, cam, d, (uint32_t)n, dx.p, dy.p, ds.p,
                         drgb.p, dinfo.p);
    else
      hipLaunchKernelGGL((debug_samples_kernel<FEAT_ALL>), grid, block, 0, 0, s->dev, cam, d, (uint32_t)n, dx.p, dy.p,
                         ds.p, drgb.p, dinfo.p);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_rgb, drgb.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_info, dinfo.p, 4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RTG_OK;
}

int rtg_tonemap_device(int device, size_t n, const float* d_rgb, uint8_t* d_out, void* hip_stream) {
  if (!d_rgb || !d_out) return fail(RTG_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(device));
  if (n) hipLaunchKernelGGL(tonemap_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, n, d_rgb, d_out);
  HIP_TRY(hipGetLastError());
  return RTG_OK;
}

int rtg_tonemap(int device, size_t n, const float* rgb, uint8_t* out) {
  if (!rgb || !out) return fail(RTG_ERR_INVALID, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(RTG_ERR_DEVICE, "no HIP device");
  HIP_TRY(hipSetDevice(device));
  DevBuf<float> di;
  DevBuf<uint8_t> dout;
  HIP_TRY(di.alloc(n));
  HIP_TRY(dout.alloc(n));
  HIP_TRY(hipMemcpy(di.p, rgb, n * sizeof(float), hipMemcpyHostToDevice));
  int rc = rtg_tonemap_device(device, n, di.p, dout.p, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, n, hipMemcpyDeviceToHost));
  return RTG_OK;
}

int rtg_debug_flatten(rtg_builder* b, const rtg_id* world, size_t n, uint32_t* n_instructions, uint32_t* features,
                      uint32_t* words_out, size_t capacity) {
  if (!b || (!world && n)) return fail(RTG_ERR_INVALID, "null argument");
  FlatScene fs;
  try {
    b->sb.flatten(world, n, &fs);
  } catch (const BuildError& e) {
    return fail(e.code, e.msg);
  }
  if (n_instructions) *n_instructions = (uint32_t)fs.lo.size();
  if (features) *features = fs.features;
  if (words_out) {
    size_t m = fs.lo.size() < capacity ? fs.lo.size() : capacity;
    for (size_t i = 0; i < m; i++) {
      std::memcpy(words_out + 8 * i, fs.lo[i].w, 16);
      std::memcpy(words_out + 8 * i + 4, fs.hi[i].w, 16);
    }
  }
  return RTG_OK;
}

int rtg_debug_math(int device, int op, size_t n, const float* in, const float* in2, float* out) {
  if (!in || !out || op < 0 || op > 5 || (op == 5 && !in2)) return fail(RTG_ERR_INVALID, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(RTG_ERR_DEVICE, "no HIP device");
  HIP_TRY(hipSetDevice(device));
  DevBuf<float> di, di2, dout;
  HIP_TRY(di.alloc(n));
  HIP_TRY(di2.alloc(n));
  HIP_TRY(dout.alloc(n));
  HIP_TRY(hipMemcpy(di.p, in, n * sizeof(float), hipMemcpyHostToDevice));
  if (in2) HIP_TRY(hipMemcpy(di2.p, in2, n * sizeof(float), hipMemcpyHostToDevice));
  if (n) hipLaunchKernelGGL(debug_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, op, n, di.p, di2.p, dout.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return RTG_OK;
}

}  // extern "C"
