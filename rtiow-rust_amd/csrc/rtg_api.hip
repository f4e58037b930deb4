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
#include <tuple>
#include <vector>

#include "../../include/rtiow_gpu.h"
#include "rt_pool.h"
#include "rt_pool_full.h"
#include "rt_sync_full.h"
#include "rt_pool2.h"
#include "rt_trace.h"
#include "scene_builder.h"

using namespace rtg;

#include "rtg_kernels.inc"

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

// Everything ONE frame in flight owns: work-queue head and statistics, launch constants, cost-ordered queue, per-sample scratch,
// path slots / stacks, timing events.  A scene handle keeps a ring of these (option frames_in_flight, default 1); a call takes
// the next one and first waits for the frame that used it last (`done`), so asynchronous calls on one handle are always safe
// and, with more than one context, overlap.
constexpr int RTG_MAX_FRAMES = 4;
struct LaunchCtx {
  unsigned long long* d_counters = nullptr;  // [0..4] N/P/H/rays/draws, [7] = work-queue head (persistent kernel), [8..] schedule statistics
  hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
  bool busy = false;            // `done` was recorded and not waited for yet
  float* d_stack = nullptr;     // full-feature pool kernel: per-wave transform stacks
  size_t stack_bytes = 0;
  uint32_t* d_slots = nullptr;  // ray-pool path slots when they live in global memory
  size_t slots_bytes = 0;
  float* d_scratch = nullptr;   // per-sample colours (sample passes: rtg_launch.inc)
  size_t scratch_bytes = 0;
  LaunchConsts* d_consts = nullptr;  // camera / frame parameters / chunk description of the launch (rt_pool.h), written on the launch stream
  uint32_t* d_lpt = nullptr;    // cost-ordered work queue (rt_pool.h LptQueue)
  size_t lpt_bytes = 0;
  LptQueue lpt_desc{};          // descriptor of the last launch (RTG_VERBOSE histogram)
  int last_kernel = 0;          // what launch_render chose last: 1 = baseline, 3 = lean ray pools, 4 = full-feature ray pools
  uint32_t last_pix_work = 0;   // pixel work items of that launch (tiles x tile area), 0 when it kept no per-sample scratch
};

struct rtg_scene {
  int device = 0;
  DevScene dev{};
  uint32_t features = 0;
  uint32_t n_prog = 0, n_mat = 0, n_tex = 0;
  uint64_t bytes = 0;
  void* buffers[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const uint32_t* d_parent = nullptr;  // buffers[8]: the wrapper around every program record (rt_pool_full.h rebuild_hit)
  hipStream_t own_stream = nullptr;  // rtg_par_cast_multi: this scene's launch stream (created on first use)
  int num_cus = 0;
  float* d_frame = nullptr;     // rtg_par_cast: device staging frame for host framebuffers
  size_t frame_bytes = 0;
  int force_chunks = 0;        // RTG_CHUNKS: 0 = automatic
  uint64_t scratch_limit = 0;  // option scratch_mb: budget of the per-sample colour scratch in bytes, 0 = half of the free HBM
  bool whole_scratch = false;  // the kernel trace reads every sample colour back: one pass regardless of the budget
  int verbose = 0;
  int window = -1;             // full-feature kernel: records of the program staged in LDS (-1 = as many as fit)
  int ray_lds = 1;             // 0: all slot fields in global memory
  int force_rccl = 0;          // rtg_par_cast_multi: run the RCCL reduce even over ONE distinct device (a clique of one)
  int multi_gather = 0;        // rtg_par_cast_multi: 1 = the packed collective (every scene ships its own tiles only; rtg_multi.inc) instead of the full-frame reduce
  float* d_pack = nullptr;     // ... this scene's tiles, packed in work-item order (on its device)
  size_t pack_bytes = 0;
  float* d_recv = nullptr;     // ... and where they arrive on the first device
  size_t recv_bytes = 0;
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
  int pool_threads = 0;                    // lean ray-pool kernel: 0 = ONE 16-wave workgroup per CU shares one LDS copy of the program (RTG_BLOCK overrides)
  int mat_lds = 1;                          // full-feature kernels: material records staged in LDS when they fit (0 = always from global memory)
  int deep_sized = 1;                       // FEAT_DEEP graphs: the general walk instantiated for the graph's real depths (0 = always 32 wrappers x 3 media levels)
  int hoist = 1;                            // full-feature pool kernel: evaluate the hoisted segment when a ray is created (flat_scene.h OP_SEG); 0 = the walk executes its records
  uint32_t seg_first = 0, seg_end = 0;      // ... its records, as record indices (0, 0: the program has none)
  int drain_share = 1;                      // pool kernels, drain-phase work sharing (rt_pool_full.h RT_DRAIN_SHARE): 0 = off
  // The second program and its kernel (rt_pool2.h; flat_scene.h "the list level, hoisted"): n_prog2 = 0 when the world has another shape
  int pool2 = 1;                            // 1 (default): programs with a second program run on the pool-2 kernel; 0: on the first full-feature kernel
  uint32_t n_prog2 = 0, n_box2 = 0;
  DevScene dev2{};                          // lo / hi = the second program (buffers[9], [10]); materials, textures, Perlin tables shared
  const P2Table* d_p2 = nullptr;            // buffers[11]
  Pool2Tuning pool2_tune{24, 48, 40, 4, 8, 24, 2};  // refill_min, box_leave, park_max, t_sphere, t_prism, t_list, t_push
  int small_frames = 1;                    // rtg_launch.inc pool_geometry: frames smaller than the chip get small workgroups and reservations
  LaunchCtx ctx[RTG_MAX_FRAMES];           // frames in flight
  int n_ctx = 1, next_ctx = 0;
  LaunchCtx* cx = &ctx[0];                 // the context of the call being made (ctx_acquire)
};

// A context that leaves the ring (frames_in_flight lowered) or whose scene is destroyed: wait for its frame, free what it
// holds (the sample scratch alone may be a large part of the HBM).  The events and the small buffers stay for reuse.
static void ctx_free_buffers(LaunchCtx* c) {
  if (c->busy) (void)hipEventSynchronize(c->done);
  c->busy = false;
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_slots) (void)hipFree(c->d_slots);
  if (c->d_stack) (void)hipFree(c->d_stack);
  if (c->d_lpt) (void)hipFree(c->d_lpt);
  c->d_scratch = nullptr, c->scratch_bytes = 0, c->d_slots = nullptr, c->slots_bytes = 0;
  c->d_stack = nullptr, c->stack_bytes = 0, c->d_lpt = nullptr, c->lpt_bytes = 0;
}

// Take the next launch context of the ring: create its small buffers on first use, wait for the frame that used it last.
static int ctx_acquire(rtg_scene* s) {
  LaunchCtx* c = &s->ctx[s->next_ctx];
  s->next_ctx = (s->next_ctx + 1) % s->n_ctx;
  if (!c->d_counters) {
    if (hipMalloc((void**)&c->d_counters, 64 * sizeof(unsigned long long)) != hipSuccess || hipMemset(c->d_counters, 0, 64 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&c->d_consts, sizeof(LaunchConsts)) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess)
      return fail(RTG_ERR_DEVICE, "scene: launch-context allocation failed");
  }
  if (c->busy) {
    hipError_t e = hipEventSynchronize(c->done);
    c->busy = false;
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(previous frame of this launch context)");
  }
  s->cx = c;
  return RTG_OK;
}
// ... and mark it in flight on `stream` once its work is enqueued
static int ctx_release(rtg_scene* s, hipStream_t stream) {
  hipError_t e = hipEventRecord(s->cx->done, stream);
  if (e != hipSuccess) return hip_fail(e, "hipEventRecord(done)");
  s->cx->busy = true;
  return RTG_OK;
}

#include "rtg_launch.inc"

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

extern "C" {

const char* rtg_version(void) { return "rtiow-rust_amd 0.2 (gfx950 HIP; flat-program ray-pool kernels; flat-program encoding 3: OP_SEG = 9, OP_SAVE / OP_MERGE / OP_EXT = 10 / 11 / 12, OP_LIST = 13, MEDIUM lo.y = 1 / density)"; }
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
  for (LaunchCtx& c : s->ctx) {
    ctx_free_buffers(&c);
    if (c.d_counters) (void)hipFree(c.d_counters);
    if (c.d_consts) (void)hipFree(c.d_consts);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    if (c.done) (void)hipEventDestroy(c.done);
  }
  if (s->d_frame) (void)hipFree(s->d_frame);
  if (s->d_pack) (void)hipFree(s->d_pack);
  if (s->d_recv) (void)hipFree(s->d_recv);
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
  {  // the wrapper around every record (rt_pool_full.h rebuild_hit): PUSH / POP pairs nest in program order, boundary streams included
    std::vector<uint32_t> parent(fs.hi.size(), 0xffffffffu), open;
    for (size_t i = 0; i < fs.hi.size(); i++) {
      const uint32_t op = fs.hi[i].w[3] & 0xffu;
      if (op == OP_POP && !open.empty()) open.pop_back();
      parent[i] = open.empty() ? 0xffffffffu : open.back();
      if (op == OP_PUSH) open.push_back((uint32_t)i);
    }
    if ((rc = upload(&s->buffers[8], parent.data(), parent.size() * sizeof(uint32_t), &s->bytes))) {
      rtg_scene_destroy(s);
      return rc;
    }
    s->d_parent = (const uint32_t*)s->buffers[8];
  }
  for (size_t i = 0; i < fs.hi.size(); i++)
    if ((fs.hi[i].w[3] & 0xffu) == OP_SEG) s->seg_first = (uint32_t)i + 1u, s->seg_end = fs.hi[i].w[2];
  if (!fs.hi2.empty()) {  // the second program (pool-2 kernel)
    if ((rc = upload(&s->buffers[9], fs.lo2.data(), fs.lo2.size() * 16, &s->bytes)) ||
        (rc = upload(&s->buffers[10], fs.hi2.data(), fs.hi2.size() * 16, &s->bytes)) ||
        (rc = upload(&s->buffers[11], &fs.p2, sizeof(P2Table), &s->bytes))) {
      rtg_scene_destroy(s);
      return rc;
    }
    s->dev2 = s->dev;
    s->dev2.lo = (const uint4*)s->buffers[9], s->dev2.hi = (const uint4*)s->buffers[10];
    s->dev2.n_prog = s->n_prog2 = (uint32_t)fs.hi2.size();
    s->dev2.lds_off = nullptr, s->dev2.lds_image_bytes = 0;
    s->d_p2 = (const P2Table*)s->buffers[11];
    for (const Packet& h : fs.hi2) s->n_box2 += (h.w[3] & 0xffu) == OP_BOX ? 1u : 0u;
  }
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
  if ((rc = ctx_acquire(s))) {  // the first launch context now: a scene that cannot allocate it is no scene
    rtg_scene_destroy(s);
    return rc;
  }
  s->next_ctx = 0;
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
    if (value && pool_lds_bytes(s->wide_bytes, s->n_mat, (uint32_t)(s->pool_threads > 0 ? s->pool_threads : RT_POOL_MAX_THREADS) / 64u, true, false) > 160 * 1024)
      return fail(RTG_ERR_INVALID, "bvh4: the 4-wide image does not fit a CU's 160 KB of LDS (the 4-wide walk only exists LDS-staged)");
    s->bvh4 = value;
  }
  else if (k == "frames_in_flight") {  // launch contexts of this handle: asynchronous rtg_par_cast_device calls overlap up to this many frames
    if (value < 1 || value > RTG_MAX_FRAMES) return fail(RTG_ERR_INVALID, "frames_in_flight: 1 .. 4");
    HIP_TRY(hipSetDevice(s->device));
    for (int i = value; i < RTG_MAX_FRAMES; i++) ctx_free_buffers(&s->ctx[i]);  // contexts that leave the ring: wait for their frame, give the HBM back
    s->n_ctx = value, s->next_ctx = 0;
  }
  else if (k == "force_rccl") s->force_rccl = value;
  else if (k == "multi_gather") s->multi_gather = value;        // rtg_par_cast_multi: the packed collective (1 / n of the bytes per scene) instead of the reduce
  else if (k == "sync") s->sync_full = value;
  else if (k == "block") {
    if (value < 64 || value > 1024 || value % 64) return fail(RTG_ERR_INVALID, "block: a multiple of 64 in [64, 1024]");
    s->pool_threads = s->full_threads = value;
  }
  else if (k == "wg_per_cu") s->wg_per_cu = value;
  else if (k == "drain_share") s->drain_share = value;
  else if (k == "hoist") s->hoist = value;
  else if (k == "pool2") s->pool2 = value;                      // 0: the first full-feature pool kernel also for programs the pool-2 kernel walks (A/B switch)
  else if (k == "deep_sized") s->deep_sized = value;
  else if (k == "mat_lds") s->mat_lds = value;
  else if (k == "small_frames") s->small_frames = value;        // 0: one geometry for every frame size (measurement switch)
  else if (k == "verbose") s->verbose = value;                  // print launch geometry / schedule statistics to stderr
  else if (k == "window") s->window = value;                    // full-feature kernel: records staged in LDS, -1 = automatic
  else if (k == "box_leave") s->pool_tune.box_leave = s->full_tune.box_leave = s->sync_tune.box_leave = u;
  else if (k == "refill_min") s->pool_tune.refill_min = s->full_tune.refill_min = s->sync_tune.refill_min = u;
  else if (k == "gather_min") s->pool_tune.gather_min = s->full_tune.gather_min = s->sync_tune.gather_min = u;
  else if (k == "run_ahead") s->pool_tune.run_ahead = s->full_tune.run_ahead = s->sync_tune.run_ahead = u;
  else if (k == "run_ahead_min") s->pool_tune.run_ahead_min = s->full_tune.run_ahead_min = s->sync_tune.run_ahead_min = u;
  else if (k == "sphere_min") s->pool_tune.sphere_min = s->full_tune.sphere_min = s->sync_tune.sphere_min = u;
  else if (k == "p2_refill") s->pool2_tune.refill_min = u;      // pool-2 kernel's schedule thresholds (rt_pool2.h Pool2Tuning)
  else if (k == "p2_box_leave") s->pool2_tune.box_leave = u;
  else if (k == "p2_park") s->pool2_tune.park_max = u;
  else if (k == "p2_sphere") s->pool2_tune.t_sphere = std::max(1u, u);
  else if (k == "p2_prism") s->pool2_tune.t_prism = std::max(1u, u);
  else if (k == "p2_list") s->pool2_tune.t_list = std::max(1u, u);
  else if (k == "p2_push") s->pool2_tune.t_push = std::max(1u, u);
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
  if (d.tile_w % 8u || d.tile_h % 8u) return fail(RTG_ERR_INVALID, "tile_w / tile_h must be multiples of 8");
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
  if ((rc = ctx_acquire(s))) return rc;
  DevCamera cam = to_dev(camera);
  bool count = stats && (params->flags & RTG_FLAG_COUNTERS);
  // From here on work of this frame may sit on `stream`: a failure must not hand the context out again while kernels of the
  // partly enqueued frame still run (they read d_consts / d_lpt / the scratch the next call would rewrite) -- drain first.
#define HIP_TRY_CTX(expr)                               \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) {                             \
      (void)hipStreamSynchronize(stream);               \
      return hip_fail(e_, #expr);                       \
    }                                                   \
  } while (0)
  if (count) {
    HIP_TRY_CTX(hipMemsetAsync(s->cx->d_counters, 0, 7 * sizeof(unsigned long long), stream));
    HIP_TRY_CTX(hipMemsetAsync(s->cx->d_counters + 8, 0, 24 * sizeof(unsigned long long), stream));
    HIP_TRY_CTX(hipMemsetAsync(s->cx->d_counters + 32, 0, 32 * sizeof(unsigned long long), stream));  // (RT_CENSUS builds)
  }
#ifdef RT_TIMELINE  // diagnostic build (rt_pool.h RT_TL_*): one 16-dword drain record per wave, dumped to $RTG_TIMELINE_OUT
  static uint32_t* d_tl = nullptr;
  const size_t tl_bytes = (size_t)1024 * 16 * 16 * sizeof(uint32_t);  // <= 1024 workgroups x 16 waves
  if (!count) {
    if (!d_tl) HIP_TRY_CTX(hipMalloc((void**)&d_tl, tl_bytes));
    HIP_TRY_CTX(hipMemsetAsync(d_tl, 0, tl_bytes, stream));
    static unsigned long long tl_ptr;
    tl_ptr = (unsigned long long)(uintptr_t)d_tl;
    HIP_TRY_CTX(hipMemcpyAsync(s->cx->d_counters + 31, &tl_ptr, sizeof(tl_ptr), hipMemcpyHostToDevice, stream));
  }
#endif
  if (stats) HIP_TRY_CTX(hipEventRecord(s->cx->ev0, stream));
  HIP_TRY_CTX(count ? launch_render<true>(s, cam, d, d_out, stream) : launch_render<false>(s, cam, d, d_out, stream));
  if ((rc = ctx_release(s, stream))) {
    (void)hipStreamSynchronize(stream);
    return rc;
  }
#undef HIP_TRY_CTX
  if (stats) {
    HIP_TRY(hipEventRecord(s->cx->ev1, stream));
    HIP_TRY(hipEventSynchronize(s->cx->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->cx->ev0, s->cx->ev1));
#ifdef RT_TIMELINE
    if (!count && getenv("RTG_TIMELINE_OUT")) {
      std::vector<uint32_t> h_tl(tl_bytes / sizeof(uint32_t));
      HIP_TRY(hipMemcpy(h_tl.data(), d_tl, tl_bytes, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(getenv("RTG_TIMELINE_OUT"), "wb")) {
        fwrite(&ms, sizeof(ms), 1, f);
        fwrite(h_tl.data(), 1, tl_bytes, f);
        fclose(f);
      }
    }
#endif
    stats->kernel_ms = ms;
    stats->samples = owned_pixels(d) * d.ns;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (count) HIP_TRY(hipMemcpy(h, s->cx->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    stats->aabb_tests = h[0], stats->prim_tests = h[1], stats->shaded_hits = h[2], stats->rays = h[3], stats->draws = h[4];
    if (count && s->verbose && s->cx->lpt_desc.n_blocks && s->cx->d_lpt) {  // cost classes of the last frame (class 0 = deepest)
      std::vector<uint32_t> ctl(LPT_CTL);
      HIP_TRY(hipMemcpy(ctl.data(), s->cx->lpt_desc.ctl, LPT_CTL * sizeof(uint32_t), hipMemcpyDeviceToHost));
      std::string line;
      for (uint32_t c = 0; c < LPT_CLASSES; c++)
        if (ctl[c]) line += " " + std::to_string(c) + ":" + std::to_string(ctl[c]);
      fprintf(stderr, "[rtg] cost-ordered queue: %u of %u blocks filed, class:blocks%s\n", ctl[LPT_CLASSES], s->cx->lpt_desc.n_blocks, line.c_str());
    }
    if (count && s->verbose) {
      unsigned long long q[24];
      HIP_TRY(hipMemcpy(q, s->cx->d_counters + 8, sizeof(q), hipMemcpyDeviceToHost));
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
        HIP_TRY(hipMemcpy(f, s->cx->d_counters + 5, sizeof(f), hipMemcpyDeviceToHost));
        fprintf(stderr, "[rtg] services %llu: finish step %.1f%% of wave time (%.0f ticks each), refill step %.1f%% (%.0f ticks per refill)\n", f[0],
                100 * f[1] / tt, f[0] ? (double)f[1] / f[0] : 0., 100 * q[15] / tt, q[6] ? (double)q[15] / q[6] : 0.);
      }
#ifdef RT_CENSUS
      {
        unsigned long long c[32];
        HIP_TRY(hipMemcpy(c, s->cx->d_counters + 32, sizeof(c), hipMemcpyDeviceToHost));
        static const char* nm[13] = {"BOX", "END", "no-ray", "held", "?", "?", "SPHERE", "RECT", "PUSH", "POP", "MEDIUM", "PRISM", "BEND"};
        for (int k = 0; k < 2; k++) {
          std::string line;
          char buf[64];
          for (int j = 0; j < 13; j++)
            if (c[16 * k + j]) snprintf(buf, sizeof buf, " %s %.1f", nm[j], (double)c[16 * k + j] / (double)c[16 * k + 15]), line += buf;
          fprintf(stderr, "[rtg] lane census at %s (%llu):%s\n", k ? "slow-pass iterations" : "box steps", c[16 * k + 15], line.c_str());
        }
      }
#endif
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

#include "rtg_multi.inc"

#include "rtg_probes.inc"

}  // extern "C"
