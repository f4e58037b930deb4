// ORACLE -- TEST INFRASTRUCTURE ONLY (see rto_core.hpp header; parity unpinned).
//
// rto_capi.cpp: the oracle's C entry points.  They mirror include/rtiow_gpu.h name for name
// (rtg_* -> rto_*) with identical POD layouts, so the parity tests drive both libraries with the
// same calls and diff the results.  Also: par_cast / cast drivers (lib.rs:321-397) and probes.
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>

#include "rto_scene.hpp"

using namespace rto;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace

extern "C" {

typedef uint32_t rto_id;
#define RTO_INVALID_ID 0xffffffffu

struct rto_camera {
  float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
  float lens_radius, exposure_start, exposure_end;
};
struct rto_params {
  uint32_t struct_size, nx, ny, ns, max_bounces;
  float t_near;
  uint64_t seed;
  uint32_t tile_w, tile_h, rank, nranks, flags, reserved;
};
struct rto_stats {
  uint32_t struct_size;
  float kernel_ms;
  uint64_t samples, aabb_tests, prim_tests, shaded_hits, rays, draws;
};

struct rto_builder {
  std::vector<TexturePtr> textures;
  std::vector<MaterialPtr> materials;
  std::vector<ObjectPtr> objects;
  std::shared_ptr<PerlinTables> perlin;
  Bvh::TieAudit ties;  // oracle-only: rto_builder_bvh_ties
};
struct rto_scene {
  World world;
};

const char* rto_last_error(void) { return g_err.c_str(); }
const char* rto_version(void) { return "rtiow oracle (CPU restatement) 0.1"; }

int rto_builder_create(rto_builder** out) {
  *out = new rto_builder();
  return 0;
}
void rto_builder_destroy(rto_builder* b) { delete b; }

static Vec3 v3(const float* p) { return Vec3(p[0], p[1], p[2]); }

rto_id rto_texture_constant(rto_builder* b, const float rgb[3]) {
  auto t = std::make_shared<Texture>();
  t->kind = Texture::CONSTANT;
  t->color = v3(rgb);
  b->textures.push_back(t);
  return (rto_id)b->textures.size() - 1;
}
rto_id rto_texture_checker(rto_builder* b, rto_id t0, rto_id t1) {
  if (t0 >= b->textures.size() || t1 >= b->textures.size()) {
    fail(-1, "checker: bad texture id");
    return RTO_INVALID_ID;
  }
  auto t = std::make_shared<Texture>();
  t->kind = Texture::CHECKER;
  t->t0 = b->textures[t0];
  t->t1 = b->textures[t1];
  b->textures.push_back(t);
  return (rto_id)b->textures.size() - 1;
}
rto_id rto_texture_perlin(rto_builder* b, float scale) {
  if (!b->perlin) {
    fail(-1, "perlin: tables not set");
    return RTO_INVALID_ID;
  }
  auto t = std::make_shared<Texture>();
  t->kind = Texture::PERLIN;
  t->scale = scale;
  t->tables = b->perlin;
  b->textures.push_back(t);
  return (rto_id)b->textures.size() - 1;
}
int rto_builder_set_perlin_tables(rto_builder* b, const float vecs[768], const uint8_t px[256],
                                  const uint8_t py[256], const uint8_t pz[256]) {
  auto t = std::make_shared<PerlinTables>();
  for (int i = 0; i < 256; i++) {
    t->vecs[i] = Vec3(vecs[3 * i], vecs[3 * i + 1], vecs[3 * i + 2]);
    t->perm_x[i] = px[i];
    t->perm_y[i] = py[i];
    t->perm_z[i] = pz[i];
  }
  b->perlin = t;
  return 0;
}

static rto_id push_mat(rto_builder* b, MaterialPtr m) {
  m->id = (uint32_t)b->materials.size();
  b->materials.push_back(m);
  return m->id;
}
static TexturePtr tex(rto_builder* b, rto_id t) {
  if (t >= b->textures.size()) return nullptr;
  return b->textures[t];
}
rto_id rto_material_lambertian(rto_builder* b, rto_id albedo) {
  auto m = std::make_shared<Material>();
  m->kind = Material::LAMBERTIAN;
  if (!(m->tex = tex(b, albedo))) return fail(-1, "bad texture"), RTO_INVALID_ID;
  return push_mat(b, m);
}
rto_id rto_material_metal(rto_builder* b, const float albedo[3], float fuzz) {
  auto m = std::make_shared<Material>();
  m->kind = Material::METAL;
  m->albedo = v3(albedo);
  m->fuzz = fuzz;
  return push_mat(b, m);
}
rto_id rto_material_dielectric(rto_builder* b, float ref_idx) {
  auto m = std::make_shared<Material>();
  m->kind = Material::DIELECTRIC;
  m->ref_idx = ref_idx;
  return push_mat(b, m);
}
rto_id rto_material_diffuse_light(rto_builder* b, rto_id emission, float brightness) {
  auto m = std::make_shared<Material>();
  m->kind = Material::DIFFUSE_LIGHT;
  if (!(m->tex = tex(b, emission))) return fail(-1, "bad texture"), RTO_INVALID_ID;
  m->brightness = brightness;
  return push_mat(b, m);
}
rto_id rto_material_isotropic(rto_builder* b, rto_id albedo) {
  auto m = std::make_shared<Material>();
  m->kind = Material::ISOTROPIC;
  if (!(m->tex = tex(b, albedo))) return fail(-1, "bad texture"), RTO_INVALID_ID;
  return push_mat(b, m);
}

static rto_id push_obj(rto_builder* b, ObjectPtr o) {
  b->objects.push_back(o);
  return (rto_id)b->objects.size() - 1;
}
static ObjectPtr obj(rto_builder* b, rto_id o) {
  if (o >= b->objects.size()) return nullptr;
  return b->objects[o];
}
static MaterialPtr mat(rto_builder* b, rto_id m) {
  if (m >= b->materials.size()) return nullptr;
  return b->materials[m];
}

rto_id rto_object_sphere(rto_builder* b, float radius, rto_id material) {
  auto s = std::make_shared<Sphere>();
  s->radius = radius;
  if (!(s->material = mat(b, material))) return fail(-1, "bad material"), RTO_INVALID_ID;
  return push_obj(b, s);
}
static std::shared_ptr<Rect> make_rect(int axis, float a0, float a1, float b0, float b1, float k,
                                       MaterialPtr m) {
  auto r = std::make_shared<Rect>();
  r->axis = axis;
  r->range0 = Range{a0, a1};
  r->range1 = Range{b0, b1};
  r->k = k;
  r->material = m;
  return r;
}
rto_id rto_object_rect(rto_builder* b, int axis, float a0, float a1, float b0, float b1, float k,
                       rto_id material) {
  if (axis < 0 || axis > 2) return fail(-1, "bad axis"), RTO_INVALID_ID;
  auto m = mat(b, material);
  if (!m) return fail(-1, "bad material"), RTO_INVALID_ID;
  return push_obj(b, make_rect(axis, a0, a1, b0, b1, k, m));
}
rto_id rto_object_flip_normals(rto_builder* b, rto_id o) {
  auto f = std::make_shared<FlipNormals>();
  if (!(f->object = obj(b, o))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, f);
}
rto_id rto_object_translate(rto_builder* b, const float offset[3], rto_id o) {
  auto t = std::make_shared<Translate>();
  t->offset = v3(offset);
  if (!(t->object = obj(b, o))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, t);
}
rto_id rto_object_scale(rto_builder* b, const float factor[3], rto_id o) {
  auto t = std::make_shared<Scale>();
  t->factor = v3(factor);
  if (!(t->object = obj(b, o))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, t);
}
rto_id rto_object_rotate_y(rto_builder* b, float degrees, rto_id o) {
  auto t = std::make_shared<RotateY>();
  float radians = degrees * 3.14159265358979323846f / 180.f;  // object.rs:478
  t->sin_theta = std::sin(radians);
  t->cos_theta = std::cos(radians);
  if (!(t->object = obj(b, o))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, t);
}
rto_id rto_object_and(rto_builder* b, rto_id o0, rto_id o1) {
  auto t = std::make_shared<And>();
  if (!(t->a = obj(b, o0)) || !(t->b = obj(b, o1))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, t);
}
// object.rs:420-473
rto_id rto_object_rect_prism(rto_builder* b, const float p0[3], const float p1[3], rto_id material) {
  auto m = mat(b, material);
  if (!m) return fail(-1, "bad material"), RTO_INVALID_ID;
  auto flip = [](ObjectPtr o) {
    auto f = std::make_shared<FlipNormals>();
    f->object = o;
    return std::static_pointer_cast<Object>(f);
  };
  auto both = [](ObjectPtr x, ObjectPtr y) {
    auto a = std::make_shared<And>();
    a->a = x;
    a->b = y;
    return std::static_pointer_cast<Object>(a);
  };
  ObjectPtr zp = make_rect(2, p0[0], p1[0], p0[1], p1[1], p1[2], m);
  ObjectPtr yp = make_rect(1, p0[0], p1[0], p0[2], p1[2], p1[1], m);
  ObjectPtr xp = make_rect(0, p0[1], p1[1], p0[2], p1[2], p1[0], m);
  ObjectPtr zn = flip(make_rect(2, p0[0], p1[0], p0[1], p1[1], p0[2], m));
  ObjectPtr yn = flip(make_rect(1, p0[0], p1[0], p0[2], p1[2], p0[1], m));
  ObjectPtr xn = flip(make_rect(0, p0[1], p1[1], p0[2], p1[2], p0[0], m));
  return push_obj(b, both(both(zp, both(yp, xp)), both(zn, both(yn, xn))));
}
rto_id rto_object_linear_move(rto_builder* b, rto_id o, const float motion[3]) {
  auto t = std::make_shared<LinearMove>();
  t->motion = v3(motion);
  if (!(t->object = obj(b, o))) return fail(-1, "bad object"), RTO_INVALID_ID;
  return push_obj(b, t);
}
rto_id rto_object_constant_medium(rto_builder* b, rto_id boundary, float density, rto_id material) {
  auto t = std::make_shared<ConstantMedium>();
  t->density = density;
  if (!(t->boundary = obj(b, boundary))) return fail(-1, "bad object"), RTO_INVALID_ID;
  if (!(t->material = mat(b, material))) return fail(-1, "bad material"), RTO_INVALID_ID;
  return push_obj(b, t);
}
rto_id rto_object_bvh(rto_builder* b, const rto_id* objects, size_t n, float e0, float e1) {
  std::vector<ObjectPtr> objs;
  for (size_t i = 0; i < n; i++) {
    auto o = obj(b, objects[i]);
    if (!o) return fail(-1, "bad object"), RTO_INVALID_ID;
    objs.push_back(o);
  }
  try {
    std::shared_ptr<Bvh> bvh = Bvh::build(std::move(objs), Range{e0, e1}, &b->ties);
    return push_obj(b, bvh);
  } catch (const std::exception& e) {
    fail(n == 0 ? -2 : -3, e.what());
    return RTO_INVALID_ID;
  }
}

// ORACLE-ONLY (no product counterpart): the tie audit of Bvh::new's sort (rto_scene.hpp TieAudit).  `seed` != 0: every later
// rto_object_bvh of this builder orders runs of equal sort keys at random (what bvh.rs:51's sort_unstable_by may do); 0 = the
// documented stable rule.  out[4] (may be NULL) = {sorts, sorts with tied keys, tied keys, sorts whose tie straddles the
// median split} accumulated over the builder's rto_object_bvh calls so far.
int rto_builder_bvh_ties(rto_builder* b, uint64_t seed, uint64_t* out) {
  if (!b) return fail(-1, "null argument");
  if (out) out[0] = b->ties.sorts, out[1] = b->ties.sorts_with_ties, out[2] = b->ties.tied_keys, out[3] = b->ties.straddling;
  b->ties.seed = seed;
  return 0;
}

rto_id rto_object_bvh_sah(rto_builder* b, const rto_id* objects, size_t n, float e0, float e1) {
  std::vector<ObjectPtr> objs;
  for (size_t i = 0; i < n; i++) {
    auto o = obj(b, objects[i]);
    if (!o) return fail(-1, "bad object"), RTO_INVALID_ID;
    objs.push_back(o);
  }
  try {
    std::shared_ptr<Bvh> bvh = Bvh::build_sah(std::move(objs), Range{e0, e1});
    return push_obj(b, bvh);
  } catch (const std::exception& e) {
    fail(n == 0 ? -2 : -3, e.what());
    return RTO_INVALID_ID;
  }
}

int rto_camera_look(const float from[3], const float at[3], const float up[3], float fov,
                    float aspect, float aperture, float focus_dist, float e0, float e1,
                    rto_camera* out) {
  Camera c = Camera::look(v3(from), v3(at), v3(up), fov, aspect, aperture, focus_dist, Range{e0, e1});
  auto put = [](float* d, Vec3 v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; };
  put(out->origin, c.origin);
  put(out->lower_left_corner, c.lower_left_corner);
  put(out->horizontal, c.horizontal);
  put(out->vertical, c.vertical);
  put(out->u, c.u);
  put(out->v, c.v);
  out->lens_radius = c.lens_radius;
  out->exposure_start = e0;
  out->exposure_end = e1;
  return 0;
}

static Camera to_camera(const rto_camera* c) {
  Camera k;
  k.origin = v3(c->origin);
  k.lower_left_corner = v3(c->lower_left_corner);
  k.horizontal = v3(c->horizontal);
  k.vertical = v3(c->vertical);
  k.u = v3(c->u);
  k.v = v3(c->v);
  k.lens_radius = c->lens_radius;
  k.exposure = Range{c->exposure_start, c->exposure_end};
  return k;
}

int rto_scene_create(rto_builder* b, const rto_id* world, size_t n, int /*device*/, rto_scene** out) {
  auto s = new rto_scene();
  for (size_t i = 0; i < n; i++) {
    auto o = obj(b, world[i]);
    if (!o) {
      delete s;
      return fail(-1, "bad world object");
    }
    s->world.list.push_back(o);
  }
  *out = s;
  return 0;
}
void rto_scene_destroy(rto_scene* s) { delete s; }

// One sample of the par_cast closure, lib.rs:366-372 (thread_rng() replaced by the per-sample
// counter stream -- the determinism contract).
static Vec3 one_sample(const World& world, const Camera& cam, const rto_params& p, uint32_t x,
                       uint32_t y, uint32_t s, Counters* counters, int* bounces) {
  SampleRng rng(p.seed, y * p.nx + x, s);
  rng.counters = counters;
  float u = ((float)x + rng.gen_f32()) / (float)p.nx;
  float v = ((float)y + rng.gen_f32()) / (float)p.ny;
  Ray r = cam.get_ray(u, v, rng);
  return color(world, r, rng, counters, (int)p.max_bounces, bounces, p.t_near);
}

static Vec3 one_pixel(const World& world, const Camera& cam, const rto_params& p, uint32_t x,
                      uint32_t y, Counters* counters) {
  Vec3 col;  // iter::Sum for Vec3 folds from Vec3::default() (vec3.rs:195-203)
  for (uint32_t s = 0; s < p.ns; s++) col = col + one_sample(world, cam, p, x, y, s, counters, nullptr);
  return col / (float)p.ns;  // lib.rs:374
}

static bool owns(const rto_params& p, uint32_t x, uint32_t row) {
  uint32_t tw = p.tile_w ? p.tile_w : 16, th = p.tile_h ? p.tile_h : 16;
  uint32_t nr = p.nranks ? p.nranks : 1;
  uint32_t tiles_x = (p.nx + tw - 1) / tw;
  uint32_t tile = (row / th) * tiles_x + (x / tw);
  return tile % nr == p.rank;
}

// par_cast, lib.rs:363-376 + Image::par_compute :324-332: rows in parallel, x sequential;
// output row 0 = y = ny-1.  `threads` <= 0 -> hardware_concurrency.
int rto_par_cast(rto_scene* s, const rto_camera* camera, const rto_params* params, float* out_rgb,
                 rto_stats* stats, int threads) {
  if (!s || !camera || !params || !out_rgb) return fail(-1, "null argument");
  if (!(camera->exposure_start < camera->exposure_end))
    return fail(-4, "Uniform::sample_single called with low >= high");  // camera.rs:55
  if (!std::isfinite(camera->exposure_start) || !std::isfinite(camera->exposure_end))  // rand 0.6.5: "non-finite boundaries"
    return fail(-4, "Uniform::sample_single called with non-finite boundaries");
  if (!std::isfinite(camera->exposure_end - camera->exposure_start))  // rand would shrink the scale; not restated (endless retry otherwise)
    return fail(-4, "exposure range wider than f32::MAX is not supported");
  const rto_params p = *params;
  Camera cam = to_camera(camera);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  std::atomic<uint32_t> next_row{0};
  std::vector<Counters> per_thread(nt);
  std::vector<uint64_t> px_done(nt, 0);
  bool count = stats && (p.flags & 1u);
  auto work = [&](int tid) {
    for (;;) {
      uint32_t row = next_row.fetch_add(1);
      if (row >= p.ny) break;
      uint32_t y = p.ny - 1 - row;
      for (uint32_t x = 0; x < p.nx; x++) {
        if (!owns(p, x, row)) continue;
        Vec3 c = one_pixel(s->world, cam, p, x, y, count ? &per_thread[tid] : nullptr);
        float* o = out_rgb + 3 * ((size_t)row * p.nx + x);
        o[0] = c.x, o[1] = c.y, o[2] = c.z;
        px_done[tid]++;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  if (stats) {
    Counters total;
    uint64_t px = 0;
    for (int t = 0; t < nt; t++) total.add(per_thread[t]), px += px_done[t];
    stats->kernel_ms = 0.f;
    stats->samples = px * p.ns;
    stats->aabb_tests = total.aabb_tests;
    stats->prim_tests = total.prim_tests;
    stats->shaded_hits = total.shaded_hits;
    stats->rays = total.rays;
    stats->draws = total.draws;
  }
  return 0;
}

// Mirror of rtg_par_cast_multi (same signature): every "device" renders its tile shard into a zero-filled frame and
// the frames are summed -- the checker for the product's shard-and-reduce entry point.
int rto_par_cast_multi(rto_scene* const* scenes, int n_scenes, const rto_camera* camera, const rto_params* params,
                       float* out_rgb, rto_stats* stats) {
  if (!scenes || n_scenes <= 0 || !camera || !params || !out_rgb) return fail(-1, "null argument");
  if (params->nranks > 1u) return fail(-1, "rto_par_cast_multi shards by itself");
  const size_t n = (size_t)params->nx * params->ny * 3;
  std::vector<float> part(n);
  for (size_t i = 0; i < n; i++) out_rgb[i] = 0.f;
  rto_stats total{};
  for (int i = 0; i < n_scenes; i++) {
    if (!scenes[i]) return fail(-1, "null scene handle");
    rto_params p = *params;
    p.rank = (uint32_t)i, p.nranks = (uint32_t)n_scenes;
    std::fill(part.begin(), part.end(), 0.f);
    rto_stats st{};
    int rc = rto_par_cast(scenes[i], camera, &p, part.data(), stats ? &st : nullptr, 0);
    if (rc) return rc;
    for (size_t k = 0; k < n; k++) out_rgb[k] = out_rgb[k] + part[k];
    total.samples += st.samples, total.aabb_tests += st.aabb_tests, total.prim_tests += st.prim_tests;
    total.shaded_hits += st.shaded_hits, total.rays += st.rays, total.draws += st.draws;
  }
  if (stats) {
    total.struct_size = stats->struct_size;
    *stats = total;
  }
  return 0;
}

// cast, lib.rs:378-397 + Image::compute :334-341: strictly sequential, ONE SmallRng stream threaded
// through every sample of every pixel, rows visited top (y = ny-1) to bottom.  This is the faithful
// restatement of benches/scene.rs:32-36 (seed 0xDEADBEEF); unverifiable against rustc here.
int rto_cast(rto_scene* s, const rto_camera* camera, uint32_t nx, uint32_t ny, uint32_t ns,
             uint32_t max_bounces, uint64_t small_rng_seed, float* out_rgb) {
  if (!s || !camera || !out_rgb) return fail(-1, "null argument");
  Camera cam = to_camera(camera);
  SmallRng rng(small_rng_seed);
  for (uint32_t row = 0; row < ny; row++) {
    uint32_t y = ny - 1 - row;
    for (uint32_t x = 0; x < nx; x++) {
      Vec3 col;
      for (uint32_t i = 0; i < ns; i++) {
        float u = ((float)x + rng.gen_f32()) / (float)nx;
        float v = ((float)y + rng.gen_f32()) / (float)ny;
        Ray r = cam.get_ray(u, v, rng);
        col = col + color(s->world, r, rng, nullptr, (int)max_bounces, nullptr);
      }
      col = col / (float)ns;
      float* o = out_rgb + 3 * ((size_t)row * nx + x);
      o[0] = col.x, o[1] = col.y, o[2] = col.z;
    }
  }
  return 0;
}

// print_ppm's per-channel quantisation, lib.rs:348-356
int rto_tonemap(int /*device*/, size_t n, const float* rgb, uint8_t* out) {
  for (size_t i = 0; i < n; i++) {
    float c = std::sqrt(rgb[i]);            // lib.rs:348
    int32_t q = f32_as_i32(255.99f * c);    // lib.rs:351 `(255.99 * x) as i32`
    out[i] = (uint8_t)std::min(255, std::max(0, q));  // .max(0).min(255)
  }
  return 0;
}

// ---- probes ---------------------------------------------------------------------------------
int rto_debug_hit_top(rto_scene* s, size_t n, const float* rays, uint64_t seed, float t_near,
                      float* out, uint32_t* out_material) {
  for (size_t i = 0; i < n; i++) {
    Ray r;
    r.origin = v3(rays + 7 * i);
    r.direction = v3(rays + 7 * i + 3);
    r.time = rays[7 * i + 6];
    SampleRng rng(seed, (uint32_t)i, 0);
    rng.set_event(1);
    HitRecord h;
    bool hit = s->world.hit_top(r, rng, nullptr, &h, t_near);
    float* o = out + 8 * i;
    o[0] = hit ? 1.f : 0.f;
    o[1] = hit ? h.t : 0.f;
    o[2] = hit ? h.p.x : 0.f, o[3] = hit ? h.p.y : 0.f, o[4] = hit ? h.p.z : 0.f;
    o[5] = hit ? h.normal.x : 0.f, o[6] = hit ? h.normal.y : 0.f, o[7] = hit ? h.normal.z : 0.f;
    if (out_material) out_material[i] = hit ? h.material->id : RTO_INVALID_ID;
  }
  return 0;
}

int rto_debug_samples(rto_scene* s, const rto_camera* camera, const rto_params* params, size_t n,
                      const uint32_t* xs, const uint32_t* ys, const uint32_t* samples, float* out_rgb,
                      uint32_t* out_info) {
  Camera cam = to_camera(camera);
  for (size_t i = 0; i < n; i++) {
    Counters c;
    int bounces = 0;
    Vec3 col = one_sample(s->world, cam, *params, xs[i], ys[i], samples[i], &c, &bounces);
    out_rgb[3 * i] = col.x, out_rgb[3 * i + 1] = col.y, out_rgb[3 * i + 2] = col.z;
    if (out_info) {
      out_info[4 * i] = (uint32_t)bounces;
      out_info[4 * i + 1] = (uint32_t)c.draws;
      out_info[4 * i + 2] = (uint32_t)c.aabb_tests;
      out_info[4 * i + 3] = (uint32_t)c.prim_tests;
    }
  }
  return 0;
}

// Traversal-order model (rto_scene.hpp OrderModel): renders the frame like rto_par_cast with the model switched on and returns one
// row of 9 counters per Bvh root the rays met -- {leaves, calls, n_ref, p_ref, n_near, p_near, differ, below_entry, hits} -- in
// `out` (room for `cap` rows); returns the number of rows, or a negative error.  The frame itself is discarded.
int rto_debug_order_model(rto_scene* s, const rto_camera* camera, const rto_params* params, int threads, uint64_t* out, int cap) {
  if (!s || !camera || !params || !out || cap <= 0) return fail(-1, "null argument");
  const rto_params p = *params;
  Camera cam = to_camera(camera);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  std::atomic<uint32_t> next_row{0};
  std::vector<OrderModel> per_thread(nt);
  auto work = [&](int tid) {
    tl_order_model = &per_thread[tid];
    for (;;) {
      uint32_t row = next_row.fetch_add(1);
      if (row >= p.ny) break;
      uint32_t y = p.ny - 1 - row;
      for (uint32_t x = 0; x < p.nx; x++)
        if (owns(p, x, row)) (void)one_pixel(s->world, cam, p, x, y, nullptr);
    }
    tl_order_model = nullptr;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  OrderModel total;
  for (auto& m : per_thread)
    for (auto& r : m.rows)
      if (r.root)
        if (OrderModel::Row* d = total.row(r.root, r.leaves)) {
          d->calls += r.calls, d->n_ref += r.n_ref, d->p_ref += r.p_ref, d->n_near += r.n_near, d->p_near += r.p_near;
          d->differ += r.differ, d->below_entry += r.below_entry, d->hits += r.hits;
        }
  int n = 0;
  for (auto& r : total.rows)
    if (r.root && n < cap) {
      uint64_t* o = out + 9 * n++;
      o[0] = r.leaves, o[1] = r.calls, o[2] = r.n_ref, o[3] = r.p_ref, o[4] = r.n_near, o[5] = r.p_near, o[6] = r.differ, o[7] = r.below_entry, o[8] = r.hits;
    }
  return n;
}

int rto_debug_math(int /*device*/, int op, size_t n, const float* in, const float* in2, float* out) {
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    switch (op) {
      case 0: out[i] = rt_logf(x); break;
      case 1: out[i] = rt_pow5f(x); break;
      case 2: out[i] = rt_sinf(x); break;
      case 3: out[i] = std::sqrt(x); break;
      case 4: out[i] = 1.f / x; break;
      case 5: out[i] = x / in2[i]; break;
      default: return fail(-1, "bad op");
    }
  }
  return 0;
}

// Host glibc versions of the same functions, for the ULP-distance report (the reference's f32::ln /
// powf / sin lower to these on a glibc host).
int rto_debug_glibc(int op, size_t n, const float* in, float* out) {
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    switch (op) {
      case 0: out[i] = logf(x); break;
      case 1: out[i] = powf(x, 5.f); break;
      case 2: out[i] = sinf(x); break;
      default: return fail(-1, "bad op");
    }
  }
  return 0;
}

// Distance scan: compares the restatements (rto_libm.hpp) with the PLATFORM libm over [lo_bits, hi_bits] (float bit
// patterns, step `stride`) on all hardware threads; returns #mismatches and the max ulp distance (NaN == NaN).
int rto_debug_ulp_scan(int op, uint32_t lo_bits, uint32_t hi_bits, uint32_t stride,
                       uint64_t* n_tested, uint64_t* n_mismatch, uint32_t* max_ulp) {
  int nt = (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  std::vector<uint64_t> tested(nt, 0), mism(nt, 0);
  std::vector<uint32_t> worst(nt, 0);
  const uint64_t total = ((uint64_t)hi_bits - lo_bits) / stride + 1;
  auto work = [&](int t) {
    const uint64_t i0 = total * t / nt, i1 = total * (t + 1) / nt;
    uint64_t my_tested = 0, my_mism = 0;  // (locals: the per-thread slots share cache lines)
    uint32_t my_worst = 0;
    for (uint64_t i = i0; i < i1; i++) {
      float x = f32_from_bits((uint32_t)(lo_bits + i * stride));
      float a, g;
      if (op == 0) a = rt_logf(x), g = logf(x);
      else if (op == 1) a = rt_pow5f(x), g = powf(x, 5.f);
      else a = rt_sinf(x), g = sinf(x);
      my_tested++;
      uint32_t ua = f32_bits(a), ug = f32_bits(g);
      if (ua != ug) {
        if ((a != a) && (g != g)) continue;
        my_mism++;
        int64_t ia = (ua & 0x80000000u) ? -(int64_t)(ua & 0x7fffffffu) : (int64_t)ua;
        int64_t ig = (ug & 0x80000000u) ? -(int64_t)(ug & 0x7fffffffu) : (int64_t)ug;
        uint64_t d = (uint64_t)(ia > ig ? ia - ig : ig - ia);
        if (d > my_worst) my_worst = (uint32_t)(d > 0xffffffffu ? 0xffffffffu : d);
      }
    }
    tested[t] = my_tested, mism[t] = my_mism, worst[t] = my_worst;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  *n_tested = 0, *n_mismatch = 0, *max_ulp = 0;
  for (int t = 0; t < nt; t++) *n_tested += tested[t], *n_mismatch += mism[t], *max_ulp = std::max(*max_ulp, worst[t]);
  return 0;
}

// RNG known-answer probes
void rto_debug_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                      uint32_t out[4]) {
  Philox4x32::block(k0, k1, c0, c1, c2, c3, out);
}
void rto_debug_small_rng_u64(uint64_t seed, size_t n, uint64_t* out) {
  SmallRng r(seed);
  for (size_t i = 0; i < n; i++) out[i] = r.next_u64();
}
void rto_debug_mcg128_u64(uint64_t state_hi, uint64_t state_lo, size_t n, uint64_t* out) {
  SmallRng r(0);
  r.state = (((unsigned __int128)state_hi << 64) | state_lo) | 1;
  for (size_t i = 0; i < n; i++) out[i] = r.next_u64();
}
void rto_debug_small_rng_f32(uint64_t seed, size_t n, float* out) {
  SmallRng r(seed);
  for (size_t i = 0; i < n; i++) out[i] = r.gen_f32();
}
void rto_debug_sample_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t event, size_t n, uint32_t* out) {
  SampleRng r(seed, pixel, sample);
  r.set_event(event);
  for (size_t i = 0; i < n; i++) out[i] = r.next_u32();
}
int rto_debug_aabb_hit(const float mn[3], const float mx[3], const float o[3], const float d[3],
                       float t0, float t1) {
  Aabb a{v3(mn), v3(mx)};
  Ray r;
  r.origin = v3(o);
  r.direction = v3(d);
  return a.hit(r, Range{t0, t1}, nullptr) ? 1 : 0;
}
int rto_debug_bounding_box(rto_builder* b, rto_id o, float e0, float e1, float out[6]) {
  auto p = obj(b, o);
  if (!p) return fail(-1, "bad object");
  Aabb a = p->bounding_box(Range{e0, e1});
  out[0] = a.min.x, out[1] = a.min.y, out[2] = a.min.z, out[3] = a.max.x, out[4] = a.max.y, out[5] = a.max.z;
  return 0;
}
float rto_debug_perlin_turb(rto_builder* b, const float p[3], int depth) {
  return perlin_turb(*b->perlin, v3(p), depth);
}

}  // extern "C"
