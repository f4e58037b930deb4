// ORACLE -- TEST INFRASTRUCTURE ONLY (see rto_core.hpp header; parity unpinned).
//
// rto_scene.hpp: the Object / Material / Texture / Bvh / Camera surface of the reference, restated
// as a recursive pointer tree exactly like the Rust `Box<dyn Object>` graph.  Every `hit` follows
// the reference's predicates literally (never an inverted comparison -- SURVEY.md H5).
#pragma once
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <vector>

#include "rto_core.hpp"

namespace rto {

// ------------------------------------------------------------------------------------------------
// perlin.rs -- tables are process-global `lazy_static`s seeded from thread_rng() in the reference
// (perlin.rs:24-29), i.e. any fixed tables are admissible; here they are per-scene data.
// ------------------------------------------------------------------------------------------------
struct PerlinTables {
  Vec3 vecs[256];
  uint8_t perm_x[256], perm_y[256], perm_z[256];
};

// perlin.rs:31-47
inline float trilinear_interp(const Vec3 corners[2][2][2], Vec3 uvw) {
  float accum = 0.f;
  Vec3 uvw3 = uvw * uvw * (Vec3::from(3.f) - 2.f * uvw);
  Vec3 uvw3_inv = Vec3::from(1.f) - uvw3;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) {
        Vec3 ijk((float)i, (float)j, (float)k);
        float weight = dot(corners[i][j][k], uvw - ijk);
        Vec3 ijk_inv = Vec3::from(1.f) - ijk;
        Vec3 m = ijk * uvw3 + ijk_inv * uvw3_inv;
        accum = accum + ((m.x * m.y) * m.z) * weight;
      }
  return accum;
}

// Rust `f32 as i32`: saturating, NaN -> 0
inline int32_t f32_as_i32(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int32_t)f;
}

// perlin.rs:49-64.  `(i + di) & 255` on i32: `as i32` saturates for |p| >= 2^31 and the release build's `+` wraps --
// done in u32 here (same low 8 bits, no signed-overflow UB)
inline float perlin_noise(const PerlinTables& T, Vec3 p) {
  Vec3 ijk(std::floor(p.x), std::floor(p.y), std::floor(p.z));
  Vec3 uvw = p - ijk;
  Vec3 corners[2][2][2];
  for (int di = 0; di < 2; di++)
    for (int dj = 0; dj < 2; dj++)
      for (int dk = 0; dk < 2; dk++) {
        uint8_t ix = T.perm_x[((uint32_t)f32_as_i32(ijk.x) + (uint32_t)di) & 255u];
        uint8_t iy = T.perm_y[((uint32_t)f32_as_i32(ijk.y) + (uint32_t)dj) & 255u];
        uint8_t iz = T.perm_z[((uint32_t)f32_as_i32(ijk.z) + (uint32_t)dk) & 255u];
        corners[di][dj][dk] = T.vecs[ix ^ iy ^ iz];
      }
  return trilinear_interp(corners, uvw);
}

// perlin.rs:66-75
inline float perlin_turb(const PerlinTables& T, Vec3 p, int depth) {
  float accum = 0.f, weight = 1.f;
  for (int i = 0; i < depth; i++) {
    accum += weight * perlin_noise(T, p);
    weight *= 0.5f;
    p = 2.f * p;
  }
  return std::fabs(accum);
}

// ------------------------------------------------------------------------------------------------
// texture.rs:6-26 -- `Arc<dyn Fn(Vec3)->Vec3>` closures; restated as a closed tree.
// ------------------------------------------------------------------------------------------------
struct Texture {
  enum Kind { CONSTANT, CHECKER, PERLIN } kind = CONSTANT;
  Vec3 color;                              // constant (texture.rs:8-10)
  std::shared_ptr<Texture> t0, t1;         // checker  (texture.rs:12-21)
  float scale = 0.f;                       // perlin   (texture.rs:23-26)
  std::shared_ptr<PerlinTables> tables;

  Vec3 eval(Vec3 p) const {
    switch (kind) {
      case CONSTANT: return color;
      case CHECKER: {
        Vec3 q = 10.f * p;
        float s = (rt_sinf(q.x) * rt_sinf(q.y)) * rt_sinf(q.z);
        return s < 0.f ? t1->eval(p) : t0->eval(p);
      }
      default: return Vec3::from(perlin_turb(*tables, scale * p, 7));
    }
  }
};
using TexturePtr = std::shared_ptr<Texture>;

// ------------------------------------------------------------------------------------------------
// material.rs
// ------------------------------------------------------------------------------------------------
struct HitRecord;

struct Material {
  enum Kind { LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, ISOTROPIC } kind = LAMBERTIAN;
  TexturePtr tex;       // Lambertian/Isotropic albedo, DiffuseLight emission
  Vec3 albedo;          // Metal
  float fuzz = 0.f;     // Metal
  float ref_idx = 0.f;  // Dielectric
  float brightness = 0.f;
  uint32_t id = 0;      // builder handle, reported by probes

  bool scatter(const Ray& ray, const HitRecord& hit, Rng& rng, Ray* out, Vec3* attenuation) const;
  Vec3 emitted(Vec3 p) const {  // material.rs:120-128
    if (kind == DIFFUSE_LIGHT) return brightness * tex->eval(p);
    return Vec3();
  }
};
using MaterialPtr = std::shared_ptr<Material>;

// object.rs:61-71
struct HitRecord {
  float t = 0.f;
  Vec3 p, normal;
  const Material* material = nullptr;
};

// material.rs:142-146
inline float schlick(float cos, float ref_idx) {
  float r0 = (1.f - ref_idx) / (1.f + ref_idx);
  r0 = r0 * r0;
  return r0 + (1.f - r0) * rt_pow5f(1.f - cos);
}

// material.rs:55-118
inline bool Material::scatter(const Ray& ray, const HitRecord& hit, Rng& rng, Ray* out,
                              Vec3* attenuation) const {
  switch (kind) {
    case LAMBERTIAN: {
      Vec3 target = hit.p + hit.normal + in_unit_sphere(rng);
      out->origin = hit.p;
      out->direction = target - hit.p;
      out->time = ray.time;
      *attenuation = tex->eval(hit.p);
      return true;
    }
    case METAL: {
      Vec3 refl = reflect(into_unit(ray.direction), hit.normal);
      Vec3 dir = refl + fuzz * in_unit_sphere(rng);
      out->origin = hit.p;
      out->direction = dir;
      out->time = ray.time;
      if (dot(dir, hit.normal) > 0.f) {
        *attenuation = albedo;
        return true;
      }
      return false;  // material.rs:73-79 (quirk H8: path ends, accum is returned)
    }
    case DIELECTRIC: {
      Vec3 outward_normal;
      float ni_over_nt, cosine;
      if (dot(ray.direction, hit.normal) > 0.f) {
        outward_normal = -hit.normal;
        ni_over_nt = ref_idx;
        cosine = ref_idx * dot(ray.direction, hit.normal) / length(ray.direction);
      } else {
        outward_normal = hit.normal;
        ni_over_nt = 1.0f / ref_idx;
        cosine = -dot(ray.direction, hit.normal) / length(ray.direction);
      }
      Vec3 direction;
      bool refracted = refract(ray.direction, outward_normal, ni_over_nt, &direction);
      // .filter(|_| rng.gen::<f32>() >= schlick(..)): the draw happens only when refract is Some
      if (refracted) refracted = rng.gen_f32() >= schlick(cosine, ref_idx);
      if (!refracted) direction = reflect(ray.direction, hit.normal);
      *attenuation = Vec3::from(1.f);
      out->origin = hit.p;
      out->direction = direction;
      out->time = ray.time;
      return true;
    }
    case DIFFUSE_LIGHT: return false;
    default: {  // ISOTROPIC, material.rs:108-115
      out->origin = hit.p;
      out->direction = in_unit_sphere(rng);
      out->time = ray.time;
      *attenuation = tex->eval(hit.p);
      return true;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// object.rs:15-40 `trait Object`
// ------------------------------------------------------------------------------------------------
struct HitCtx {
  Rng* rng;            // the `&mut dyn FnMut() -> f32` of object.rs:33 is rng.gen::<f32>() (lib.rs:41,53)
  Counters* counters;  // may be null
};

struct Object {
  virtual ~Object() = default;
  virtual bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const = 0;
  virtual Aabb bounding_box(Range exposure) const = 0;
  // (traversal-order model below, not in the reference) does a ConstantMedium sit in this subtree? -- its `hit` draws from the RNG
  virtual bool has_medium() const { return false; }
};
using ObjectPtr = std::shared_ptr<Object>;

// object.rs:74-119
struct Sphere final : Object {
  float radius;
  MaterialPtr material;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    if (ctx.counters) ctx.counters->prim_tests++;
    float a = dot(ray.direction, ray.direction);
    float b = dot(ray.origin, ray.direction);
    float c = dot(ray.origin, ray.origin) - radius * radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0.f) {
      float roots[2] = {(-b - std::sqrt(discriminant)) / a, (-b + std::sqrt(discriminant)) / a};
      for (float t : roots) {
        if (t < t_range.end && t >= t_range.start) {
          Vec3 p = ray.point_at_parameter(t);
          rec->t = t;
          rec->p = p;
          rec->normal = p / radius;
          rec->material = material.get();
          return true;
        }
      }
    }
    return false;
  }
  Aabb bounding_box(Range) const override {
    return Aabb{-Vec3::from(radius), Vec3::from(radius)};
  }
};

// object.rs:131-234; axis: 0=StaticX (others Y,Z), 1=StaticY (X,Z), 2=StaticZ (X,Y)  (:157-181)
struct Rect final : Object {
  int axis;
  Range range0, range1;
  float k;
  MaterialPtr material;
  int other1() const { return axis == 0 ? 1 : 0; }
  int other2() const { return axis == 2 ? 1 : 2; }
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    if (ctx.counters) ctx.counters->prim_tests++;
    float t = (k - ray.origin[axis]) / ray.direction[axis];
    if (t < t_range.start || t >= t_range.end) return false;
    float x = ray.origin[other1()] + t * ray.direction[other1()];
    float y = ray.origin[other2()] + t * ray.direction[other2()];
    if (x < range0.start || x >= range0.end || y < range1.start || y >= range1.end) return false;
    Vec3 p = ray.point_at_parameter(t);
    Vec3 normal;
    normal.at(axis) = 1.f;
    rec->t = t;
    rec->p = p;
    rec->material = material.get();
    rec->normal = normal;
    return true;
  }
  Aabb bounding_box(Range) const override {
    Vec3 mn, mx;
    mn.at(axis) = k - 0.0001f;
    mx.at(axis) = k + 0.0001f;
    mn.at(other1()) = range0.start;
    mx.at(other1()) = range0.end;
    mn.at(other2()) = range1.start;
    mx.at(other2()) = range1.end;
    return Aabb{mn, mx};
  }
};

// object.rs:239-258
struct FlipNormals final : Object {
  ObjectPtr object;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    if (!object->hit(ray, t_range, ctx, rec)) return false;
    rec->normal = -rec->normal;
    return true;
  }
  Aabb bounding_box(Range e) const override { return object->bounding_box(e); }
  bool has_medium() const override { return object->has_medium(); }
};

// object.rs:262-292
struct Translate final : Object {
  Vec3 offset;
  ObjectPtr object;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    Ray t_ray = ray;
    t_ray.origin = ray.origin - offset;
    if (!object->hit(t_ray, t_range, ctx, rec)) return false;
    rec->p = rec->p + offset;
    return true;
  }
  Aabb bounding_box(Range e) const override {
    Aabb b = object->bounding_box(e);
    return Aabb{b.min + offset, b.max + offset};
  }
  bool has_medium() const override { return object->has_medium(); }
};

// object.rs:296-328
struct Scale final : Object {
  Vec3 factor;
  ObjectPtr object;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    Ray t_ray = ray;
    t_ray.origin = ray.origin / factor;
    t_ray.direction = ray.direction / factor;
    if (!object->hit(t_ray, t_range, ctx, rec)) return false;
    rec->p = rec->p * factor;
    rec->normal = rec->normal / factor;
    return true;
  }
  Aabb bounding_box(Range e) const override {
    Aabb b = object->bounding_box(e);
    return Aabb{b.min * factor, b.max * factor};
  }
  bool has_medium() const override { return object->has_medium(); }
};

// object.rs:335-390 (+ rotate_y :477-484)
struct RotateY final : Object {
  ObjectPtr object;
  float sin_theta, cos_theta;
  static Vec3 rot(Vec3 p, float s, float c) {  // object.rs:349-355
    return Vec3(dot(p, Vec3(c, 0.f, s)), dot(p, Vec3(0.f, 1.f, 0.f)), dot(p, Vec3(-s, 0.f, c)));
  }
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    Ray rot_ray = ray;
    rot_ray.origin = rot(ray.origin, -sin_theta, cos_theta);
    rot_ray.direction = rot(ray.direction, -sin_theta, cos_theta);
    if (!object->hit(rot_ray, t_range, ctx, rec)) return false;
    rec->p = rot(rec->p, sin_theta, cos_theta);
    rec->normal = rot(rec->normal, sin_theta, cos_theta);
    return true;
  }
  Aabb bounding_box(Range e) const override {  // object.rs:372-389
    Aabb b = object->bounding_box(e);
    Vec3 mn = Vec3::from(F32_MAX), mx = Vec3::from(F32_MIN);
    for (int i = 0; i < 8; i++) {
      Vec3 r = rot(b.corner(i), sin_theta, cos_theta);
      mn = Vec3(rs_min(mn.x, r.x), rs_min(mn.y, r.y), rs_min(mn.z, r.z));
      mx = Vec3(rs_max(mx.x, r.x), rs_max(mx.y, r.y), rs_max(mx.z, r.z));
    }
    return Aabb{mn, mx};
  }
  bool has_medium() const override { return object->has_medium(); }
};

// object.rs:394-417
struct And final : Object {
  ObjectPtr a, b;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    HitRecord h0, h1;
    bool hit0 = a->hit(ray, t_range, ctx, &h0);
    if (hit0) t_range.end = h0.t;
    bool hit1 = b->hit(ray, t_range, ctx, &h1);
    if (hit1) {
      *rec = h1;
      return true;
    }
    if (hit0) {
      *rec = h0;
      return true;
    }
    return false;
  }
  Aabb bounding_box(Range e) const override { return a->bounding_box(e).merge(b->bounding_box(e)); }
  bool has_medium() const override { return a->has_medium() || b->has_medium(); }
};

// object.rs:489-528
struct LinearMove final : Object {
  ObjectPtr object;
  Vec3 motion;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    Ray m = ray;
    m.origin = ray.origin - ray.time * motion;
    return object->hit(m, t_range, ctx, rec);  // quirk H8: hit.p is NOT moved back
  }
  Aabb bounding_box(Range e) const override {
    Aabb bb = object->bounding_box(e);
    Aabb s{bb.min + e.start * motion, bb.max + e.start * motion};
    Aabb t{bb.min + e.end * motion, bb.max + e.end * motion};
    return s.merge(t);
  }
  bool has_medium() const override { return object->has_medium(); }
};

// object.rs:533-580
struct ConstantMedium final : Object {
  ObjectPtr boundary;
  float density;
  MaterialPtr material;
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    HitRecord hit1, hit2;
    if (boundary->hit(ray, Range{F32_MIN, F32_MAX}, ctx, &hit1)) {
      if (boundary->hit(ray, Range{hit1.t + 0.0001f, F32_MAX}, ctx, &hit2)) {
        hit1.t = rs_max(hit1.t, t_range.start);
        hit2.t = rs_min(hit2.t, t_range.end);
        if (hit1.t >= hit2.t) return false;
        float distance_inside = (hit2.t - hit1.t) * length(ray.direction);
        float hit_distance = -(1.f / density) * rt_logf(ctx.rng->gen_f32());
        if (hit_distance < distance_inside) {
          float t = hit1.t + hit_distance / length(ray.direction);
          rec->t = t;
          rec->p = ray.point_at_parameter(t);
          rec->normal = Vec3(1.f, 0.f, 0.f);
          rec->material = material.get();
          return true;
        }
      }
    }
    return false;
  }
  Aabb bounding_box(Range e) const override { return boundary->bounding_box(e); }
  bool has_medium() const override { return true; }
};

// ------------------------------------------------------------------------------------------------
// bvh.rs
// ------------------------------------------------------------------------------------------------
// Traversal-ORDER MODEL (not in the reference).  Counts what an order-dependent walk of the SAME tree would cost,
// beside the reference's walk, without touching it: at every outermost call of Bvh::hit on a tree without a ConstantMedium the
// model re-walks the tree NEAR CHILD FIRST -- the child on the side the ray comes from along the node's split axis (the axis
// Bvh::new sorted on: left = the lower half, bvh.rs:51-72) -- with a tie-safe rule: a box is entered when min(best, far) >= start
// and a hit replaces the best one when t < best, or t == best and its leaf comes earlier in the reference's depth-first order
// (which is what `t < t_range.end`, object.rs:99, gives the reference).  Per root: calls, the reference's Aabb::hit / primitive
// tests, the model's, results that differ from the reference's (t or leaf), and accepted hits with t below the entry distance of
// their own leaf box (the roundoff cases in which a box test and a primitive test disagree: only there can the orders differ).
struct OrderModel {
  struct Row {
    const void* root = nullptr;
    uint64_t leaves = 0, calls = 0, n_ref = 0, p_ref = 0, n_near = 0, p_near = 0, differ = 0, below_entry = 0, hits = 0;
  };
  Row rows[8];
  Row* row(const void* root, uint64_t leaves) {
    for (auto& r : rows) {
      if (r.root == root) return &r;
      if (r.root == nullptr) {
        r.root = root, r.leaves = leaves;
        return &r;
      }
    }
    return nullptr;
  }
};
inline thread_local OrderModel* tl_order_model = nullptr;  // set by rto_debug_order_model around a render
inline thread_local bool tl_order_model_busy = false;

struct Bvh final : Object {
  Aabb bbox;
  size_t size = 0;
  std::unique_ptr<Bvh> left, right;  // BvhContents::Node
  ObjectPtr leaf;                    // BvhContents::Leaf
  int axis = 0;                      // (order model) the axis this node's objects were sorted on
  bool medium = false;               // (order model) a ConstantMedium below
  bool has_medium() const override { return medium; }

  // Tie audit (tests/test_bvh_ties.py): what `sort_unstable_by` (bvh.rs:51) is free to do.  Counts the sorts whose keys
  // tie and those where a run of equal keys STRADDLES the median split (only there can the tie order change which leaf
  // goes left or right, i.e. the tree's shape); with seed != 0 every run of equal keys is shuffled after the stable sort
  // -- any outcome an unstable sort could produce, at every level independently.
  struct TieAudit {
    uint64_t seed = 0;  // 0 = the documented rule: stable (ties keep input order)
    uint64_t sorts = 0, sorts_with_ties = 0, tied_keys = 0, straddling = 0;
    uint64_t next() {  // splitmix64
      uint64_t z = (seed += 0x9e3779b97f4a7c15ull);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      return z ^ (z >> 31);
    }
  };

  // bvh.rs:22-81.  `sort_unstable_by` tie order is rustc-version specific; this restatement uses a
  // STABLE sort (ties keep input order), which the build documents as its tie rule.
  static std::unique_ptr<Bvh> build(std::vector<ObjectPtr> objs, Range exposure, TieAudit* audit = nullptr) {
    if (objs.empty()) throw std::runtime_error("Can't create a BVH from zero objects.");  // bvh.rs:60
    auto axis_range = [&](int axis) {  // bvh.rs:27-35
      float start = F32_MAX, end = F32_MIN;
      for (auto& o : objs) {
        Aabb bb = o->bounding_box(exposure);
        float mn = rs_min(bb.min[axis], bb.max[axis]);
        float mx = rs_max(bb.min[axis], bb.max[axis]);
        start = rs_min(start, mn);
        end = rs_max(end, mx);
      }
      return end - start;
    };
    float ranges[3] = {axis_range(0), axis_range(1), axis_range(2)};
    for (float r : ranges)
      if (r != r) throw std::runtime_error("NaN extent in Bvh::new (partial_cmp unwrap)");  // bvh.rs:45
    // bvh.rs:38-47: descending sort of 3 pairs (insertion sort => stable): first maximum wins
    int axis = 0;
    if (ranges[1] > ranges[axis]) axis = 1;
    if (ranges[2] > ranges[axis]) axis = 2;
    // bvh.rs:51-57: sort by centroid*2
    std::vector<std::pair<float, ObjectPtr>> keyed;
    keyed.reserve(objs.size());
    for (auto& o : objs) {
      Aabb bb = o->bounding_box(exposure);
      float key = bb.min[axis] + bb.max[axis];
      if (key != key) throw std::runtime_error("NaN centroid in Bvh::new (partial_cmp unwrap)");
      keyed.emplace_back(key, o);
    }
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const auto& a, const auto& b) { return a.first < b.first; });
    if (audit && keyed.size() > 1) {
      audit->sorts++;
      bool any = false;
      for (size_t i = 0; i < keyed.size();) {
        size_t j = i + 1;
        while (j < keyed.size() && keyed[j].first == keyed[i].first) j++;
        if (j - i > 1) {
          any = true;
          audit->tied_keys += j - i;
          if (audit->seed != 0)  // Fisher-Yates inside the run of equal keys
            for (size_t k = j - 1; k > i; k--) std::swap(keyed[k], keyed[i + audit->next() % (k - i + 1)]);
        }
        i = j;
      }
      const size_t h = keyed.size() / 2;
      if (any) audit->sorts_with_ties++;
      if (keyed[h - 1].first == keyed[h].first) audit->straddling++;
    }
    auto node = std::make_unique<Bvh>();
    if (keyed.size() == 1) {  // bvh.rs:61-65
      node->bbox = keyed[0].second->bounding_box(exposure);
      node->size = 1;
      node->leaf = keyed[0].second;
      node->medium = node->leaf->has_medium();
      return node;
    }
    node->axis = axis;
    size_t half = keyed.size() / 2;  // bvh.rs:68-72
    std::vector<ObjectPtr> l, r;
    for (size_t i = 0; i < keyed.size(); i++) (i < half ? l : r).push_back(keyed[i].second);
    node->right = build(std::move(r), exposure, audit);
    node->left = build(std::move(l), exposure, audit);
    node->bbox = node->left->bbox.merge(node->right->bbox);
    node->size = node->left->size + node->right->size;
    node->medium = node->left->medium || node->right->medium;
    return node;
  }

  // NOT IN THE REFERENCE ("next"): surface-area-heuristic builder over the same node
  // type.  Full sweep on each axis over centroid-sorted objects; cost = SA(left)*n_left + SA(right)*n_right;
  // ties keep the lower axis / earlier split; stable sorts.  One object per leaf like Bvh::new, so the
  // traversal code (hit) is unchanged; only the tree shape differs.
  static float half_area(const Aabb& b) {
    Vec3 e = b.max - b.min;
    return (e.x * e.y + e.y * e.z) + e.z * e.x;
  }
  static std::unique_ptr<Bvh> build_sah(std::vector<ObjectPtr> objs, Range exposure) {
    if (objs.empty()) throw std::runtime_error("Can't create a BVH from zero objects.");
    auto node = std::make_unique<Bvh>();
    if (objs.size() == 1) {
      node->bbox = objs[0]->bounding_box(exposure);
      node->size = 1;
      node->leaf = objs[0];
      node->medium = node->leaf->has_medium();
      return node;
    }
    const size_t n = objs.size();
    float best_cost = 0.f;
    int best_axis = -1;
    size_t best_split = 0;
    std::vector<ObjectPtr> best_order;
    for (int axis = 0; axis < 3; axis++) {
      std::vector<std::pair<float, ObjectPtr>> keyed;
      for (auto& o : objs) {
        Aabb bb = o->bounding_box(exposure);
        keyed.emplace_back(bb.min[axis] + bb.max[axis], o);
      }
      std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      std::vector<float> right_area(n);
      Aabb acc = keyed[n - 1].second->bounding_box(exposure);
      for (size_t i = n - 1; i >= 1; i--) {
        acc = (i == n - 1) ? acc : acc.merge(keyed[i].second->bounding_box(exposure));
        right_area[i] = half_area(acc);
      }
      Aabb left = keyed[0].second->bounding_box(exposure);
      for (size_t i = 1; i < n; i++) {  // split: [0, i) | [i, n)
        if (i > 1) left = left.merge(keyed[i - 1].second->bounding_box(exposure));
        float cost = half_area(left) * (float)i + right_area[i] * (float)(n - i);
        if (best_axis < 0 || cost < best_cost) {
          best_cost = cost, best_axis = axis, best_split = i;
          best_order.clear();
          for (auto& k : keyed) best_order.push_back(k.second);
        }
      }
    }
    std::vector<ObjectPtr> l(best_order.begin(), best_order.begin() + best_split), r(best_order.begin() + best_split, best_order.end());
    node->right = build_sah(std::move(r), exposure);
    node->left = build_sah(std::move(l), exposure);
    node->bbox = node->left->bbox.merge(node->right->bbox);
    node->size = node->left->size + node->right->size;
    node->medium = node->left->medium || node->right->medium;
    node->axis = best_axis;
    return node;
  }

  // ---- order model (see OrderModel) ----
  struct ModelBest {
    float t;
    uint64_t leaf;  // index of the winning leaf in the reference's depth-first order
    bool any;
  };
  void model_walk(const Ray& ray, float t_min, uint64_t first_leaf, ModelBest& best, OrderModel::Row& m) const {
    m.n_near++;
    // Aabb::hit's arithmetic (aabb.rs:16-27) with the tie-safe comparison
    Vec3 inv_d(1.f / ray.direction.x, 1.f / ray.direction.y, 1.f / ray.direction.z);
    Vec3 t0 = (bbox.min - ray.origin) * inv_d, t1 = (bbox.max - ray.origin) * inv_d;
    Vec3 a(inv_d.x < 0.f ? t1.x : t0.x, inv_d.y < 0.f ? t1.y : t0.y, inv_d.z < 0.f ? t1.z : t0.z);
    Vec3 b(inv_d.x < 0.f ? t0.x : t1.x, inv_d.y < 0.f ? t0.y : t1.y, inv_d.z < 0.f ? t0.z : t1.z);
    const float start = rs_max(t_min, rs_max(rs_max(a.x, a.y), a.z));
    const float end = rs_min(best.t, rs_min(rs_min(b.x, b.y), b.z));
    if (!(best.any ? end >= start : end > start)) return;  // (nothing found yet: the range's end is exclusive, as in the reference)
    if (leaf) {
      Counters c;
      HitCtx cx{nullptr, &c};
      HitRecord r;
      const float upto = (best.any && best.t < F32_MAX) ? std::nextafter(best.t, F32_MAX) : best.t;  // accepts t <= best once a hit is held
      const bool h = leaf->hit(ray, Range{t_min, upto}, cx, &r);
      m.p_near += c.prim_tests;
      if (h && (r.t < best.t || (best.any && r.t == best.t && first_leaf < best.leaf))) {
        if (r.t < start) m.below_entry++;
        best.t = r.t, best.leaf = first_leaf, best.any = true;
      }
      return;
    }
    const float da = axis == 0 ? ray.direction.x : (axis == 1 ? ray.direction.y : ray.direction.z);
    if (da < 0.f) {  // the ray comes from the upper side: right (upper half) first
      right->model_walk(ray, t_min, first_leaf + left->size, best, m);
      left->model_walk(ray, t_min, first_leaf, best, m);
    } else {
      left->model_walk(ray, t_min, first_leaf, best, m);
      right->model_walk(ray, t_min, first_leaf + left->size, best, m);
    }
  }
  // the reference's walk once more, only to learn WHICH leaf wins (same arithmetic as hit below)
  bool ref_leaf(const Ray& ray, Range t_range, uint64_t first_leaf, float& t, uint64_t& which) const {
    if (!bbox.hit(ray, t_range, nullptr)) return false;
    if (leaf) {
      HitCtx cx{nullptr, nullptr};
      HitRecord r;
      if (!leaf->hit(ray, t_range, cx, &r)) return false;
      t = r.t, which = first_leaf;
      return true;
    }
    float tl = 0.f, tr = 0.f;
    uint64_t wl = 0, wr = 0;
    const bool hl = left->ref_leaf(ray, t_range, first_leaf, tl, wl);
    if (hl) t_range.end = tl;
    const bool hr = right->ref_leaf(ray, t_range, first_leaf + left->size, tr, wr);
    if (hl && hr) {
      if (tl < tr) t = tl, which = wl;
      else t = tr, which = wr;
      return true;
    }
    if (hl) { t = tl, which = wl; return true; }
    if (hr) { t = tr, which = wr; return true; }
    return false;
  }

  // bvh.rs:84-120
  bool hit(const Ray& ray, Range t_range, HitCtx& ctx, HitRecord* rec) const override {
    if (tl_order_model && !tl_order_model_busy && !medium) {
      if (OrderModel::Row* m = tl_order_model->row(this, size)) {
        tl_order_model_busy = true;
        Counters c;
        HitCtx cx{ctx.rng, &c};
        const bool r = hit(ray, t_range, cx, rec);  // the reference's walk, counted
        if (ctx.counters) ctx.counters->add(c);
        m->calls++, m->n_ref += c.aabb_tests, m->p_ref += c.prim_tests, m->hits += r ? 1u : 0u;
        ModelBest best{t_range.end, 0, false};
        model_walk(ray, t_range.start, 0, best, *m);
        float t_ref = 0.f;
        uint64_t leaf_ref = 0;
        const bool r2 = ref_leaf(ray, t_range, 0, t_ref, leaf_ref);
        if (r2 != best.any || (r2 && (t_ref != best.t || leaf_ref != best.leaf))) m->differ++;
        tl_order_model_busy = false;
        return r;
      }
    }
    if (!bbox.hit(ray, t_range, ctx.counters)) return false;
    if (leaf) return leaf->hit(ray, t_range, ctx, rec);
    HitRecord hl, hr;
    bool hit_left = left->hit(ray, t_range, ctx, &hl);
    if (hit_left) t_range.end = hl.t;
    bool hit_right = right->hit(ray, t_range, ctx, &hr);
    if (hit_left && hit_right) {
      *rec = (hl.t < hr.t) ? hl : hr;
      return true;
    }
    if (hit_left) {
      *rec = hl;
      return true;
    }
    if (hit_right) {
      *rec = hr;
      return true;
    }
    return false;
  }
  Aabb bounding_box(Range) const override { return bbox; }
};

// ------------------------------------------------------------------------------------------------
// camera.rs
// ------------------------------------------------------------------------------------------------
struct Camera {
  Vec3 origin, lower_left_corner, horizontal, vertical, u, v;
  float lens_radius = 0.f;
  Range exposure{0.f, 1.f};

  // camera.rs:18-50.  tan lowers to the platform libm (host-side setup, not on the GPU path).
  static Camera look(Vec3 look_from, Vec3 look_at, Vec3 up, float fov, float aspect, float aperture,
                     float focus_dist, Range exposure) {
    Camera c;
    c.lens_radius = aperture / 2.f;
    float theta = fov * 3.14159265358979323846f / 180.f;
    float half_height = std::tan(theta / 2.f);
    float half_width = aspect * half_height;
    c.origin = look_from;
    Vec3 w = into_unit(look_from - look_at);
    c.u = into_unit(cross(up, w));
    c.v = cross(w, c.u);
    c.lower_left_corner = c.origin - half_width * focus_dist * c.u - half_height * focus_dist * c.v -
                          focus_dist * w;
    c.horizontal = 2.f * half_width * focus_dist * c.u;
    c.vertical = 2.f * half_height * focus_dist * c.v;
    c.exposure = exposure;
    return c;
  }

  // camera.rs:52-63
  Ray get_ray(float s, float t, Rng& rng) const {
    Vec3 rd = lens_radius * in_unit_disc(rng);
    Vec3 offset = rd.x * u + rd.y * v;
    float time = rng.gen_range_f32(exposure.start, exposure.end);
    Ray r;
    r.origin = origin + offset;
    r.direction = lower_left_corner + s * horizontal + t * vertical - origin - offset;
    r.time = time;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// lib.rs:23-101 -- World + color()
// ------------------------------------------------------------------------------------------------
struct World {
  std::vector<ObjectPtr> list;  // impl World for [Box<dyn Object>]  (lib.rs:33-49)
  // impl World for Bvh (lib.rs:51-55) is a one-element list holding a Bvh: identical arithmetic
  // (obj.hit(ray, 0.001..f32::MAX)).

  // NEAR is the literal 0.001 of lib.rs:35,53; a parameter here because the C ABI exposes it.
  bool hit_top(const Ray& ray, Rng& rng, Counters* counters, HitRecord* rec, float NEAR = 0.001f) const {
    if (counters) counters->rays++;
    float nearest = F32_MAX;
    bool any = false;
    HitCtx ctx{&rng, counters};
    for (auto& obj : list) {
      HitRecord r;
      if (obj->hit(ray, Range{NEAR, nearest}, ctx, &r)) {
        nearest = r.t;
        *rec = r;
        any = true;
      }
    }
    return any;
  }
};

// lib.rs:60-101.  `max_bounces` is the literal 50 of lib.rs:93.
inline Vec3 color(const World& world, Ray ray, Rng& rng, Counters* counters, int max_bounces,
                  int* bounces_out, float t_near = 0.001f) {
  Vec3 accum;
  Vec3 strength = Vec3::from(1.f);
  int bounces = 0;
  HitRecord hit;
  rng.set_event(1);  // determinism contract: event k = k-th hit_top + its scatter (rto_core.hpp)
  while (world.hit_top(ray, rng, counters, &hit, t_near)) {
    if (counters) counters->shaded_hits++;
    accum = accum + strength * hit.material->emitted(hit.p);
    Ray new_ray;
    Vec3 attenuation;
    if (hit.material->scatter(ray, hit, rng, &new_ray, &attenuation)) {
      ray = new_ray;
      strength = strength * attenuation;
    } else {
      if (bounces_out) *bounces_out = bounces;
      return accum;
    }
    if (bounces == max_bounces) {
      if (bounces_out) *bounces_out = bounces;
      return accum;
    }
    bounces += 1;
    rng.set_event((uint32_t)bounces + 1);
  }
  if (bounces_out) *bounces_out = bounces;
  return Vec3();
}

}  // namespace rto
