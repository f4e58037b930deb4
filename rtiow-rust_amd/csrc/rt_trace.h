// rt_trace.h -- the hot path as device code: hit_top (flat-program interpreter), materials,
// textures, camera and color().  Every predicate is the reference's, literally.
#pragma once
#include "flat_scene.h"
#include "rt_device.h"

namespace rtg {

struct DevScene {
  const uint4* lo;           // instruction packets 0
  const uint4* hi;           // instruction packets 1
  const uint4* mat;          // 2 packets per material
  const uint4* tex;          // 2 packets per texture
  const float4* perlin_vecs; // 256 gradients (perlin.rs VECS)
  const uint8_t* perlin_perm;// PERM_X | PERM_Y | PERM_Z, 3 x 256
  const uint32_t* lds_off;   // lean programs: byte offset of every record inside the LDS image (rt_pool.h); launches of the
                             // 4-wide variant (rt_pool.h WIDE) pass the words of the 4-wide image here
  uint32_t n_prog;
  uint32_t n_mat;
  uint32_t lds_image_bytes;  // size of that image, 0 = the program has none (WIDE launches: size of the 4-wide image)
};

struct DevCamera {  // camera.rs:6-15
  V3 origin, llc, horizontal, vertical, u, v;
  float lens_radius, e0, e1;
};

// n / d for a divisor that is fixed for the launch: libdivide's branch-free form, q = mulhi(n, m); (((n - q) >> 1) + q) >> s --
// exact for every 32-bit n (d >= 2; power of two: m = 0, s = log2 d - 1; d = 1 passes n through).  A 32-bit division costs ~25
// VALU instructions on gfx950; the work-item <-> pixel maps of the pool kernels do eight of them per sample.
struct FastDiv {
  uint32_t d, m, s;
};
inline FastDiv make_fastdiv(uint32_t d) {  // host
  FastDiv f{d, 0u, 0u};
  if (d < 2u) return f;
  uint32_t fl = 31u;
  while (!(d >> fl)) fl--;
  if ((d & (d - 1u)) == 0u) {
    f.s = fl - 1u;
    return f;
  }
  const uint64_t num = 1ull << (32u + fl);
  uint64_t pm = num / d;
  const uint64_t rem = num % d;
  pm *= 2u;
  if (rem * 2u >= d) pm += 1u;
  f.m = (uint32_t)(pm + 1u), f.s = fl;
  return f;
}

struct DevParams {
  uint32_t nx, ny, ns, max_bounces;
  float t_near;
  uint32_t seed_lo, seed_hi;
  uint32_t tile_w, tile_h, rank, nranks;
};
// derived from DevParams (make_pixmap): tiles per row and the divisors of the work-item <-> pixel maps (rt_pool.h).  Its own
// struct in the launch constants, read field by field where a map is evaluated: inside DevParams every pass that loads the
// frame parameters paid for 19 more SGPRs (C2 +0.5 %).
struct PixMap {
  uint32_t tiles_x;
  FastDiv d_tile_w, d_tile_h, d_tiles_x, d_nranks, d_px_per_tile, d_blocks_x;
};
inline PixMap make_pixmap(const DevParams& d) {  // host
  PixMap m;
  m.tiles_x = (d.nx + d.tile_w - 1u) / d.tile_w;
  m.d_tile_w = make_fastdiv(d.tile_w), m.d_tile_h = make_fastdiv(d.tile_h), m.d_tiles_x = make_fastdiv(m.tiles_x);
  m.d_nranks = make_fastdiv(d.nranks), m.d_px_per_tile = make_fastdiv(d.tile_w * d.tile_h), m.d_blocks_x = make_fastdiv(d.tile_w >> 3);
  return m;
}

struct HitRec {  // object.rs:61-71
  float t;
  V3 p, n;
  uint32_t mat;
};

struct Counts {
  uint32_t aabb, prim, shaded, rays;
};

RT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }

// object.rs:84-111 (sphere at the local origin)
RT_DEV bool sphere_hit_t(V3 o, V3 d, float radius, float t0, float t1, float& t_out) {
  float a = vdot(d, d);
  float b = vdot(o, d);
  float c = vdot(o, o) - radius * radius;
  float disc = b * b - a * c;
  if (disc > 0.f) {
    float sq = __builtin_sqrtf(disc);
    float t = (-b - sq) / a;
    if (t < t1 && t >= t0) {
      t_out = t;
      return true;
    }
    t = (-b + sq) / a;
    if (t < t1 && t >= t0) {
      t_out = t;
      return true;
    }
  }
  return false;
}

// object.rs:185-218; other axes: X->(Y,Z), Y->(X,Z), Z->(X,Y)  (object.rs:157-181)
RT_DEV bool rect_hit_t(V3 o, V3 d, uint32_t axis, float k, float r0s, float r0e, float r1s, float r1e,
                       float t0, float t1, float& t_out) {
  uint32_t o1 = axis == 0 ? 1u : 0u, o2 = axis == 2 ? 1u : 2u;
  float t = (k - vget(o, axis)) / vget(d, axis);
  if (t < t0 || t >= t1) return false;
  float x = vget(o, o1) + t * vget(d, o1);
  float y = vget(o, o2) + t * vget(d, o2);
  if (x < r0s || x >= r0e || y < r1s || y >= r1e) return false;
  t_out = t;
  return true;
}

// rect_prism(p0, p1, m), object.rs:420-473: And(And(+Z, And(+Y, +X)), And(Flip(-Z), And(Flip(-Y), Flip(-X)))).
// `And` tries its second object with the range already shrunk to the first one's hit and a later hit
// replaces an earlier one (object.rs:403-409), so the six Rect::hit run in that order against a shrinking
// `best`.  Returns the number of faces that hit (0 = None); `face` = axis | 4 when the normal is flipped.
RT_DEV uint32_t prism_hit_t(const uint4 lo, const uint4 hi, V3 o, V3 d, float t0, float t1, float& t_out, uint32_t& face) {
  const float p0x = u2f(lo.x), p1x = u2f(lo.y), p0y = u2f(lo.z), p1y = u2f(lo.w), p0z = u2f(hi.x), p1z = u2f(hi.y);
  uint32_t n = 0;
  float best = t1, t;
  if (rect_hit_t(o, d, 2u, p1z, p0x, p1x, p0y, p1y, t0, best, t)) best = t, face = 2u, n++;
  if (rect_hit_t(o, d, 1u, p1y, p0x, p1x, p0z, p1z, t0, best, t)) best = t, face = 1u, n++;
  if (rect_hit_t(o, d, 0u, p1x, p0y, p1y, p0z, p1z, t0, best, t)) best = t, face = 0u, n++;
  if (rect_hit_t(o, d, 2u, p0z, p0x, p1x, p0y, p1y, t0, best, t)) best = t, face = 2u | 4u, n++;
  if (rect_hit_t(o, d, 1u, p0y, p0x, p1x, p0z, p1z, t0, best, t)) best = t, face = 1u | 4u, n++;
  if (rect_hit_t(o, d, 0u, p0x, p0y, p1y, p0z, p1z, t0, best, t)) best = t, face = 0u | 4u, n++;
  t_out = best;
  return n;
}
RT_DEV V3 prism_normal(uint32_t face) {  // object.rs:212 + FlipNormals (object.rs:249-252)
  const uint32_t axis = face & 3u;
  V3 n = mk(axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f);
  return (face & 4u) ? vneg(n) : n;
}

// object.rs:349-355
RT_DEV V3 rot_y(V3 p, float s, float c) {
  return mk(vdot(p, mk(c, 0.f, s)), vdot(p, mk(0.f, 1.f, 0.f)), vdot(p, mk(-s, 0.f, c)));
}

// Boundary primitive of a ConstantMedium: only `t` is consumed (object.rs:551-554).
RT_DEV bool prim_hit_t(uint4 lo, uint4 hi, V3 o, V3 d, float t0, float t1, float& t) {
  uint32_t op = hi.w & 0xffu;
  if (op == OP_SPHERE) {
    V3 lo_o = o;
    if (hi.w & F_TRANSLATE) lo_o = vsub(o, mk(u2f(lo.x), u2f(lo.y), u2f(lo.z)));
    return sphere_hit_t(lo_o, d, u2f(lo.w), t0, t1, t);
  }
  return rect_hit_t(o, d, (hi.w >> F_AXIS_SHIFT) & 3u, u2f(lo.x), u2f(lo.y), u2f(lo.z), u2f(lo.w),
                    u2f(hi.x), t0, t1, t);
}

// ConstantMedium's two boundary queries (object.rs:551-552) against ONE primitive record:
//   h1 = boundary.hit(ray, f32::MIN..f32::MAX);  h2 = boundary.hit(ray, h1.t + 0.0001..f32::MAX)
// For a sphere both calls evaluate the same a, b, c, discriminant and roots (Sphere::hit is a pure function
// of the ray); only the accepted range differs, so the roots are computed once and the range predicates
// of object.rs:99 are applied twice.  Returns true when both queries hit.
RT_DEV bool boundary_pair_t(uint4 lo, uint4 hi, V3 o, V3 d, float& t1, float& t2, uint32_t& n_tests) {
  n_tests = 1;
  if ((hi.w & 0xffu) != OP_SPHERE) {
    if (!prim_hit_t(lo, hi, o, d, -F32_MAX, F32_MAX, t1)) return false;
    n_tests = 2;
    return prim_hit_t(lo, hi, o, d, t1 + 0.0001f, F32_MAX, t2);
  }
  V3 c = o;
  if (hi.w & F_TRANSLATE) c = vsub(o, mk(u2f(lo.x), u2f(lo.y), u2f(lo.z)));
  const float radius = u2f(lo.w);
  const float a = vdot(d, d), b = vdot(c, d), cc = vdot(c, c) - radius * radius;
  const float disc = b * b - a * cc;
  if (!(disc > 0.f)) return false;
  const float sq = __builtin_sqrtf(disc);
  const float r1 = (-b - sq) / a, r2 = (-b + sq) / a;
  if (r1 < F32_MAX && r1 >= -F32_MAX) t1 = r1;
  else if (r2 < F32_MAX && r2 >= -F32_MAX) t1 = r2;
  else return false;
  n_tests = 2;
  const float lo2 = t1 + 0.0001f;
  if (r1 < F32_MAX && r1 >= lo2) t2 = r1;
  else if (r2 < F32_MAX && r2 >= lo2) t2 = r2;
  else return false;
  return true;
}

// `boundary.hit(ray, t_lo..t_hi)` of ConstantMedium (object.rs:551-552) when the boundary is an object graph:
// a nested walk over the boundary's own records [first, end) that only keeps the closest t.  Same
// predicates and visiting order as hit_top; no hit record, no media.  Rare path -> out of line, program
// read from global memory, private ray stack.
// Everything travels by value (program pointers in, {hit, t, test counts} out): no caller object has its address taken.
struct BoundaryHit {
  float t;
  uint32_t any, n_aabb, n_prim;
};
__device__ __attribute__((noinline)) BoundaryHit boundary_hit_t(const uint4* __restrict__ prog_lo, const uint4* __restrict__ prog_hi,
                                                                uint32_t first, uint32_t end_pc, V3 o, V3 d, float time,
                                                                float t_lo, float t_hi) {
  uint32_t n_aabb = 0, n_prim = 0;
  V3 inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
  float best = t_hi;
  bool any = false;
  int depth = 0;
  V3 so[MAX_XFORM_DEPTH], sd[MAX_XFORM_DEPTH];
  uint32_t pc = first;
  while (pc < end_pc) {
    const uint4 hi = prog_hi[pc], lo = prog_lo[pc];
    const uint32_t op = hi.w & 0xffu;
    if (op == OP_BOX) {
      n_aabb++;
      float t0x = (u2f(lo.x) - o.x) * inv.x, t1x = (u2f(lo.y) - o.x) * inv.x;
      float t0y = (u2f(lo.z) - o.y) * inv.y, t1y = (u2f(lo.w) - o.y) * inv.y;
      float t0z = (u2f(hi.x) - o.z) * inv.z, t1z = (u2f(hi.y) - o.z) * inv.z;
      float ax = inv.x < 0.f ? t1x : t0x, bx = inv.x < 0.f ? t0x : t1x;
      float ay = inv.y < 0.f ? t1y : t0y, by = inv.y < 0.f ? t0y : t1y;
      float az = inv.z < 0.f ? t1z : t0z, bz = inv.z < 0.f ? t0z : t1z;
      float start = rs_max(t_lo, rs_max(rs_max(ax, ay), az));
      float fin = rs_min(best, rs_min(rs_min(bx, by), bz));
      pc = (fin > start) ? pc + 1 : hi.z;
    } else if (op == OP_SPHERE) {
      n_prim++;
      V3 lo_o = o;
      if (hi.w & F_TRANSLATE) lo_o = vsub(o, mk(u2f(lo.x), u2f(lo.y), u2f(lo.z)));
      if (hi.w & F_MOVE) {  // fused LinearMove (flat_scene.h): motion in the OP_EXT record behind
        const uint4 mv = prog_lo[++pc];
        lo_o = vsub(lo_o, smul(time, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
      }
      float t;
      if (sphere_hit_t(lo_o, d, u2f(lo.w), t_lo, best, t)) best = t, any = true;
      pc++;
    } else if (op == OP_RECT) {
      n_prim++;
      float t;
      if (rect_hit_t(o, d, (hi.w >> F_AXIS_SHIFT) & 3u, u2f(lo.x), u2f(lo.y), u2f(lo.z), u2f(lo.w), u2f(hi.x), t_lo, best, t))
        best = t, any = true;
      pc++;
    } else if (op == OP_PRISM) {
      n_prim += 6;
      float t;
      uint32_t face;
      if (prism_hit_t(lo, hi, o, d, t_lo, best, t, face)) best = t, any = true;
      pc++;
    } else if (op == OP_PUSH) {
      const uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      so[depth] = o, sd[depth] = d;
      depth++;
      const V3 a = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
      if (hi.w & F_PRE_TRANSLATE) o = vsub(o, mk(u2f(lo.w), u2f(hi.x), u2f(hi.y)));
      if (kind == XF_TRANSLATE) {
        o = vsub(o, a);
      } else if (kind == XF_ROTATE_Y) {
        o = rot_y(o, -a.x, a.y), d = rot_y(d, -a.x, a.y);
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_SCALE) {
        o = vdiv(o, a), d = vdiv(d, a);
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_MOVE) {
        o = vsub(o, smul(time, a));
      }
      pc++;
    } else if (op == OP_POP) {
      const uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      depth--;
      o = so[depth], d = sd[depth];
      if (kind == XF_ROTATE_Y || kind == XF_SCALE) inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      pc++;
    } else {
      pc++;
    }
  }
  return BoundaryHit{best, any ? 1u : 0u, n_aabb, n_prim};
}

// The GENERAL walk (programs with FEAT_DEEP: more than MAX_XFORM_DEPTH nested wrappers, a ConstantMedium inside another
// medium's boundary, a medium below an `And` below a Bvh).  One function serves the main walk (LEVEL 0, `rec` != null: keeps the
// hit record, lib.rs:33-55) and the boundary queries of ConstantMedium::hit (object.rs:551-552; LEVEL >= 1, `rec` == null: only
// the closest t matters) -- a medium met at LEVEL n runs its two queries at LEVEL n + 1 over its boundary's record stream, with
// the event's RNG stream handed down so that the draws happen in the reference's order (object.rs:562).  Same predicates,
// same visiting order, same counters as hit_top / boundary_hit_t; one lane per ray, private stacks: the slow, general path.
// XD = wrapper levels the ray stacks hold, ML = levels of boundary queries below the main walk: chosen per scene by what its graph
// needs (FEAT_DEEP_FEW_WRAPPERS / FEAT_DEEP_ONE_LEVEL) -- the stacks are private memory and registers of every lane.
template <int LEVEL, bool COUNT, int XD = MAX_DEEP_XFORM_DEPTH, int ML = MAX_MEDIUM_NESTING>
__device__ __attribute__((noinline)) bool walk_deep(const DevScene& sc, uint32_t first, uint32_t end_pc, V3 o, V3 d, const float time,
                                                    const float t_lo, const float t_hi, SampleRng& rng, HitRec* rec, float& t_out,
                                                    Counts& cnt) {
  V3 inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
  float best = t_hi;
  bool any = false;
  int depth = 0, tag = 0, sp = 0;
  uint32_t nhits = 0, root_hits = 0;
  V3 so[XD], sd[XD];
  // OP_SAVE .. OP_MERGE: the hit in front of an And-with-medium leaf of a Bvh
  float sv_best[MAX_SAVE_NESTING];
  HitRec sv_rec[MAX_SAVE_NESTING];
  int sv_tag[MAX_SAVE_NESTING];
  uint32_t sv_nhits[MAX_SAVE_NESTING], sv_root[MAX_SAVE_NESTING];
  bool sv_any[MAX_SAVE_NESTING];
  uint32_t pc = first;
  while (pc < end_pc) {
    const uint4 hi = sc.hi[pc];
    const uint32_t op = hi.w & 0xffu;
    if (op == OP_END) break;
    const uint4 lo = sc.lo[pc];
    if (op == OP_BOX) {  // Aabb::hit, aabb.rs:16-27
      if (COUNT) cnt.aabb++;
      if (hi.w & F_BVH_ROOT) root_hits = nhits;
      float t0x = (u2f(lo.x) - o.x) * inv.x, t1x = (u2f(lo.y) - o.x) * inv.x;
      float t0y = (u2f(lo.z) - o.y) * inv.y, t1y = (u2f(lo.w) - o.y) * inv.y;
      float t0z = (u2f(hi.x) - o.z) * inv.z, t1z = (u2f(hi.y) - o.z) * inv.z;
      float ax = inv.x < 0.f ? t1x : t0x, bx = inv.x < 0.f ? t0x : t1x;
      float ay = inv.y < 0.f ? t1y : t0y, by = inv.y < 0.f ? t0y : t1y;
      float az = inv.z < 0.f ? t1z : t0z, bz = inv.z < 0.f ? t0z : t1z;
      float start = rs_max(t_lo, rs_max(rs_max(ax, ay), az));
      float end = rs_min(best, rs_min(rs_min(bx, by), bz));
      pc = (end > start) ? pc + 1 : hi.z;
    } else if (op == OP_SPHERE) {
      if (COUNT) cnt.prim++;
      V3 off = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
      V3 lo_o = o;
      if (hi.w & F_TRANSLATE) lo_o = vsub(o, off);
      if (hi.w & F_MOVE) {  // fused LinearMove (flat_scene.h)
        const uint4 mv = sc.lo[++pc];
        lo_o = vsub(lo_o, smul(time, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
      }
      float t;
      if (sphere_hit_t(lo_o, d, u2f(lo.w), t_lo, best, t)) {
        if (rec) {
          V3 p = vadd(lo_o, smul(t, d));
          V3 n = sdiv(p, u2f(lo.w));
          if (hi.w & F_TRANSLATE) p = vadd(p, off);
          if (hi.w & F_FLIP) n = vneg(n);
          rec->t = t, rec->p = p, rec->n = n, rec->mat = hi.z;
        }
        best = t, any = true, tag = depth, nhits++;
      }
      pc++;
    } else if (op == OP_RECT) {
      if (COUNT) cnt.prim++;
      uint32_t axis = (hi.w >> F_AXIS_SHIFT) & 3u;
      float t;
      if (rect_hit_t(o, d, axis, u2f(lo.x), u2f(lo.y), u2f(lo.z), u2f(lo.w), u2f(hi.x), t_lo, best, t)) {
        if (rec) {
          V3 n = mk(axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f);
          if (hi.w & F_FLIP) n = vneg(n);
          rec->t = t, rec->p = vadd(o, smul(t, d)), rec->n = n, rec->mat = hi.z;
        }
        best = t, any = true, tag = depth, nhits++;
      }
      pc++;
    } else if (op == OP_PRISM) {
      if (COUNT) cnt.prim += 6;
      float t;
      uint32_t face = 0;
      const uint32_t nh = prism_hit_t(lo, hi, o, d, t_lo, best, t, face);
      if (nh) {
        if (rec) rec->t = t, rec->p = vadd(o, smul(t, d)), rec->n = prism_normal(face), rec->mat = hi.z;
        best = t, any = true, tag = depth, nhits += nh;
      }
      pc++;
    } else if (op == OP_PUSH) {
      uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      so[depth] = o, sd[depth] = d;
      depth++;
      V3 a = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
      if (hi.w & F_PRE_TRANSLATE) o = vsub(o, mk(u2f(lo.w), u2f(hi.x), u2f(hi.y)));
      if (kind == XF_TRANSLATE) {
        o = vsub(o, a);
      } else if (kind == XF_ROTATE_Y) {
        o = rot_y(o, -a.x, a.y), d = rot_y(d, -a.x, a.y);
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_SCALE) {
        o = vdiv(o, a), d = vdiv(d, a);
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_MOVE) {
        o = vsub(o, smul(time, a));
      }
      pc++;
    } else if (op == OP_POP) {
      uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      depth--;
      if (rec && any && tag == depth + 1) {  // the current best hit was found inside this wrapper
        V3 a = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
        if (kind == XF_TRANSLATE) {
          rec->p = vadd(rec->p, a);
        } else if (kind == XF_ROTATE_Y) {
          rec->p = rot_y(rec->p, a.x, a.y), rec->n = rot_y(rec->n, a.x, a.y);
        } else if (kind == XF_SCALE) {
          rec->p = vmul(rec->p, a), rec->n = vdiv(rec->n, a);
        } else if (kind == XF_FLIP) {
          rec->n = vneg(rec->n);
        }
        if (hi.w & F_PRE_TRANSLATE) rec->p = vadd(rec->p, mk(u2f(lo.w), u2f(hi.x), u2f(hi.y)));
      }
      if (any && tag == depth + 1) tag = depth;
      o = so[depth], d = sd[depth];
      if (kind == XF_ROTATE_Y || kind == XF_SCALE) inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      pc++;
    } else if (op == OP_MEDIUM) {  // ConstantMedium::hit, object.rs:545-575
      const bool general = (hi.w & F_GENERAL_BOUNDARY) != 0u;
      const uint4 blo = sc.lo[pc + 1], bhi = sc.hi[pc + 1];
      float t1 = 0.f, t2 = 0.f;
      bool h1 = false, h2 = false;
      if (general) {
        if constexpr (LEVEL < ML) h1 = walk_deep<LEVEL + 1, COUNT, XD, ML>(sc, pc + 1, hi.x, o, d, time, -F32_MAX, F32_MAX, rng, nullptr, t1, cnt);
      } else {
        if (COUNT) cnt.prim++;
        h1 = prim_hit_t(blo, bhi, o, d, -F32_MAX, F32_MAX, t1);
      }
      if (h1) {
        if (general) {
          if constexpr (LEVEL < ML) h2 = walk_deep<LEVEL + 1, COUNT, XD, ML>(sc, pc + 1, hi.x, o, d, time, t1 + 0.0001f, F32_MAX, rng, nullptr, t2, cnt);
        } else {
          if (COUNT) cnt.prim++;
          h2 = prim_hit_t(blo, bhi, o, d, t1 + 0.0001f, F32_MAX, t2);
        }
        if (h2) {
          t1 = rs_max(t1, t_lo);
          t2 = rs_min(t2, best);
          if (!(t1 >= t2)) {
            float distance_inside = (t2 - t1) * vlen(d);
            float hit_distance = -(1.f / u2f(lo.x)) * rt_logf(rng.gen_f32());
            if (hit_distance < distance_inside) {
              float t = t1 + hit_distance / vlen(d);
              bool accept = !(hi.w & F_UNDER_BVH) || nhits == root_hits || !(best < t);  // merge rule: see hit_top
              if (accept) {
                if (rec) rec->t = t, rec->p = vadd(o, smul(t, d)), rec->n = mk(1.f, 0.f, 0.f), rec->mat = hi.z;
                best = t, any = true, tag = depth, nhits++;
              }
            }
          }
        }
      }
      pc = hi.x;  // first record after the boundary's stream
    } else if (op == OP_SAVE) {
      sv_best[sp] = best, sv_any[sp] = any, sv_tag[sp] = tag, sv_nhits[sp] = nhits, sv_root[sp] = root_hits;
      if (rec) sv_rec[sp] = *rec;
      sp++;
      pc++;
    } else if (op == OP_MERGE) {  // bvh.rs:104-112 for the leaf that ends here: (Some(hl), Some(hr)) => if hl.t < hr.t { hl } else { hr }
      sp--;
      const bool leaf_hit = nhits != sv_nhits[sp];            // hr
      const bool earlier = sv_nhits[sp] != sv_root[sp];       // hl: a hit inside the same outermost Bvh, in front of this leaf
      if (leaf_hit && earlier && sv_best[sp] < best) {
        best = sv_best[sp], tag = sv_tag[sp];
        if (rec) *rec = sv_rec[sp];
      }
      root_hits = sv_root[sp];
      pc++;
    } else {
      pc++;  // OP_BEND (end of a boundary stream)
    }
  }
  t_out = best;
  return any;
}

// World::hit_top (lib.rs:33-55) over the flat program.  `best` plays `nearest` / the shrinking
// t_range.end; t_range.start is always t_near.  Returns Some/None, fills `rec` (world space).
template <uint32_t FEAT, bool COUNT>
RT_DEV bool hit_top(const DevScene& sc, V3 o, V3 d, float time, float t_near, SampleRng& rng,
                    HitRec& rec, Counts& cnt) {
  if (COUNT) cnt.rays++;
  if (FEAT & FEAT_DEEP) {  // graph shapes only the general walk handles
    float t;
    constexpr int XD = (FEAT & FEAT_DEEP_FEW_WRAPPERS) ? DEEP_FEW_WRAPPERS : MAX_DEEP_XFORM_DEPTH;
    constexpr int ML = (FEAT & FEAT_DEEP_ONE_LEVEL) ? 1 : MAX_MEDIUM_NESTING;
    return walk_deep<0, COUNT, XD, ML>(sc, 0u, sc.n_prog, o, d, time, t_near, F32_MAX, rng, &rec, t, cnt);
  }
  V3 inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);  // aabb.rs:17, hoisted: depends on the ray only
  float best = F32_MAX;
  bool any = false;
  int depth = 0, tag = 0;
  uint32_t nhits = 0, root_hits = 0;  // hits accepted so far / at entry of the current outermost Bvh
  V3 so[MAX_XFORM_DEPTH], sd[MAX_XFORM_DEPTH];
  uint32_t pc = 0;
  for (;;) {
    const uint4 hi = sc.hi[pc];
    const uint32_t op = hi.w & 0xffu;
    if (op == OP_END) break;
    const uint4 lo = sc.lo[pc];
    if (op == OP_BOX) {  // Aabb::hit, aabb.rs:16-27
      if (COUNT) cnt.aabb++;
      if ((FEAT & FEAT_MEDIUM) && (hi.w & F_BVH_ROOT)) root_hits = nhits;
      float t0x = (u2f(lo.x) - o.x) * inv.x, t1x = (u2f(lo.y) - o.x) * inv.x;
      float t0y = (u2f(lo.z) - o.y) * inv.y, t1y = (u2f(lo.w) - o.y) * inv.y;
      float t0z = (u2f(hi.x) - o.z) * inv.z, t1z = (u2f(hi.y) - o.z) * inv.z;
      float ax = inv.x < 0.f ? t1x : t0x, bx = inv.x < 0.f ? t0x : t1x;
      float ay = inv.y < 0.f ? t1y : t0y, by = inv.y < 0.f ? t0y : t1y;
      float az = inv.z < 0.f ? t1z : t0z, bz = inv.z < 0.f ? t0z : t1z;
      float start = rs_max(t_near, rs_max(rs_max(ax, ay), az));
      float end = rs_min(best, rs_min(rs_min(bx, by), bz));
      pc = (end > start) ? pc + 1 : hi.z;
      continue;
    }
    if (op == OP_SPHERE) {
      if (COUNT) cnt.prim++;
      V3 off = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
      V3 lo_o = o;
      if (hi.w & F_TRANSLATE) lo_o = vsub(o, off);  // object.rs:275-278
      if ((FEAT & FEAT_XFORM) && (hi.w & F_MOVE)) {  // fused LinearMove, object.rs:505-508 (flat_scene.h)
        const uint4 mv = sc.lo[++pc];
        lo_o = vsub(lo_o, smul(time, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
      }
      float t;
      if (sphere_hit_t(lo_o, d, u2f(lo.w), t_near, best, t)) {
        V3 p = vadd(lo_o, smul(t, d));              // ray.rs:15
        V3 n = sdiv(p, u2f(lo.w));                  // object.rs:104
        if (hi.w & F_TRANSLATE) p = vadd(p, off);   // object.rs:279-282
        if (hi.w & F_FLIP) n = vneg(n);             // object.rs:249-252
        rec.t = t, rec.p = p, rec.n = n, rec.mat = hi.z;
        best = t, any = true, tag = depth, nhits++;
      }
      pc++;
      continue;
    }
    if ((FEAT & FEAT_RECT) && op == OP_RECT) {
      if (COUNT) cnt.prim++;
      uint32_t axis = (hi.w >> F_AXIS_SHIFT) & 3u;
      float t;
      if (rect_hit_t(o, d, axis, u2f(lo.x), u2f(lo.y), u2f(lo.z), u2f(lo.w), u2f(hi.x), t_near, best, t)) {
        V3 n = mk(axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f);
        if (hi.w & F_FLIP) n = vneg(n);
        rec.t = t, rec.p = vadd(o, smul(t, d)), rec.n = n, rec.mat = hi.z;
        best = t, any = true, tag = depth, nhits++;
      }
      pc++;
      continue;
    }
    if ((FEAT & FEAT_RECT) && op == OP_PRISM) {
      if (COUNT) cnt.prim += 6;
      float t;
      uint32_t face = 0;
      const uint32_t nh = prism_hit_t(lo, hi, o, d, t_near, best, t, face);
      if (nh) {
        rec.t = t, rec.p = vadd(o, smul(t, d)), rec.n = prism_normal(face), rec.mat = hi.z;
        best = t, any = true, tag = depth, nhits += nh;
      }
      pc++;
      continue;
    }
    if ((FEAT & FEAT_XFORM) && op == OP_PUSH) {
      uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      so[depth] = o, sd[depth] = d;
      depth++;
      V3 a = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
      if (hi.w & F_PRE_TRANSLATE) o = vsub(o, mk(u2f(lo.w), u2f(hi.x), u2f(hi.y)));  // the enclosing Translate, object.rs:275-278
      if (kind == XF_TRANSLATE) {
        o = vsub(o, a);                                  // object.rs:275-278
      } else if (kind == XF_ROTATE_Y) {
        o = rot_y(o, -a.x, a.y), d = rot_y(d, -a.x, a.y);  // object.rs:357-361
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_SCALE) {
        o = vdiv(o, a), d = vdiv(d, a);                  // object.rs:309-313
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      } else if (kind == XF_MOVE) {
        o = vsub(o, smul(time, a));                      // object.rs:505-508
      }
      pc++;
      continue;
    }
    if ((FEAT & FEAT_XFORM) && op == OP_POP) {
      uint32_t kind = (hi.w >> F_KIND_SHIFT) & 7u;
      depth--;
      if (any && tag == depth + 1) {  // the current best hit was found inside this wrapper
        V3 a = mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));
        if (kind == XF_TRANSLATE) {
          rec.p = vadd(rec.p, a);                         // object.rs:279-282
        } else if (kind == XF_ROTATE_Y) {
          rec.p = rot_y(rec.p, a.x, a.y), rec.n = rot_y(rec.n, a.x, a.y);  // object.rs:365-369
        } else if (kind == XF_SCALE) {
          rec.p = vmul(rec.p, a), rec.n = vdiv(rec.n, a); // object.rs:314-318
        } else if (kind == XF_FLIP) {
          rec.n = vneg(rec.n);                            // object.rs:249-252
        }                                                 // XF_MOVE: hit is NOT moved back (object.rs:504-511)
        if (hi.w & F_PRE_TRANSLATE) rec.p = vadd(rec.p, mk(u2f(lo.w), u2f(hi.x), u2f(hi.y)));  // object.rs:279-282
        tag = depth;
      }
      o = so[depth], d = sd[depth];
      if (kind == XF_ROTATE_Y || kind == XF_SCALE) inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      pc++;
      continue;
    }
    if ((FEAT & FEAT_MEDIUM) && op == OP_MEDIUM) {  // ConstantMedium::hit, object.rs:545-575
      const bool general = (hi.w & F_GENERAL_BOUNDARY) != 0u;
      const uint4 blo = sc.lo[pc + 1], bhi = sc.hi[pc + 1];
      float t1, t2;
      bool h1, h2 = false;
      if (general) {
        const BoundaryHit b1 = boundary_hit_t(sc.lo, sc.hi, pc + 1, hi.x, o, d, time, -F32_MAX, F32_MAX);
        h1 = b1.any != 0u, t1 = b1.t;
        if (COUNT) cnt.aabb += b1.n_aabb, cnt.prim += b1.n_prim;
      } else {
        if (COUNT) cnt.prim++;
        h1 = prim_hit_t(blo, bhi, o, d, -F32_MAX, F32_MAX, t1);
      }
      if (h1) {
        if (general) {
          const BoundaryHit b2 = boundary_hit_t(sc.lo, sc.hi, pc + 1, hi.x, o, d, time, t1 + 0.0001f, F32_MAX);
          h2 = b2.any != 0u, t2 = b2.t;
          if (COUNT) cnt.aabb += b2.n_aabb, cnt.prim += b2.n_prim;
        } else {
          if (COUNT) cnt.prim++;
          h2 = prim_hit_t(blo, bhi, o, d, t1 + 0.0001f, F32_MAX, t2);
        }
        if (h2) {
          t1 = rs_max(t1, t_near);
          t2 = rs_min(t2, best);
          if (!(t1 >= t2)) {
            float distance_inside = (t2 - t1) * vlen(d);
            float hit_distance = -(1.f / u2f(lo.x)) * rt_logf(rng.gen_f32());
            if (hit_distance < distance_inside) {
              float t = t1 + hit_distance / vlen(d);
              // Merge rule: in a list / And the later hit always replaces (lib.rs:41-44,
              // object.rs:409).  Below a Bvh, when an earlier hit was found inside the same
              // outermost Bvh, that hit is some ancestor's `hl` and this one its `hr`: hr loses
              // when hl.t < hr.t (bvh.rs:104-112).  Only a medium can return t >= t_range.end.
              bool accept = !(hi.w & F_UNDER_BVH) || nhits == root_hits || !(best < t);
              if (accept) {
                rec.t = t, rec.p = vadd(o, smul(t, d)), rec.n = mk(1.f, 0.f, 0.f), rec.mat = hi.z;
                best = t, any = true, tag = depth, nhits++;
              }
            }
          }
        }
      }
      pc = hi.x;  // first record after the boundary's stream
      continue;
    }
    pc++;  // OP_SEG: this interpreter hoists nothing -- it steps over the record and executes the segment's primitives (flat_scene.h)
  }
  return any;
}

// ---- textures (texture.rs, perlin.rs) ------------------------------------------------------------
RT_DEV int32_t f32_as_i32(float f) {  // Rust `as i32`: saturating, NaN -> 0
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (int32_t)0x80000000;
  return (int32_t)f;
}

__device__ __attribute__((noinline)) float perlin_noise(const DevScene& sc, V3 p) {  // perlin.rs:49-64 + :31-47
  V3 ijk = mk(__builtin_floorf(p.x), __builtin_floorf(p.y), __builtin_floorf(p.z));
  V3 uvw = vsub(p, ijk);
  uint32_t bi = (uint32_t)f32_as_i32(ijk.x), bj = (uint32_t)f32_as_i32(ijk.y), bk = (uint32_t)f32_as_i32(ijk.z);
  V3 uvw3 = vmul(vmul(uvw, uvw), vsub(splat(3.f), smul(2.f, uvw)));
  V3 uvw3_inv = vsub(splat(1.f), uvw3);
  float accum = 0.f;
#pragma unroll 1
  for (uint32_t i = 0; i < 2; i++)
#pragma unroll 1
    for (uint32_t j = 0; j < 2; j++)
#pragma unroll 1
      for (uint32_t k = 0; k < 2; k++) {
        uint32_t ix = sc.perlin_perm[(bi + i) & 255u];
        uint32_t iy = sc.perlin_perm[256u + ((bj + j) & 255u)];
        uint32_t iz = sc.perlin_perm[512u + ((bk + k) & 255u)];
        float4 g = sc.perlin_vecs[ix ^ iy ^ iz];
        V3 f = mk((float)i, (float)j, (float)k);
        float weight = vdot(mk(g.x, g.y, g.z), vsub(uvw, f));
        V3 f_inv = vsub(splat(1.f), f);
        V3 m = vadd(vmul(f, uvw3), vmul(f_inv, uvw3_inv));
        accum = accum + ((m.x * m.y) * m.z) * weight;
      }
  return accum;
}

RT_DEV float perlin_turb(const DevScene& sc, V3 p, int depth) {  // perlin.rs:66-75
  float accum = 0.f, weight = 1.f;
#pragma unroll 1
  for (int i = 0; i < depth; i++) {
    accum += weight * perlin_noise(sc, p);
    weight *= 0.5f;
    p = smul(2.f, p);
  }
  return __builtin_fabsf(accum);
}

// out of line on purpose: Perlin turbulence (7 octaves x 8 gradient fetches) is a rare path and would
// otherwise dominate the register allocation of every kernel that can reach it
__device__ __attribute__((noinline)) V3 texture_eval(const DevScene& sc, uint32_t idx, V3 p) {
  for (;;) {
    uint4 lo = sc.tex[2 * idx], hi = sc.tex[2 * idx + 1];
    uint32_t kind = hi.w;
    if (kind == TEX_CONSTANT) return mk(u2f(lo.x), u2f(lo.y), u2f(lo.z));           // texture.rs:8
    if (kind == TEX_PERLIN) return splat(perlin_turb(sc, smul(u2f(lo.w), p), 7));   // texture.rs:23
    V3 q = smul(10.f, p);                                                           // texture.rs:12-21
    float s = (rt_sinf(q.x) * rt_sinf(q.y)) * rt_sinf(q.z);
    idx = s < 0.f ? hi.y : hi.x;
  }
}

// material's texture value at p: inlined colour for constant textures, table walk otherwise
template <uint32_t FEAT>
RT_DEV V3 material_texture(const DevScene& sc, uint4 mlo, uint4 mhi, V3 p) {
  if ((FEAT & FEAT_TEXTURE) && ((mhi.w >> 8) & 0xffu) != TEX_CONSTANT) return texture_eval(sc, mhi.x, p);
  return mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
}

template <bool INLINE_POW = false>
RT_DEV float schlick(float cos, float ref_idx) {  // material.rs:142-146
  float r0 = (1.f - ref_idx) / (1.f + ref_idx);
  r0 = r0 * r0;
  return r0 + (1.f - r0) * (INLINE_POW ? rt_pow5f_inline(1.f - cos) : rt_pow5f(1.f - cos));
}

// camera.rs:52-63
RT_DEV void get_ray(const DevCamera& cam, float s, float t, SampleRng& rng, V3& o, V3& d, float& time) {
  V3 rd = smul(cam.lens_radius, in_unit_disc(rng));
  V3 offset = vadd(smul(rd.x, cam.u), smul(rd.y, cam.v));
  time = rng.gen_range(cam.e0, cam.e1);
  o = vadd(cam.origin, offset);
  d = vsub(vsub(vadd(vadd(cam.llc, smul(s, cam.horizontal)), smul(t, cam.vertical)), cam.origin), offset);
}

// color(), lib.rs:60-101, with Material::{emitted,scatter} (material.rs:55-128) inlined.
template <uint32_t FEAT, bool COUNT>
RT_DEV V3 color(const DevScene& sc, V3 o, V3 d, float time, const DevParams& P, SampleRng& rng,
                Counts& cnt, uint32_t& bounces_out) {
  V3 accum = mk(0.f, 0.f, 0.f);
  V3 strength = splat(1.f);
  uint32_t bounces = 0;
  HitRec hit;
  rng.set_event(1);
  while (hit_top<FEAT, COUNT>(sc, o, d, time, P.t_near, rng, hit, cnt)) {
    if (COUNT) cnt.shaded++;
    const uint4 mlo = sc.mat[2 * hit.mat], mhi = sc.mat[2 * hit.mat + 1];
    const uint32_t kind = mhi.w & 0xffu;
    const float param = u2f(mlo.w);
    V3 emitted = mk(0.f, 0.f, 0.f);  // material.rs:120-128
    if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, material_texture<FEAT>(sc, mlo, mhi, hit.p));
    accum = vadd(accum, vmul(strength, emitted));  // lib.rs:76
    V3 nd, att;
    bool scattered = true;
    if (kind == MAT_LAMBERTIAN) {  // material.rs:57-65
      V3 target = vadd(vadd(hit.p, hit.n), in_unit_sphere(rng));
      nd = vsub(target, hit.p);
      att = material_texture<FEAT>(sc, mlo, mhi, hit.p);
    } else if (kind == MAT_METAL) {  // material.rs:66-80
      V3 refl = reflect(vunit(d), hit.n);
      nd = vadd(refl, smul(param, in_unit_sphere(rng)));
      att = mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
      scattered = vdot(nd, hit.n) > 0.f;
    } else if (kind == MAT_DIELECTRIC) {  // material.rs:81-107
      V3 outward;
      float ni_over_nt, cosine;
      float dn = vdot(d, hit.n);
      if (dn > 0.f) {
        outward = vneg(hit.n);
        ni_over_nt = param;
        cosine = param * dn / vlen(d);
      } else {
        outward = hit.n;
        ni_over_nt = 1.0f / param;
        cosine = -dn / vlen(d);
      }
      // refract, vec3.rs:321-330
      V3 uv = vunit(d);
      float dt = vdot(uv, outward);
      float disc = 1.0f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
      bool refracted = disc > 0.f;
      if (refracted) {
        nd = vsub(smul(ni_over_nt, vsub(uv, smul(dt, outward))), smul(__builtin_sqrtf(disc), outward));
        refracted = rng.gen_f32() >= schlick(cosine, param);  // material.rs:97: draw only if Some
      }
      if (!refracted) nd = reflect(d, hit.n);
      att = splat(1.f);
    } else if (kind == MAT_DIFFUSE_LIGHT) {  // material.rs:108
      scattered = false;
    } else {  // Isotropic, material.rs:109-116
      nd = in_unit_sphere(rng);
      att = material_texture<FEAT>(sc, mlo, mhi, hit.p);
    }
    if (!scattered) {  // lib.rs:88-91
      bounces_out = bounces;
      return accum;
    }
    o = hit.p, d = nd;  // time is carried over by every material
    strength = vmul(strength, att);
    if (bounces == P.max_bounces) {  // lib.rs:93-95
      bounces_out = bounces;
      return accum;
    }
    bounces += 1;
    rng.set_event(bounces + 1);
  }
  bounces_out = bounces;
  return mk(0.f, 0.f, 0.f);  // lib.rs:100
}

// One sample of par_cast's closure, lib.rs:366-372.
template <uint32_t FEAT, bool COUNT>
RT_DEV V3 sample_color(const DevScene& sc, const DevCamera& cam, const DevParams& P, uint32_t x, uint32_t y,
                       uint32_t s, Counts& cnt, uint32_t& bounces, uint32_t& draws) {
  SampleRng rng;
  rng.init(((uint64_t)P.seed_hi << 32) | P.seed_lo, y * P.nx + x, s);
  float u = ((float)x + rng.gen_f32()) / (float)P.nx;
  float v = ((float)y + rng.gen_f32()) / (float)P.ny;
  V3 o, d;
  float time;
  get_ray(cam, u, v, rng, o, d, time);
  V3 c = color<FEAT, COUNT>(sc, o, d, time, P, rng, cnt, bounces);
  draws = rng.draws;
  return c;
}

}  // namespace rtg
