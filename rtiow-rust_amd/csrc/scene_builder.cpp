// scene_builder.cpp -- host-side scene graph, Bvh::new, bounding boxes and the flattener.
// Compiled with -ffp-contract=off: the f32 arithmetic here (boxes, centroids) must round exactly as
// the reference's would.
#include "scene_builder.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rtg {

namespace {
constexpr float kF32Max = std::numeric_limits<float>::max();
constexpr uint32_t kNone = 0xffffffffu;

// Rust f32::min / f32::max (NaN-ignoring), used by Aabb::merge (aabb.rs:9-14) and Bvh::new.
inline float fmin_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float fmax_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

Box3 merge(const Box3& p, const Box3& q) {
  Box3 r;
  for (int i = 0; i < 3; i++) {
    r.mn[i] = fmin_rs(p.mn[i], q.mn[i]);
    r.mx[i] = fmax_rs(p.mx[i], q.mx[i]);
  }
  return r;
}

uint32_t fbits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

// object.rs:373-379: rot(c) = (c.(cos,0,sin), c.(0,1,0), c.(-sin,0,cos)), dot = (x*x'+y*y')+z*z'
void rot_y(const float p[3], float s, float c, float out[3]) {
  out[0] = (p[0] * c + p[1] * 0.f) + p[2] * s;
  out[1] = (p[0] * 0.f + p[1] * 1.f) + p[2] * 0.f;
  out[2] = (p[0] * (-s) + p[1] * 0.f) + p[2] * c;
}
}  // namespace

uint32_t SceneBuilder::add_object(const HostObject& o) {
  auto valid = [&](uint32_t id) { return id < objects.size(); };
  switch (o.kind) {
    case HostObject::SPHERE:
    case HostObject::RECT:
      if (o.mat >= materials.size()) throw BuildError{-1, "object: bad material handle"};
      if (o.kind == HostObject::RECT && (o.axis < 0 || o.axis > 2)) throw BuildError{-1, "rect: bad axis"};
      break;
    case HostObject::AND:
      if (!valid(o.a) || !valid(o.b)) throw BuildError{-1, "and: bad object handle"};
      break;
    case HostObject::MEDIUM:
      if (!valid(o.a)) throw BuildError{-1, "constant_medium: bad boundary handle"};
      if (o.mat >= materials.size()) throw BuildError{-1, "constant_medium: bad material handle"};
      break;
    case HostObject::BVH: break;
    default:
      if (!valid(o.a)) throw BuildError{-1, "wrapper: bad object handle"};
  }
  objects.push_back(o);
  return (uint32_t)objects.size() - 1;
}

// Object::bounding_box for every object kind (object.rs:113,220,255,285,321,372,412,514,577; bvh.rs:122)
Box3 SceneBuilder::bounding_box(uint32_t id, float e0, float e1) const {
  const HostObject& o = objects[id];
  Box3 r;
  switch (o.kind) {
    case HostObject::SPHERE:
      for (int i = 0; i < 3; i++) r.mn[i] = -o.f[0], r.mx[i] = o.f[0];
      return r;
    case HostObject::RECT: {
      int o1 = o.axis == 0 ? 1 : 0, o2 = o.axis == 2 ? 1 : 2;
      r.mn[o.axis] = o.f[0] - 0.0001f;
      r.mx[o.axis] = o.f[0] + 0.0001f;
      r.mn[o1] = o.f[1], r.mx[o1] = o.f[2];
      r.mn[o2] = o.f[3], r.mx[o2] = o.f[4];
      return r;
    }
    case HostObject::FLIP: return bounding_box(o.a, e0, e1);
    case HostObject::TRANSLATE: {
      Box3 b = bounding_box(o.a, e0, e1);
      for (int i = 0; i < 3; i++) r.mn[i] = b.mn[i] + o.f[i], r.mx[i] = b.mx[i] + o.f[i];
      return r;
    }
    case HostObject::SCALE: {
      Box3 b = bounding_box(o.a, e0, e1);
      for (int i = 0; i < 3; i++) r.mn[i] = b.mn[i] * o.f[i], r.mx[i] = b.mx[i] * o.f[i];
      return r;
    }
    case HostObject::ROTATE_Y: {
      Box3 b = bounding_box(o.a, e0, e1);
      for (int i = 0; i < 3; i++) r.mn[i] = kF32Max, r.mx[i] = -kF32Max;
      for (int c = 0; c < 8; c++) {  // aabb.rs:29-43: x outermost, z innermost
        float p[3] = {(c & 4) ? b.mx[0] : b.mn[0], (c & 2) ? b.mx[1] : b.mn[1], (c & 1) ? b.mx[2] : b.mn[2]};
        float q[3];
        rot_y(p, o.f[0], o.f[1], q);
        for (int i = 0; i < 3; i++) r.mn[i] = fmin_rs(r.mn[i], q[i]), r.mx[i] = fmax_rs(r.mx[i], q[i]);
      }
      return r;
    }
    case HostObject::AND: return merge(bounding_box(o.a, e0, e1), bounding_box(o.b, e0, e1));
    case HostObject::MOVE: {
      Box3 b = bounding_box(o.a, e0, e1), s, t;
      for (int i = 0; i < 3; i++) {
        s.mn[i] = b.mn[i] + e0 * o.f[i], s.mx[i] = b.mx[i] + e0 * o.f[i];
        t.mn[i] = b.mn[i] + e1 * o.f[i], t.mx[i] = b.mx[i] + e1 * o.f[i];
      }
      return merge(s, t);
    }
    case HostObject::MEDIUM: return bounding_box(o.a, e0, e1);
    default: return bvh_nodes[o.a].box;  // BVH
  }
}

// Bvh::new, bvh.rs:22-81.  Widest-axis median split; the reference's sort_unstable_by tie order is
// rustc-specific, this build's documented tie rule is "stable" (SURVEY.md a17).
int32_t SceneBuilder::build_bvh(std::vector<uint32_t> objs, float e0, float e1) {
  float extent[3];
  for (int axis = 0; axis < 3; axis++) {  // bvh.rs:27-35
    float lo = kF32Max, hi = -kF32Max;
    for (uint32_t id : objs) {
      Box3 bb = bounding_box(id, e0, e1);
      lo = fmin_rs(lo, fmin_rs(bb.mn[axis], bb.mx[axis]));
      hi = fmax_rs(hi, fmax_rs(bb.mn[axis], bb.mx[axis]));
    }
    extent[axis] = hi - lo;
    if (extent[axis] != extent[axis]) throw BuildError{-3, "Bvh::new: NaN extent (partial_cmp().unwrap())"};
  }
  int axis = 0;  // bvh.rs:38-47: descending sort of three, first maximum wins
  if (extent[1] > extent[axis]) axis = 1;
  if (extent[2] > extent[axis]) axis = 2;

  std::vector<std::pair<float, uint32_t>> keyed;  // bvh.rs:51-57
  keyed.reserve(objs.size());
  for (uint32_t id : objs) {
    Box3 bb = bounding_box(id, e0, e1);
    float key = bb.mn[axis] + bb.mx[axis];
    if (key != key) throw BuildError{-3, "Bvh::new: NaN centroid (partial_cmp().unwrap())"};
    keyed.emplace_back(key, id);
  }
  std::stable_sort(keyed.begin(), keyed.end(),
                   [](const std::pair<float, uint32_t>& p, const std::pair<float, uint32_t>& q) {
                     return p.first < q.first;
                   });
  HostBvhNode node;
  if (keyed.size() == 1) {  // bvh.rs:61-65
    node.box = bounding_box(keyed[0].second, e0, e1);
    node.leaf = keyed[0].second;
  } else {  // bvh.rs:66-79
    size_t half = keyed.size() / 2;
    std::vector<uint32_t> l, r;
    for (size_t i = 0; i < keyed.size(); i++) (i < half ? l : r).push_back(keyed[i].second);
    node.right = build_bvh(std::move(r), e0, e1);
    node.left = build_bvh(std::move(l), e0, e1);
    node.box = merge(bvh_nodes[node.left].box, bvh_nodes[node.right].box);
  }
  bvh_nodes.push_back(node);
  return (int32_t)bvh_nodes.size() - 1;
}

// NOT IN THE REFERENCE: surface-area-heuristic split instead of the widest-axis median.
// Same node type and one object per leaf, so the flattened program and its traversal are unchanged; only
// the tree shape differs (closest-hit results are invariant except at exact-t ties).  Full sweep on every
// axis over centroid-sorted objects, cost = SA(left) * n_left + SA(right) * n_right, strict `<` keeps the
// lower axis / earlier split on ties, stable sorts -- deterministic and mirrored by the test oracle.
int32_t SceneBuilder::build_bvh_sah(std::vector<uint32_t> objs, float e0, float e1) {
  auto half_area = [](const Box3& b) {
    float ex = b.mx[0] - b.mn[0], ey = b.mx[1] - b.mn[1], ez = b.mx[2] - b.mn[2];
    return (ex * ey + ey * ez) + ez * ex;
  };
  HostBvhNode node;
  const size_t n = objs.size();
  if (n == 1) {
    node.box = bounding_box(objs[0], e0, e1);
    node.leaf = objs[0];
    bvh_nodes.push_back(node);
    return (int32_t)bvh_nodes.size() - 1;
  }
  float best_cost = 0.f;
  int best_axis = -1;
  size_t best_split = 0;
  std::vector<uint32_t> best_order;
  for (int axis = 0; axis < 3; axis++) {
    std::vector<std::pair<float, uint32_t>> keyed;
    keyed.reserve(n);
    for (uint32_t id : objs) {
      Box3 bb = bounding_box(id, e0, e1);
      float key = bb.mn[axis] + bb.mx[axis];
      if (key != key) throw BuildError{-3, "Bvh (sah): NaN centroid"};
      keyed.emplace_back(key, id);
    }
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const std::pair<float, uint32_t>& p, const std::pair<float, uint32_t>& q) { return p.first < q.first; });
    std::vector<float> right_area(n);
    Box3 acc = bounding_box(keyed[n - 1].second, e0, e1);
    for (size_t i = n - 1; i >= 1; i--) {
      if (i != n - 1) acc = merge(acc, bounding_box(keyed[i].second, e0, e1));
      right_area[i] = half_area(acc);
    }
    Box3 left = bounding_box(keyed[0].second, e0, e1);
    for (size_t i = 1; i < n; i++) {  // split [0, i) | [i, n)
      if (i > 1) left = merge(left, bounding_box(keyed[i - 1].second, e0, e1));
      float cost = half_area(left) * (float)i + right_area[i] * (float)(n - i);
      if (best_axis < 0 || cost < best_cost) {
        best_cost = cost, best_axis = axis, best_split = i;
        best_order.clear();
        for (auto& k : keyed) best_order.push_back(k.second);
      }
    }
  }
  std::vector<uint32_t> l(best_order.begin(), best_order.begin() + best_split), r(best_order.begin() + best_split, best_order.end());
  node.right = build_bvh_sah(std::move(r), e0, e1);
  node.left = build_bvh_sah(std::move(l), e0, e1);
  node.box = merge(bvh_nodes[node.left].box, bvh_nodes[node.right].box);
  bvh_nodes.push_back(node);
  return (int32_t)bvh_nodes.size() - 1;
}

uint32_t SceneBuilder::add_bvh(const uint32_t* objs, size_t n, float e0, float e1, bool sah) {
  if (n == 0) throw BuildError{-2, "Can't create a BVH from zero objects."};  // bvh.rs:60
  std::vector<uint32_t> v(objs, objs + n);
  for (uint32_t id : v)
    if (id >= objects.size()) throw BuildError{-1, "bvh: bad object handle"};
  HostObject o;
  o.kind = HostObject::BVH;
  o.a = (uint32_t)(sah ? build_bvh_sah(std::move(v), e0, e1) : build_bvh(std::move(v), e0, e1));
  objects.push_back(o);
  return (uint32_t)objects.size() - 1;
}

// ---- flattening ------------------------------------------------------------------------------------
namespace {
void push(FlatScene* s, float a, float b, float c, float d, uint32_t e, uint32_t f, uint32_t g, uint32_t h) {
  s->lo.push_back(Packet{{fbits(a), fbits(b), fbits(c), fbits(d)}});
  s->hi.push_back(Packet{{e, f, g, h}});
}
}  // namespace

// copies of material facts into a primitive's flag word: MatKind and "reads a non-constant texture"
static uint32_t mat_flags(const SceneBuilder& b, uint32_t mat) {
  const HostMaterial& m = b.materials[mat];
  const bool textured = m.tex != 0xffffffffu && b.textures[m.tex].kind != TEX_CONSTANT;
  return (m.kind << F_MATKIND_SHIFT) | (textured ? F_TEXTURED : 0u);
}

// Peel FlipNormals* [Translate] FlipNormals* [LinearMove] FlipNormals* Sphere, or FlipNormals* Rect, into one fused record.
// (FlipNormals commutes exactly with Translate and LinearMove: one negates the normal, the others shift the origin / p.
// The LinearMove must sit INSIDE the Translate -- the order of the two subtractions is the reference's, object.rs:275-278 then
// 505-508 -- and is only peeled where the caller can execute it: `allow_move`.)
static bool fuse_primitive(const SceneBuilder& b, uint32_t id, FlatScene* out, bool emit_it, bool allow_move) {
  bool flip = false, have_t = false, have_m = false;
  float off[3] = {0, 0, 0}, motion[3] = {0, 0, 0};
  for (;;) {
    const HostObject& o = b.objects[id];
    if (o.kind == HostObject::FLIP) {
      flip = !flip;
      id = o.a;
    } else if (o.kind == HostObject::TRANSLATE && !have_t && !have_m) {
      have_t = true;
      off[0] = o.f[0], off[1] = o.f[1], off[2] = o.f[2];
      id = o.a;
    } else if (o.kind == HostObject::MOVE && !have_m && allow_move) {
      have_m = true;
      motion[0] = o.f[0], motion[1] = o.f[1], motion[2] = o.f[2];
      id = o.a;
    } else if (o.kind == HostObject::SPHERE) {
      if (emit_it) {
        push(out, off[0], off[1], off[2], o.f[0], 0, 0, o.mat,
             OP_SPHERE | (have_t ? F_TRANSLATE : 0u) | (have_m ? F_MOVE : 0u) | (flip ? F_FLIP : 0u) | mat_flags(b, o.mat));
        if (have_m) {
          push(out, motion[0], motion[1], motion[2], 0, 0, 0, 0, OP_EXT);
          out->features |= FEAT_XFORM;  // the ray's time matters: not a lean program
        }
      }
      return true;
    } else if (o.kind == HostObject::RECT && !have_t && !have_m) {
      if (emit_it) {
        push(out, o.f[0], o.f[1], o.f[2], o.f[3], fbits(o.f[4]), 0, o.mat,
             OP_RECT | ((uint32_t)o.axis << F_AXIS_SHIFT) | (flip ? F_FLIP : 0u) | mat_flags(b, o.mat));
        out->features |= FEAT_RECT;
      }
      return true;
    } else {
      return false;
    }
  }
}

// Is `id` exactly the And-tree rect_prism(p0, p1, material) builds (object.rs:420-473)?  Bit-compares
// every Rect field, so anything else (different materials, hand-made boxes) stays six RECT records.
static bool match_prism(const SceneBuilder& b, uint32_t id, float p0[3], float p1[3], uint32_t* mat) {
  auto obj = [&](uint32_t i) -> const HostObject& { return b.objects[i]; };
  auto is = [&](uint32_t i, HostObject::Kind k) { return i < b.objects.size() && obj(i).kind == k; };
  if (!is(id, HostObject::AND)) return false;
  const uint32_t pos = obj(id).a, neg = obj(id).b;
  if (!is(pos, HostObject::AND) || !is(neg, HostObject::AND)) return false;
  if (!is(obj(pos).b, HostObject::AND) || !is(obj(neg).b, HostObject::AND)) return false;
  uint32_t face[6] = {obj(pos).a, obj(obj(pos).b).a, obj(obj(pos).b).b, obj(neg).a, obj(obj(neg).b).a, obj(obj(neg).b).b};
  for (int i = 3; i < 6; i++) {
    if (!is(face[i], HostObject::FLIP)) return false;
    face[i] = obj(face[i]).a;
  }
  for (int i = 0; i < 6; i++)
    if (!is(face[i], HostObject::RECT) || obj(face[i]).axis != 2 - (i % 3) || obj(face[i]).mat != obj(face[0]).mat) return false;
  const HostObject &zp = obj(face[0]), &yp = obj(face[1]);
  // Rect fields: f = (k, r0.start, r0.end, r1.start, r1.end)
  p0[0] = zp.f[1], p1[0] = zp.f[2], p0[1] = zp.f[3], p1[1] = zp.f[4], p1[2] = zp.f[0], p0[2] = yp.f[3];
  auto same = [](float a, float c) { return fbits(a) == fbits(c); };
  for (int i = 0; i < 6; i++) {
    const HostObject& r = obj(face[i]);
    const int axis = 2 - (i % 3), a1 = axis == 0 ? 1 : 0, a2 = axis == 2 ? 1 : 2;
    const float k = i < 3 ? p1[axis] : p0[axis];
    if (!same(r.f[0], k) || !same(r.f[1], p0[a1]) || !same(r.f[2], p1[a1]) || !same(r.f[3], p0[a2]) || !same(r.f[4], p1[a2]))
      return false;
  }
  *mat = zp.mat;
  return true;
}

bool SceneBuilder::holds_medium(uint32_t id) const {
  const HostObject& o = objects[id];
  switch (o.kind) {
    case HostObject::MEDIUM: return true;
    case HostObject::SPHERE:
    case HostObject::RECT: return false;
    case HostObject::AND: return holds_medium(o.a) || holds_medium(o.b);
    case HostObject::BVH: {
      std::vector<int32_t> todo(1, (int32_t)o.a);
      while (!todo.empty()) {
        const HostBvhNode& n = bvh_nodes[todo.back()];
        todo.pop_back();
        if (n.leaf != kNone) {
          if (holds_medium(n.leaf)) return true;
        } else {
          todo.push_back(n.left), todo.push_back(n.right);
        }
      }
      return false;
    }
    default: return holds_medium(o.a);  // wrappers
  }
}

void SceneBuilder::emit_bvh(int32_t node_id, int depth, FlatScene* out, int boundary, bool mark_roots, int saves) const {
  const HostBvhNode& n = bvh_nodes[node_id];
  size_t at = out->lo.size();
  push(out, n.box.mn[0], n.box.mx[0], n.box.mn[1], n.box.mx[1], fbits(n.box.mn[2]), fbits(n.box.mx[2]), 0, OP_BOX);
  if (n.leaf != kNone) {
    emit(n.leaf, true, depth, out, boundary, mark_roots, saves);
  } else {
    emit_bvh(n.left, depth, out, boundary, mark_roots, saves);
    emit_bvh(n.right, depth, out, boundary, mark_roots, saves);
  }
  out->hi[at].w[2] = (uint32_t)out->lo.size();  // skip pointer: first instruction after the subtree
}

void SceneBuilder::emit(uint32_t id, bool under_bvh, int depth, FlatScene* out, int boundary, bool mark_roots, int saves) const {
  const HostObject& o = objects[id];
  if (fuse_primitive(*this, id, out, true, true)) return;
  switch (o.kind) {
    case HostObject::AND: {
      float p0[3], p1[3];
      uint32_t mat;
      if (match_prism(*this, id, p0, p1, &mat)) {
        push(out, p0[0], p1[0], p0[1], p1[1], fbits(p0[2]), fbits(p1[2]), mat, OP_PRISM | mat_flags(*this, mat));
        out->features |= FEAT_RECT;
        return;
      }
      if (under_bvh && holds_medium(id)) {
        // `And` below a Bvh with a ConstantMedium inside.  Inside the And the later hit replaces the earlier one whatever
        // its t (object.rs:403-409); the Bvh node above compares (bvh.rs:104-112), and a medium may return t >= t_range.end.
        // The flat stream says so with a SAVE / MERGE pair around the And's stream, whose own objects are emitted as a
        // list's (media replace; Bvhs inside are outermost ones again).  Only the general walk executes these two records.
        if (saves >= MAX_SAVE_NESTING)
          throw BuildError{-5, "And-with-medium leaves of Bvhs nested deeper than the general walk's save stack (4)"};
        out->features |= FEAT_DEEP;
        push(out, 0, 0, 0, 0, 0, 0, 0, OP_SAVE);
        emit(o.a, false, depth, out, boundary, true, saves + 1);
        emit(o.b, false, depth, out, boundary, true, saves + 1);
        push(out, 0, 0, 0, 0, 0, 0, 0, OP_MERGE);
        return;
      }
    }
      emit(o.a, under_bvh, depth, out, boundary, mark_roots, saves);
      emit(o.b, under_bvh, depth, out, boundary, mark_roots, saves);
      return;
    case HostObject::BVH: {
      size_t root = out->lo.size();
      emit_bvh((int32_t)o.a, depth, out, boundary, mark_roots, saves);
      // (a boundary query of the scheduled kernels keeps no hit count: no merge rule there, roots stay unmarked)
      if (!under_bvh && mark_roots) out->hi[root].w[3] |= F_BVH_ROOT;
      return;
    }
    case HostObject::MEDIUM: {
      if (boundary > 0) {
        // a ConstantMedium inside another medium's boundary (`ConstantMedium<O: Object>`, object.rs:533-575, with O holding a
        // medium): the general walk follows MAX_MEDIUM_NESTING levels of boundary queries
        if (boundary >= MAX_MEDIUM_NESTING)
          throw BuildError{-5, "media nested in media boundaries deeper than the general walk follows (3)"};
        out->features |= FEAT_DEEP;
      }
      {
        const size_t at = out->lo.size();
        const bool single = fuse_primitive(*this, o.a, out, false, false);  // (a moving boundary is an object graph: boundary_pair_t knows no time)
        push(out, o.f[0], 1.f / o.f[0], 0, 0, 0, 0, o.mat,  // (1 / density: object.rs:562's divide, done once -- the same correctly rounded f32 quotient)
             OP_MEDIUM | (under_bvh ? F_UNDER_BVH : 0u) | (single ? 0u : F_GENERAL_BOUNDARY) | mat_flags(*this, o.mat));
        // the boundary's own stream: evaluated twice per medium test by a nested walk (object.rs:551-552),
        // skipped by the main walk.  It starts a fresh wrapper depth (its rays are saved on a private stack).
        if (single) {
          fuse_primitive(*this, o.a, out, true, false);
        } else {
          out->deep_media = std::max(out->deep_media, boundary + 1);
          emit(o.a, false, 0, out, boundary + 1, holds_medium(o.a), 0), out->features |= FEAT_BOUNDARY;
          push(out, 0, 0, 0, 0, 0, 0, (uint32_t)at, OP_BEND);  // where a range-query walk of the stream finishes
        }
        out->hi[at].w[0] = (uint32_t)out->lo.size();  // end_pc
      }
      out->features |= FEAT_MEDIUM;
      return;
    }
    default: break;
  }
  // generic wrapper: PUSH, subtree, POP
  uint32_t kind;
  switch (o.kind) {
    case HostObject::TRANSLATE: kind = XF_TRANSLATE; break;
    case HostObject::ROTATE_Y: kind = XF_ROTATE_Y; break;
    case HostObject::SCALE: kind = XF_SCALE; break;
    case HostObject::MOVE: kind = XF_MOVE; break;
    case HostObject::FLIP: kind = XF_FLIP; break;
    default: throw BuildError{-1, "flatten: unknown object kind"};
  }
  if (depth >= MAX_DEEP_XFORM_DEPTH)
    throw BuildError{-5, "transform wrappers nested deeper than the general walk's ray stack (32)"};
  out->deep_wrappers = std::max(out->deep_wrappers, depth + 1);
  if (depth >= MAX_XFORM_DEPTH) out->features |= FEAT_DEEP;  // deeper than the scheduled kernels' ray stack: the general walk
  out->features |= FEAT_XFORM;
  // Translate{RotateY{x}} / Translate{LinearMove{x}} (every transformed object of main.rs): one wrapper level
  const HostObject* w = &o;
  float pre[3] = {0, 0, 0};
  uint32_t pre_flag = 0;
  if (o.kind == HostObject::TRANSLATE && (objects[o.a].kind == HostObject::ROTATE_Y || objects[o.a].kind == HostObject::MOVE)) {
    pre[0] = o.f[0], pre[1] = o.f[1], pre[2] = o.f[2];
    pre_flag = F_PRE_TRANSLATE;
    w = &objects[o.a];
    kind = w->kind == HostObject::ROTATE_Y ? XF_ROTATE_Y : XF_MOVE;
  }
  size_t at = out->lo.size();
  push(out, w->f[0], w->f[1], w->f[2], pre[0], fbits(pre[1]), fbits(pre[2]), 0, OP_PUSH | (kind << F_KIND_SHIFT) | pre_flag);
  emit(w->a, under_bvh, depth + 1, out, boundary, mark_roots, saves);
  out->hi[at].w[2] = (uint32_t)out->lo.size();
  push(out, w->f[0], w->f[1], w->f[2], pre[0], fbits(pre[1]), fbits(pre[2]), (uint32_t)at, OP_POP | (kind << F_KIND_SHIFT) | pre_flag);
}

// Is world object `id` ONE plain primitive record -- a (fused) SPHERE, a RECT or a rect_prism?
static bool is_plain_primitive(const SceneBuilder& b, uint32_t id) {
  if (fuse_primitive(b, id, nullptr, false, true)) return true;
  float p0[3], p1[3];
  uint32_t mat;
  return match_prism(b, id, p0, p1, &mat);
}

void SceneBuilder::flatten(const uint32_t* world, size_t n, FlatScene* out) const {
  for (size_t i = 0; i < n; i++)
    if (world[i] >= objects.size()) throw BuildError{-1, "scene: bad world object handle"};
  // The hoisted segment (flat_scene.h OP_SEG): the longest run of >= 2 consecutive top-level plain primitives.  Only programs
  // the full-feature POOL kernel renders get one (a Bvh somewhere, not a lean BOX / SPHERE program, no FEAT_DEEP shape): the
  // first flattening finds that out, the second one emits the record.
  size_t seg0 = 0, seg1 = 0;
  for (size_t i = 0; i < n;) {
    size_t j = i;
    while (j < n && is_plain_primitive(*this, world[j])) j++;
    if (j - i >= 2 && j - i > seg1 - seg0) seg0 = i, seg1 = j;
    i = j > i ? j : i + 1;
  }
  flatten_program(world, n, out, 0, 0);
  bool has_box = false;
  for (const Packet& h : out->hi) has_box |= (h.w[3] & 0xffu) == OP_BOX;
  const bool full_pool = has_box && (out->features & (FEAT_XFORM | FEAT_MEDIUM | FEAT_RECT | FEAT_TEXTURE | FEAT_BOUNDARY)) != 0u &&
                         !(out->features & FEAT_DEEP);
  if (hoist_segments && full_pool && seg1 > seg0) flatten_program(world, n, out, seg0, seg1);
  if (full_pool) (void)flatten_pool2(world, n, out);  // the second program, when the world has the shape for it
  if (out->features & FEAT_DEEP) {  // which instantiation of the general walk this graph needs (flat_scene.h)
    if (out->deep_wrappers <= DEEP_FEW_WRAPPERS) out->features |= FEAT_DEEP_FEW_WRAPPERS;
    if (out->deep_media <= 1) out->features |= FEAT_DEEP_ONE_LEVEL;
  }
  finish_materials(out);
}

// world objects [seg0, seg1) (seg1 > seg0) are preceded by an OP_SEG record that skips them
void SceneBuilder::flatten_program(const uint32_t* world, size_t n, FlatScene* out, size_t seg0, size_t seg1) const {
  out->lo.clear(), out->hi.clear(), out->mat.clear(), out->tex.clear();
  out->features = 0;
  out->deep_wrappers = 0, out->deep_media = 0;
  // Runs of consecutive list-level objects that hold no Bvh are straight-line code every ray executes in
  // the same order: their first record is marked F_GATHER so a scheduler can batch the rays there.
  size_t run_start = 0;
  bool in_run = false;
  auto close_run = [&](size_t end) {
    // (the FULL run counts, the records of a hoisted segment included: the lock-step kernel, the baseline kernel and the pool kernel
    // with hoisting switched off execute them)
    if (in_run && end - run_start >= 4) out->hi[run_start].w[3] |= F_GATHER;
    in_run = false;
  };
  size_t seg_at = 0;
  for (size_t i = 0; i < n; i++) {
    const size_t at = out->lo.size();
    if (seg1 > seg0 && i == seg0) seg_at = at, push(out, 0, 0, 0, 0, 0, 0, 0, OP_SEG);
    emit(world[i], false, 0, out);
    if (seg1 > seg0 && i + 1 == seg1) out->hi[seg_at].w[2] = (uint32_t)out->lo.size();
    bool has_box = false;
    for (size_t r = at; r < out->lo.size(); r++) has_box |= (out->hi[r].w[3] & 0xffu) == OP_BOX;
    if (has_box) {
      close_run(at);
    } else if (!in_run) {
      in_run = true, run_start = at;
    }
  }
  close_run(out->lo.size());
  push(out, 0, 0, 0, 0, 0, 0, 0, OP_END);
}

// The second program (flat_scene.h "the list level, hoisted"; rt_pool2.h).  Every top-level object must be one of
//   P  a plain primitive (one fused SPHERE / RECT / PRISM record),
//   M  a ConstantMedium over one plain primitive (not moving),
//   B  a Bvh whose leaves are all plain primitives,
//   W  one wrapper level (what emit() turns into one PUSH / POP pair) around a B;
// at most P2_MAX_ITEMS items (runs of P, M, W) and P2_MAX_MEDIA media, and at least one Bvh.  Layout: the WALK -- Bvh streams
// (emit()'s own records), an OP_LIST record wherever list-level items stand between them, W as PUSH / stream / POP, OP_END -- then,
// behind OP_END, the records of the P and M items for the evaluation at ray creation and for rebuild_hit.  Returns false (and
// leaves lo2 / hi2 empty) for any other shape: the first program and the older kernels render it.
bool SceneBuilder::flatten_pool2(const uint32_t* world, size_t n, FlatScene* out) const {
  out->lo2.clear(), out->hi2.clear();
  out->p2 = P2Table{};
  auto bvh_of_plain = [&](uint32_t id) {
    if (objects[id].kind != HostObject::BVH) return false;
    std::vector<int32_t> todo(1, (int32_t)objects[id].a);
    while (!todo.empty()) {
      const HostBvhNode& nd = bvh_nodes[todo.back()];
      todo.pop_back();
      if (nd.leaf != kNone) {
        if (!is_plain_primitive(*this, nd.leaf)) return false;
      } else {
        todo.push_back(nd.left), todo.push_back(nd.right);
      }
    }
    return true;
  };
  auto classify = [&](uint32_t id) -> char {
    const HostObject& o = objects[id];
    if (is_plain_primitive(*this, id)) return 'P';
    if (o.kind == HostObject::MEDIUM) return fuse_primitive(*this, o.a, nullptr, false, false) ? 'M' : 0;
    if (bvh_of_plain(id)) return 'B';
    if (o.kind == HostObject::TRANSLATE || o.kind == HostObject::ROTATE_Y || o.kind == HostObject::SCALE || o.kind == HostObject::MOVE ||
        o.kind == HostObject::FLIP) {  // emit()'s generic wrapper: Translate{RotateY | LinearMove} is ONE level
      const HostObject* w = &o;
      if (o.kind == HostObject::TRANSLATE && (objects[o.a].kind == HostObject::ROTATE_Y || objects[o.a].kind == HostObject::MOVE)) w = &objects[o.a];
      return bvh_of_plain(w->a) ? 'W' : 0;
    }
    return 0;
  };
  std::vector<char> cls(n);
  bool any_bvh = false;
  for (size_t i = 0; i < n; i++) {
    cls[i] = classify(world[i]);
    if (!cls[i]) return false;
    any_bvh |= cls[i] == 'B' || cls[i] == 'W';
  }
  if (!any_bvh) return false;
  FlatScene w;  // the walk, then the item records
  P2Table t{};
  struct Pending { size_t first_obj, end_obj; };  // P / M items wait for their records (emitted behind OP_END)
  std::vector<std::pair<uint32_t, Pending>> pending;  // (item index, world objects)
  auto new_item = [&](uint32_t kind) -> int {
    if (t.n_items >= P2_MAX_ITEMS) return -1;
    t.item[t.n_items].kind = kind;
    return (int)t.n_items++;
  };
  uint32_t group_first = 0;  // first item of the list-level run that is being collected
  auto close_group = [&]() {
    if (t.n_items > group_first) push(&w, 0, 0, 0, 0, group_first, t.n_items - group_first, 0, OP_LIST);
    group_first = t.n_items;
  };
  for (size_t i = 0; i < n;) {
    if (cls[i] == 'P') {
      size_t j = i;
      while (j < n && cls[j] == 'P') j++;
      const int it = new_item(P2_PRIMS);
      if (it < 0) return false;
      pending.push_back({(uint32_t)it, Pending{i, j}});
      i = j;
    } else if (cls[i] == 'M') {
      const int it = new_item(P2_MEDIUM);
      if (it < 0 || t.n_media >= P2_MAX_MEDIA) return false;
      t.n_media++;
      pending.push_back({(uint32_t)it, Pending{i, i + 1}});
      i++;
    } else if (cls[i] == 'B') {
      close_group();
      emit(world[i], false, 0, &w);
      i++;
    } else {  // W: its root test is the last item of the run in front of it
      const int it = new_item(P2_WRAPPED);
      if (it < 0) return false;
      close_group();
      const size_t at = w.lo.size();
      emit(world[i], false, 0, &w);
      if ((w.hi[at].w[3] & 0xffu) != OP_PUSH || (w.hi[at + 1].w[3] & 0xffu) != OP_BOX) return false;  // (cannot happen for a W)
      t.item[it].a = (uint32_t)at, t.item[it].b = (uint32_t)at + 1u, t.item[it].c = (uint32_t)w.lo.size();
      t.n_wrapped++;
      i++;
    }
  }
  close_group();
  // a POP behind which the walk meets nothing but other POPs need not restore the ray
  {
    bool ray_needed = false;
    for (size_t r = w.lo.size(); r-- > 0;) {
      const uint32_t op = w.hi[r].w[3] & 0xffu;
      if (op == OP_POP && !ray_needed) w.hi[r].w[3] |= F_P2_DEAD_POP;
      if (op != OP_POP) ray_needed = true;  // (an OP_LIST record too: a medium's commit takes |d|, a NaN candidate replays its run on the ray)
    }
  }
  push(&w, 0, 0, 0, 0, 0, 0, 0, OP_END);
  for (auto& pi : pending) {
    P2Item& it = t.item[pi.first];
    it.a = (uint32_t)w.lo.size();
    for (size_t k = pi.second.first_obj; k < pi.second.end_obj; k++) emit(world[k], false, 0, &w);
    it.b = (uint32_t)w.lo.size();
  }
  if (w.features & (FEAT_DEEP | FEAT_BOUNDARY)) return false;
  out->lo2 = std::move(w.lo), out->hi2 = std::move(w.hi);
  out->p2 = t;
  return true;
}

void SceneBuilder::finish_materials(FlatScene* out) const {
  // Albedo range of every scattering material -- decides whether the pool kernels' "accum is +0" argument holds
  // (rt_pool.h PoolField): walks the WHOLE texture tree (a checker's children, recursively; texture.rs:12-21), any
  // reachable Perlin texture is "bright" (turb <= 2 sqrt(3) x the longest table vector, perlin.rs:31-75), and a Perlin
  // table with vectors longer than unit length (or NaN) has no bound at all.
  float perlin_len2 = 0.f;
  bool perlin_bad = false;
  if (has_perlin)
    for (int i = 0; i < 256; i++) {
      const float* v = &perlin_vecs[4 * i];
      const float l2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
      if (!(l2 <= 1.0002f)) perlin_bad = true;
      if (l2 > perlin_len2) perlin_len2 = l2;
    }
  auto range_of_constant = [&](const float c[3]) {
    for (int i = 0; i < 3; i++) {
      if (!(c[i] >= 0.f && c[i] <= 4.f)) out->features |= FEAT_WIDE_ALBEDO;
      else if (c[i] > 1.f) out->features |= FEAT_BRIGHT_ALBEDO;
    }
  };
  std::vector<uint32_t> todo;
  auto range_of_texture = [&](uint32_t root) {
    todo.assign(1, root);
    while (!todo.empty()) {
      const uint32_t id = todo.back();
      todo.pop_back();
      const HostTexture& t = textures[id];
      if (t.kind == TEX_CONSTANT) range_of_constant(t.rgb);
      else if (t.kind == TEX_PERLIN) out->features |= perlin_bad ? FEAT_WIDE_ALBEDO : FEAT_BRIGHT_ALBEDO;
      else todo.push_back(t.t0), todo.push_back(t.t1);  // checker: children were created earlier (ids < id): no cycles
    }
  };
  for (const HostMaterial& m : materials) {
    float c[3] = {m.albedo[0], m.albedo[1], m.albedo[2]};
    uint32_t texkind = TEX_CONSTANT, tex = 0;
    const bool scatters_with_albedo = m.kind != MAT_DIFFUSE_LIGHT && m.kind != MAT_DIELECTRIC;
    if (m.tex != kNone) {
      const HostTexture& t = textures[m.tex];
      texkind = t.kind;
      tex = m.tex;
      if (t.kind == TEX_CONSTANT) c[0] = t.rgb[0], c[1] = t.rgb[1], c[2] = t.rgb[2];
      else out->features |= FEAT_TEXTURE;
      if (scatters_with_albedo) range_of_texture(m.tex);
    } else if (scatters_with_albedo) {
      range_of_constant(c);
    }
    out->mat.push_back(Packet{{fbits(c[0]), fbits(c[1]), fbits(c[2]), fbits(m.param)}});
    out->mat.push_back(Packet{{tex, 0, 0, m.kind | (texkind << 8)}});
  }
  for (const HostTexture& t : textures) {
    out->tex.push_back(Packet{{fbits(t.rgb[0]), fbits(t.rgb[1]), fbits(t.rgb[2]), fbits(t.scale)}});
    out->tex.push_back(Packet{{t.t0, t.t1, 0, t.kind}});
  }
  out->perlin_vecs = perlin_vecs;
  out->perlin_perm = perlin_perm;
  out->has_perlin = has_perlin;
}

}  // namespace rtg
