// flat_scene.h -- the HBM layout of a flattened scene ("flat program").
//
// The reference walks a recursive `Box<dyn Object>` graph (object.rs:15-40) whose traversal ORDER is
// observable: hits use strict `<` (object.rs:99,195), the right BVH child is searched with
// t_max = left.t (bvh.rs:98-102), `And` likewise (object.rs:403-409) and ConstantMedium draws from the
// RNG during traversal (object.rs:562).  Because that order is always "left subtree, then right
// subtree" -- never ray dependent -- the whole graph linearises into ONE instruction stream in
// depth-first order with skip pointers: no traversal stack, exactly the reference's visiting order.
//
//   list world  (lib.rs:40-45)      -> its objects' streams back to back
//   And(a, b)   (object.rs:396-410) -> a's stream, then b's stream
//   Bvh node    (bvh.rs:84-120)     -> BOX{aabb, skip}, left stream, right stream   (skip = first
//   Bvh leaf                        -> BOX{aabb, skip}, object's stream              instr. after)
//   Translate/Scale/RotateY/LinearMove/FlipNormals over a subtree -> PUSH, subtree, POP
//   Sphere, Translate{Sphere}, FlipNormals thereof -> one fused SPHERE record
//   LinearMove{Sphere}, Translate{LinearMove{Sphere}} (main.rs:218-229), FlipNormals thereof -> one fused SPHERE record with
//                                                     F_MOVE + one data record (OP_EXT: the motion vector) behind it
//   Rect<A>, FlipNormals(Rect<A>)                  -> one RECT record
//   rect_prism(p0, p1, m) (the exact And-tree of object.rs:420-473) -> one PRISM record
//   Translate{RotateY{x}}, Translate{LinearMove{x}} -> ONE PUSH/POP pair (F_PRE_TRANSLATE): same arithmetic, in sequence
//   ConstantMedium{boundary}                       -> MEDIUM{end} followed by the boundary's own stream (one fused
//                                                     primitive record in every reference scene; any object graph without
//                                                     a nested medium otherwise), skipped by the main walk
//
// Every instruction is 32 bytes = two 16-byte packets, stored as two SoA arrays of uint4
// (`lo[i]`, `hi[i]`) so that a lane fetches an instruction with two 16-byte loads.
#pragma once
#include <stdint.h>

namespace rtg {

enum Op : uint32_t {
  OP_END = 0,
  OP_BOX = 1,     // lo = (min.x, max.x, min.y, max.y)  hi = (min.z, max.z, skip_pc, op): per-axis (min,max) pairs sit in
                  //      adjacent registers so the slab test runs as packed f32 math (v_pk_add_f32 / v_pk_mul_f32)
  OP_SPHERE = 2,  // lo = (off.x, off.y, off.z, radius) hi = (-, -, material, op|flags)
  OP_RECT = 3,    // lo = (k, r0.start, r0.end, r1.start) hi = (r1.end, -, material, op|flags)
  OP_PUSH = 4,    // lo = (a, b, c, -)                   hi = (-, -, matching_pop, op|kind)
  OP_POP = 5,     // lo = (a, b, c, -)                   hi = (-, -, matching_push, op|kind)
  OP_MEDIUM = 6,  // lo = (density, 1 / density, -, -)   hi = (end_pc, -, material, op|flags); boundary = records (pc, end_pc)
  OP_PRISM = 7,   // lo = (p0.x, p1.x, p0.y, p1.y)       hi = (p0.z, p1.z, material, op|flags): rect_prism(p0, p1, material)
                  //      (object.rs:420-473), its six Rect::hit in the And-tree's order inside ONE instruction
  OP_BEND = 8,    // hi = (-, -, medium_pc, op): end of the record stream of a MEDIUM whose boundary is an object graph
                  //      (F_GENERAL_BOUNDARY).  The main walk never reaches it (MEDIUM.end_pc points behind it); a walk that
                  //      runs the boundary's stream as a range query (rt_pool_full.h GENB) finishes the query here.
  OP_SEG = 9,     // hi = (-, -, skip_pc, op): in front of the hoisted SEGMENT of a list world -- a run of consecutive top-level plain
                  //      primitives (SPHERE incl. F_MOVE + its OP_EXT, RECT, PRISM), the records up to skip_pc.  What such a primitive returns
                  //      for a ray does not depend on the hits found so far except through `t < t_range.end` (object.rs:99,195), and a
                  //      later hit needs a strictly smaller t (lib.rs:41-44), so the run's result is "its closest candidate, first wins,
                  //      if closer than what was found before it".  A kernel that evaluates the candidate when the ray is CREATED
                  //      (rt_pool_full.h hoist_eval: full-width, every lane on the same record) commits it here and continues at
                  //      skip_pc; every other interpreter steps over this record and executes the run.  Same tests, same closest hit.
  // Only in programs with FEAT_DEEP (graph shapes the scheduled kernels do not walk; the general walk of rt_trace.h does):
  OP_SAVE = 10,   // in front of the stream of an `And` that sits below a Bvh and holds a ConstantMedium: remember the hit so far
  OP_MERGE = 11,  // behind it: bvh.rs:104-112 for that leaf -- the earlier hit `hl` wins when hl.t < hr.t (only a medium can
                  // return t >= t_range.end; inside the And itself the later hit replaces, object.rs:403-409)
  OP_EXT = 12,    // data-only continuation of the record in front of it (a SPHERE with F_MOVE: lo = motion.xyz); never executed:
                  // the record that owns it steps over it
  // Only in the SECOND program of a scene, the one the pool-2 kernel walks (rt_pool2.h; P2Table below):
  OP_LIST = 13,   // hi = (first item, item count, -, op): commits the list-level items [first, first + count) of the P2Table, in
                  //      order, from what was evaluated for them when the ray was created.  A W item is the last of its record.
};

// ---- the list level, hoisted (pool-2 kernel, rt_pool2.h) -------------------------------------------------------------------
// A list world (lib.rs:33-49) tests EVERY top-level object for every ray, in order, against a shrinking t_range.end.  What a
// top-level object contributes depends on the hits found before it only in a narrow way:
//   a plain primitive (Sphere / Rect / rect_prism, fused wrappers included): through `t < t_range.end` (object.rs:99,195) -- a run of
//     consecutive ones reduces to "its closest candidate, first wins, if closer than what was found before" (OP_SEG above);
//   a ConstantMedium over ONE primitive: its two boundary queries run over f32::MIN..f32::MAX (object.rs:551-552), independent of
//     the range; the range enters in `max(t1, start)`, `min(t2, end)` (:553-554) and decides whether the medium draws (:562);
//   a wrapper around a Bvh (Translate{RotateY{Bvh}}, main.rs:295-316): whether the walk enters it is Aabb::hit of the Bvh's root
//     box for the transformed ray -- start and far plane distances independent of the range, then `min(end, far) > start`.
// So everything expensive at list level is evaluated when the ray is CREATED (shade / camera pass: 60 lanes wide, every lane on the
// same record) into a per-path side record of two dwords per item, and the walk COMMITS a list-level run with one cheap record
// (OP_LIST) that needs no company: the walk of the second program is Bvh streams and OP_LIST records only.  Items, in world order:
enum P2Kind : uint32_t {
  P2_NONE = 0,
  P2_PRIMS = 1,   // a = first record, b = end record: a run of plain primitives (records behind the program's OP_END).  Side: (t, pc | face | textured)
  P2_MEDIUM = 2,  // a = the MEDIUM record (its boundary primitive = a + 1).  Side: (t1, t2) of the two boundary queries, t1 = +inf: not crossed
  P2_WRAPPED = 3, // a = the PUSH record, b = the wrapped Bvh's root BOX (= a + 1), c = first record behind the matching POP.  Side: (start, far)
};
struct P2Item {
  uint32_t kind, a, b, c;
};
constexpr uint32_t P2_MAX_ITEMS = 5;  // side record = 2 x 5 dwords + ln of the first two draws of the event's stream (media, object.rs:562)
constexpr uint32_t P2_MAX_MEDIA = 2;
constexpr uint32_t F_P2_DEAD_POP = 1u << 12;  // POP (second program): only POPs and OP_END follow in the walk -- the ray need not be restored
struct P2Table {
  P2Item item[P2_MAX_ITEMS];
  uint32_t n_items, n_media, n_wrapped, pad;
};

// flag bits in hi.w above the 8-bit opcode
constexpr uint32_t F_TRANSLATE = 1u << 8;   // SPHERE: origin -= off on the way in, p += off on the way out
constexpr uint32_t F_FLIP = 1u << 9;        // SPHERE/RECT: normal = -normal (odd number of FlipNormals)
constexpr uint32_t F_AXIS_SHIFT = 10;       // RECT: bits 10-11 = orthogonal axis (0/1/2)
constexpr uint32_t F_UNDER_BVH = 1u << 12;  // MEDIUM: lives below a Bvh node (hit-merge rule of bvh.rs:104-112)
constexpr uint32_t F_BVH_ROOT = 1u << 13;   // BOX: root of an outermost Bvh (a Bvh not nested below another Bvh)
constexpr uint32_t F_GENERAL_BOUNDARY = 1u << 14;  // MEDIUM: the boundary is an object graph (several records), not one primitive
constexpr uint32_t F_GATHER = 1u << 15;     // any non-BOX record: head of a run of >= 4 list-level records without a Bvh (scheduling
                                            // hint: every ray passes here and then executes the same records in the same order)
constexpr uint32_t F_MATKIND_SHIFT = 16;    // SPHERE/RECT/MEDIUM: bits 16-18 = MatKind of the record's material (copy, for schedulers)
constexpr uint32_t F_TEXTURED = 1u << 19;   // SPHERE/RECT/PRISM/MEDIUM: the material reads a checker / Perlin texture (copy, for schedulers)
constexpr uint32_t F_MOVE = 1u << 20;      // SPHERE: a LinearMove sits between the (optional, F_TRANSLATE) Translate and the sphere: after the
                                            // translate, origin -= time * motion (object.rs:505-508; nothing on the way out); motion = lo.xyz of the
                                            // OP_EXT record behind this one, which the SPHERE steps over (pc += 2)
constexpr uint32_t F_KIND_SHIFT = 8;        // PUSH/POP: bits 8-10 = XformKind
constexpr uint32_t F_PRE_TRANSLATE = 1u << 11;  // PUSH/POP (RotateY / LinearMove): an enclosing Translate rides along,
                                                // offset = (lo.w, hi.x, hi.y): applied first on the way in, last on the way out

enum XformKind : uint32_t {
  XF_TRANSLATE = 0,  // (a,b,c) = offset               object.rs:267-283
  XF_ROTATE_Y = 1,   // (a,b)   = (sin_theta, cos_theta) object.rs:341-370
  XF_SCALE = 2,      // (a,b,c) = factor               object.rs:301-319
  XF_MOVE = 3,       // (a,b,c) = motion               object.rs:496-512
  XF_FLIP = 4,       //                                  object.rs:241-253
};

constexpr int MAX_XFORM_DEPTH = 4;  // PUSH nesting the scheduled kernels' ray stack holds (a medium's boundary stream starts a fresh count)
constexpr int MAX_DEEP_XFORM_DEPTH = 32;  // ... and the general walk's (FEAT_DEEP)
constexpr int MAX_MEDIUM_NESTING = 3;     // media inside the boundary of a medium inside ...: levels the general walk follows
constexpr int MAX_SAVE_NESTING = 4;       // OP_SAVE .. OP_MERGE pairs inside one another

// Material record, 32 bytes (two uint4): lo = (c.r, c.g, c.b, param) hi = (texture, -, -, kind|texkind<<8)
//   Lambertian / Isotropic: c = albedo when the texture is constant, else `texture` indexes tex[]
//   Metal: c = albedo, param = fuzz;  Dielectric: param = ref_idx
//   DiffuseLight: c = emission when constant, param = brightness
enum MatKind : uint32_t { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
enum TexKind : uint32_t { TEX_CONSTANT = 0, TEX_CHECKER = 1, TEX_PERLIN = 2 };
// Texture record, 32 bytes: lo = (r, g, b, scale) hi = (t0, t1, -, kind)

// Feature bits of a flattened scene: select the kernel instantiation (lean book-1 path vs full path)
constexpr uint32_t FEAT_XFORM = 1u;    // PUSH/POP present
constexpr uint32_t FEAT_MEDIUM = 2u;   // MEDIUM present
constexpr uint32_t FEAT_RECT = 4u;     // RECT present
constexpr uint32_t FEAT_TEXTURE = 8u;  // a non-constant texture is referenced
constexpr uint32_t FEAT_BOUNDARY = 16u; // a ConstantMedium whose boundary is an object graph (nested boundary walk)
constexpr uint32_t FEAT_DEEP = 128u;   // a graph shape only the general walk handles (rt_trace.h walk_deep, baseline kernel): more than MAX_XFORM_DEPTH
                                       // nested wrappers, a medium inside a medium's boundary, a medium below an And below a Bvh
// Only together with FEAT_DEEP, chosen per scene from the depths the flattener measured (FlatScene::deep_wrappers / deep_media):
// which instantiation of the general walk renders it.  The walk's private stacks are sized by its template parameters --
// 32 wrapper levels x 4 nesting levels of media is 4.2 KB of scratch per lane and 178 VGPRs whatever the graph needs.
constexpr uint32_t FEAT_DEEP_FEW_WRAPPERS = 256u;  // no more than DEEP_FEW_WRAPPERS wrappers are ever open at once: ray stacks of 8 instead of 32
constexpr uint32_t FEAT_DEEP_ONE_LEVEL = 512u;     // no medium inside a medium's boundary: the walk recurses one level (its boundary queries), not three
constexpr int DEEP_FEW_WRAPPERS = 8;
constexpr uint32_t FEAT_BRIGHT_ALBEDO = 64u; // an albedo component may exceed 1 (a constant in (1, 4], or Perlin turbulence, <= 3.47): the pool
                                            // kernels then need max_bounces <= 63 for the strength to stay finite
constexpr uint32_t FEAT_WIDE_ALBEDO = 32u;  // an albedo component outside [0, 4]: path strength may overflow or change sign, so the pool
                                            // kernels' "accum is +0" does not hold (rt_pool.h PoolField): baseline kernel

}  // namespace rtg
