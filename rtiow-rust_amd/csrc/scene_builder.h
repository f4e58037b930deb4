// scene_builder.h -- host side of the boundary: the builder behind the rtg_object_* / rtg_material_* /
// rtg_texture_* calls, Bvh::new (bvh.rs:22-81), bounding boxes (object.rs) and the flattener that
// turns the object graph into the flat program of flat_scene.h.  Pure host C++ (no HIP).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "flat_scene.h"

namespace rtg {

struct Box3 {
  float mn[3], mx[3];
};

struct Packet {
  uint32_t w[4];
};

struct HostObject {
  enum Kind : uint8_t { SPHERE, RECT, FLIP, TRANSLATE, SCALE, ROTATE_Y, AND, MOVE, MEDIUM, BVH } kind;
  float f[6] = {0, 0, 0, 0, 0, 0};  // kind-specific scalars (see scene_builder.cpp)
  uint32_t a = 0xffffffffu, b = 0xffffffffu;  // children (object ids) / bvh root node id
  uint32_t mat = 0xffffffffu;
  int axis = 0;
};

struct HostBvhNode {  // bvh.rs:9-19
  Box3 box;
  int32_t left = -1, right = -1;  // BvhContents::Node
  uint32_t leaf = 0xffffffffu;    // BvhContents::Leaf (object id)
};

struct HostMaterial {
  uint32_t kind;
  uint32_t tex = 0xffffffffu;
  float albedo[3] = {0, 0, 0};
  float param = 0.f;  // fuzz / ref_idx / brightness
};

struct HostTexture {
  uint32_t kind;
  float rgb[3] = {0, 0, 0};
  float scale = 0.f;
  uint32_t t0 = 0, t1 = 0;
};

struct FlatScene {
  std::vector<Packet> lo, hi;    // program
  std::vector<Packet> mat, tex;  // 2 packets per record
  std::array<float, 1024> perlin_vecs{};  // 256 x float4
  std::array<uint8_t, 768> perlin_perm{};
  uint32_t features = 0;
  bool has_perlin = false;
  int deep_wrappers = 0;  // most wrappers (PUSH) open at once in any one stream of the program
  int deep_media = 0;     // deepest level of boundary queries a walk of the program recurses into (0 = no object-graph boundary)
  // the SECOND program (flat_scene.h "the list level, hoisted"): empty unless the world has the shape the pool-2 kernel walks
  std::vector<Packet> lo2, hi2;
  P2Table p2{};
};

class SceneBuilder {
 public:
  std::vector<HostObject> objects;
  std::vector<HostBvhNode> bvh_nodes;
  std::vector<HostMaterial> materials;
  std::vector<HostTexture> textures;
  std::array<float, 1024> perlin_vecs{};
  std::array<uint8_t, 768> perlin_perm{};
  bool has_perlin = false;

  // All return an id or throw BuildError.
  uint32_t add_object(const HostObject& o);
  uint32_t add_bvh(const uint32_t* objs, size_t n, float e0, float e1, bool sah = false);
  Box3 bounding_box(uint32_t obj, float e0, float e1) const;
  void flatten(const uint32_t* world, size_t n, FlatScene* out) const;
  bool hoist_segments = true;  // flatten() may put an OP_SEG record in front of a list world's longest run of plain primitives

 private:
  int32_t build_bvh(std::vector<uint32_t> objs, float e0, float e1);
  int32_t build_bvh_sah(std::vector<uint32_t> objs, float e0, float e1);
  // `boundary` = nesting level of medium boundaries the stream belongs to (0 = the main walk); `mark_roots`: outermost Bvhs of
  // this stream get F_BVH_ROOT (always in the main walk; in a boundary stream only when it holds a medium itself)
  void emit(uint32_t obj, bool under_bvh, int depth, FlatScene* out, int boundary = 0, bool mark_roots = true, int saves = 0) const;
  void emit_bvh(int32_t node, int depth, FlatScene* out, int boundary, bool mark_roots, int saves) const;
  bool holds_medium(uint32_t obj) const;
  void flatten_program(const uint32_t* world, size_t n, FlatScene* out, size_t seg0, size_t seg1) const;
  bool flatten_pool2(const uint32_t* world, size_t n, FlatScene* out) const;
  void finish_materials(FlatScene* out) const;
};

struct BuildError {
  int code;
  std::string msg;
};

}  // namespace rtg
