// rtiow.hpp -- C++ spelling of the reference crate's surface over the C ABI (include/rtiow_gpu.h), so
// scene code from the reference's src/lib.rs / src/main.rs transliterates line for line:
//
//   Rust                                               C++
//   Box::new(object::Translate { offset, object })     boxed(object::Translate{offset, object})
//   object::Sphere { radius: 50., material: glass }    object::Sphere{50.f, glass}
//   object::Rect { orthogonal_to: StaticY, range0: 123. ..423., range1: 147. ..412., k: 554., material }
//                                                      object::Rect<StaticY>{{123.f,423.f},{147.f,412.f},554.f,material}
//   object::rotate_y(15., bvh)                         object::rotate_y(15.f, bvh)
//   bvh::from_scene(boxes, exposure)                   bvh::from_scene(std::move(boxes), exposure)
//   Camera::look(from, at, up, fov, aspect, ap, fd, exposure)   same
//   par_cast(NX, NY, NS, &camera, world)               par_cast(NX, NY, NS, camera, world)
//
// Objects are plain value types (like the Rust structs); they become builder handles only when
// par_cast flattens the world, mirroring how a Rust `-sys` binding would add `fn flatten(&self, b)`
// to `trait Object`.  The reference's panics surface as rtiow::Error.
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtiow_gpu.h"

namespace rtiow {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc != RTG_OK) throw Error(rc, rtg_last_error());
}
inline rtg_id check_id(rtg_id id) {
  if (id == RTG_INVALID_ID) throw Error(RTG_ERR_INVALID, rtg_last_error());
  return id;
}

// vec3.rs:13 -- only what scene construction needs (the hot-path arithmetic lives on the GPU)
struct Vec3 {
  float x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  static Vec3 from(float v) { return Vec3(v, v, v); }
  const float* data() const { return &x; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator*(float s, Vec3 v) { return Vec3(s * v.x, s * v.y, s * v.z); }
inline Vec3 operator+(float s, Vec3 v) { return Vec3(s + v.x, s + v.y, s + v.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }

struct Range {  // std::ops::Range<f32>
  float start, end;
};

// texture.rs:6 `Texture = Arc<dyn Fn(Vec3)->Vec3>`: closed here (H3: closures cannot be flattened)
struct Texture {
  std::function<rtg_id(rtg_builder*)> emit;
};
namespace texture {
inline Texture constant(Vec3 c) {
  return Texture{[c](rtg_builder* b) { return check_id(rtg_texture_constant(b, c.data())); }};
}
inline Texture checker(Texture t0, Texture t1) {
  return Texture{[t0, t1](rtg_builder* b) { return check_id(rtg_texture_checker(b, t0.emit(b), t1.emit(b))); }};
}
inline Texture perlin(float scale) {
  return Texture{[scale](rtg_builder* b) { return check_id(rtg_texture_perlin(b, scale)); }};
}
}  // namespace texture

// material.rs:10-39
struct Material {
  std::function<rtg_id(rtg_builder*)> emit;
  static Material Lambertian(Texture albedo) {
    return Material{[albedo](rtg_builder* b) { return check_id(rtg_material_lambertian(b, albedo.emit(b))); }};
  }
  static Material Metal(Vec3 albedo, float fuzz) {
    return Material{[albedo, fuzz](rtg_builder* b) { return check_id(rtg_material_metal(b, albedo.data(), fuzz)); }};
  }
  static Material Dielectric(float ref_idx) {
    return Material{[ref_idx](rtg_builder* b) { return check_id(rtg_material_dielectric(b, ref_idx)); }};
  }
  static Material DiffuseLight(Texture emission, float brightness) {
    return Material{[emission, brightness](rtg_builder* b) {
      return check_id(rtg_material_diffuse_light(b, emission.emit(b), brightness));
    }};
  }
  static Material Isotropic(Texture albedo) {
    return Material{[albedo](rtg_builder* b) { return check_id(rtg_material_isotropic(b, albedo.emit(b))); }};
  }
};

// `Box<dyn Object>` (object.rs:42): type-erased object that knows how to flatten itself
struct BoxedObject {
  std::function<rtg_id(rtg_builder*)> emit;
};
template <class O>
BoxedObject boxed(O o) {
  return BoxedObject{[o](rtg_builder* b) { return o.flatten(b); }};
}
using Scene = std::vector<BoxedObject>;  // Vec<Box<dyn Object>>

struct StaticX { static constexpr int AXIS = 0; };
struct StaticY { static constexpr int AXIS = 1; };
struct StaticZ { static constexpr int AXIS = 2; };

namespace object {
struct Sphere {  // object.rs:75-80
  float radius;
  Material material;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_sphere(b, radius, material.emit(b))); }
};
template <class A>
struct Rect {  // object.rs:131-144
  Range range0, range1;
  float k;
  Material material;
  rtg_id flatten(rtg_builder* b) const {
    return check_id(rtg_object_rect(b, A::AXIS, range0.start, range0.end, range1.start, range1.end, k, material.emit(b)));
  }
};
template <class O>
struct FlipNormals {  // object.rs:239
  O object;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_flip_normals(b, object.flatten(b))); }
};
template <class O>
FlipNormals<O> flip_normals(O o) { return FlipNormals<O>{std::move(o)}; }
template <class O>
struct Translate {  // object.rs:262-265
  Vec3 offset;
  O object;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_translate(b, offset.data(), object.flatten(b))); }
};
template <class O>
Translate(Vec3, O) -> Translate<O>;
template <class O>
struct Scale {  // object.rs:296-299
  Vec3 factor;
  O object;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_scale(b, factor.data(), object.flatten(b))); }
};
template <class O>
Scale(Vec3, O) -> Scale<O>;
template <class O>
struct RotateY {  // object.rs:335-339; built by rotate_y
  O object;
  float degrees;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_rotate_y(b, degrees, object.flatten(b))); }
};
template <class O>
RotateY<O> rotate_y(float degrees, O o) { return RotateY<O>{std::move(o), degrees}; }  // object.rs:477
template <class T, class S>
struct And {  // object.rs:394
  T first;
  S second;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_and(b, first.flatten(b), second.flatten(b))); }
};
template <class T, class S>
And(T, S) -> And<T, S>;
struct RectPrism {  // object.rs:420-473 rect_prism
  Vec3 p0, p1;
  Material material;
  rtg_id flatten(rtg_builder* b) const {
    return check_id(rtg_object_rect_prism(b, p0.data(), p1.data(), material.emit(b)));
  }
};
inline RectPrism rect_prism(Vec3 p0, Vec3 p1, Material m) { return RectPrism{p0, p1, std::move(m)}; }
template <class O>
struct LinearMove {  // object.rs:489-494
  O object;
  Vec3 motion;
  rtg_id flatten(rtg_builder* b) const { return check_id(rtg_object_linear_move(b, object.flatten(b), motion.data())); }
};
template <class O>
LinearMove(O, Vec3) -> LinearMove<O>;
template <class O>
struct ConstantMedium {  // object.rs:533-541
  O boundary;
  float density;
  Material material;
  rtg_id flatten(rtg_builder* b) const {
    return check_id(rtg_object_constant_medium(b, boundary.flatten(b), density, material.emit(b)));
  }
};
template <class O>
ConstantMedium(O, float, Material) -> ConstantMedium<O>;
}  // namespace object

namespace bvh {
struct Bvh {  // bvh.rs:9-19; itself an Object (bvh.rs:84)
  std::shared_ptr<Scene> objects;
  Range exposure;
  rtg_id flatten(rtg_builder* b) const {
    std::vector<rtg_id> ids;
    for (const auto& o : *objects) ids.push_back(o.emit(b));
    return check_id(rtg_object_bvh(b, ids.data(), ids.size(), exposure.start, exposure.end));
  }
};
inline Bvh from_scene(Scene scene, Range exposure) {  // bvh.rs:128
  return Bvh{std::make_shared<Scene>(std::move(scene)), exposure};
}
}  // namespace bvh

// camera.rs
struct Camera {
  rtg_camera c;
  static Camera look(Vec3 look_from, Vec3 look_at, Vec3 up, float fov, float aspect, float aperture,
                     float focus_dist, Range exposure) {
    Camera cam;
    check(rtg_camera_look(look_from.data(), look_at.data(), up.data(), fov, aspect, aperture, focus_dist,
                          exposure.start, exposure.end, &cam.c));
    return cam;
  }
};

// lib.rs:321 `pub struct Image(Vec<Vec<Vec3>>)`: rows top to bottom
struct Image {
  size_t nx = 0, ny = 0;
  std::vector<float> rgb;  // ny * nx * 3
};

struct CastOptions {
  uint64_t seed = 0xDEADBEEF;
  int device = 0;
  // perlin.rs:24-29 tables (only needed when a perlin texture is used)
  const float* perlin_vecs = nullptr;
  const uint8_t *perm_x = nullptr, *perm_y = nullptr, *perm_z = nullptr;
};

// par_cast, lib.rs:363: `world` is the list world of lib.rs:33; pass {boxed(bvh::from_scene(..))}
// for the `impl World for Bvh` of lib.rs:51.
inline Image par_cast(size_t nx, size_t ny, size_t ns, const Camera& camera, const Scene& world,
                      const CastOptions& opt = CastOptions()) {
  rtg_builder* b = nullptr;
  check(rtg_builder_create(&b));
  std::unique_ptr<rtg_builder, void (*)(rtg_builder*)> guard(b, rtg_builder_destroy);
  if (opt.perlin_vecs) check(rtg_builder_set_perlin_tables(b, opt.perlin_vecs, opt.perm_x, opt.perm_y, opt.perm_z));
  std::vector<rtg_id> ids;
  for (const auto& o : world) ids.push_back(o.emit(b));
  rtg_scene* s = nullptr;
  check(rtg_scene_create(b, ids.data(), ids.size(), opt.device, &s));
  std::unique_ptr<rtg_scene, void (*)(rtg_scene*)> sguard(s, rtg_scene_destroy);
  rtg_params p{};
  p.struct_size = sizeof(p);
  p.nx = (uint32_t)nx, p.ny = (uint32_t)ny, p.ns = (uint32_t)ns;
  p.max_bounces = 50;  // lib.rs:93
  p.t_near = 0.001f;   // lib.rs:35
  p.seed = opt.seed;
  Image img;
  img.nx = nx, img.ny = ny;
  img.rgb.assign(nx * ny * 3, 0.f);
  check(rtg_par_cast(s, &camera.c, &p, img.rgb.data(), nullptr));
  return img;
}

// print_ppm, lib.rs:344-361 (host post-process)
inline void print_ppm(const Image& image, FILE* out = stdout) {
  std::fprintf(out, "P3\n%zu %zu\n255\n", image.nx, image.ny);
  auto to_u8 = [](float x) {
    float v = 255.99f * x;
    int i = (v != v) ? 0 : (v >= 2147483648.f ? 2147483647 : (v <= -2147483648.f ? (-2147483647 - 1) : (int)v));
    return i < 0 ? 0 : (i > 255 ? 255 : i);
  };
  for (size_t i = 0; i < image.nx * image.ny; i++) {
    const float* c = &image.rgb[3 * i];
    std::fprintf(out, "%d %d %d\n", to_u8(std::sqrt(c[0])), to_u8(std::sqrt(c[1])), to_u8(std::sqrt(c[2])));
  }
}

}  // namespace rtiow
