// crate_mirror_demo.cpp -- the reference's src/main.rs, transliterated against rtiow.hpp (C++ mirror of the crate
// over the C ABI).  Scene bodies follow src/lib.rs:103-193 and src/main.rs:11-108 line for line.
//   g++ -std=c++17 -O2 crate_mirror_demo.cpp -o rtiow_main -L../../csrc -lrtiow_gpu -Wl,-rpath,'$ORIGIN/../../csrc'
//   ./rtiow_main cornell 300 300 100 > out.ppm
#include <chrono>
#include <cstdio>
#include <cstring>
#include <tuple>

#include "../rtiow.hpp"

using namespace rtiow;

// lib.rs:103-166
static Scene cornell_box() {
  auto diffuse_color = [](Vec3 c) { return Material::Lambertian(texture::constant(c)); };
  Material red = diffuse_color(Vec3(0.65f, 0.05f, 0.05f));
  Material white = diffuse_color(Vec3::from(0.73f));
  Material green = diffuse_color(Vec3(0.12f, 0.45f, 0.15f));
  Material light = Material::DiffuseLight(texture::constant(Vec3::from(1.f)), 15.f);
  Scene s;
  s.push_back(boxed(object::Rect<StaticY>{{213.f, 343.f}, {227.f, 332.f}, 554.f, light}));
  s.push_back(boxed(object::Rect<StaticY>{{0.f, 555.f}, {0.f, 555.f}, 0.f, white}));                          // floor
  s.push_back(boxed(object::flip_normals(object::Rect<StaticZ>{{0.f, 555.f}, {0.f, 555.f}, 555.f, white})));  // rear wall
  s.push_back(boxed(object::flip_normals(object::Rect<StaticY>{{0.f, 555.f}, {0.f, 555.f}, 555.f, white})));  // ceiling
  s.push_back(boxed(object::Rect<StaticX>{{0.f, 555.f}, {0.f, 555.f}, 0.f, red}));                            // right wall
  s.push_back(boxed(object::flip_normals(object::Rect<StaticX>{{0.f, 555.f}, {0.f, 555.f}, 555.f, green})));  // left wall
  return s;
}

// lib.rs:168-193
static Scene cornell_box_with_boxes() {
  Scene scene = cornell_box();
  Material white = Material::Lambertian(texture::constant(Vec3::from(0.73f)));
  scene.push_back(boxed(object::Translate{
      Vec3(130.f, 0.f, 65.f),
      object::rotate_y(-18.f, object::rect_prism(Vec3(0.f, 0.f, 0.f), Vec3(165.f, 165.f, 165.f), white))}));
  scene.push_back(boxed(object::Translate{
      Vec3(265.f, 0.f, 295.f),
      object::rotate_y(15.f, object::rect_prism(Vec3(0.f, 0.f, 0.f), Vec3(165.f, 330.f, 165.f), white))}));
  return scene;
}

static Camera cornell_camera(size_t nx, size_t ny, Range exposure) {  // main.rs:12-27
  return Camera::look(Vec3(278.f, 278.f, -800.f), Vec3(278.f, 278.f, 0.f), Vec3(0.f, 1.f, 0.f), 40.f,
                      (float)nx / (float)ny, 0.0f, 10.f, exposure);
}

// main.rs:11-30
static std::tuple<Scene, Camera, Range> cornell_box_scene(size_t nx, size_t ny) {
  Range exposure{0.f, 1.f};
  return {cornell_box_with_boxes(), cornell_camera(nx, ny, exposure), exposure};
}

// main.rs:33-67
static std::tuple<Scene, Camera, Range> motion_test(size_t nx, size_t ny) {
  Range exposure{0.f, 1.f};
  Scene scene = cornell_box();
  scene.push_back(boxed(object::Translate{
      Vec3(278.f, 278.f, 278.f),
      object::LinearMove{object::Sphere{65.f, Material::Lambertian(texture::constant(Vec3::from(0.73f)))},
                         Vec3(0.f, 100.f, 0.f)}}));
  return {scene, cornell_camera(nx, ny, exposure), exposure};
}

// main.rs:70-108
static std::tuple<Scene, Camera, Range> volume_test(size_t nx, size_t ny) {
  Range exposure{0.f, 1.f};
  Scene scene = cornell_box();
  scene.push_back(boxed(object::Translate{
      Vec3(278.f, 278.f, 278.f),
      object::ConstantMedium{object::Sphere{180.f, Material::Lambertian(texture::constant(Vec3::from(0.73f)))},
                             0.01f, Material::Isotropic(texture::constant(Vec3(0.2f, 0.2f, 1.0f)))}}));
  return {scene, cornell_camera(nx, ny, exposure), exposure};
}

static const bool USE_BVH = false;  // main.rs:321

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "cornell";
  size_t NX = argc > 2 ? (size_t)atoi(argv[2]) : 300, NY = argc > 3 ? (size_t)atoi(argv[3]) : 300;
  size_t NS = argc > 4 ? (size_t)atoi(argv[4]) : 100;
  std::fprintf(stderr, "Parallel casting %zu x %zu image using %zux oversampling.\n", NX, NY, NS);
  try {
    auto [world, camera, exposure] = !strcmp(which, "motion")   ? motion_test(NX, NY)
                                     : !strcmp(which, "volume") ? volume_test(NX, NY)
                                                                : cornell_box_scene(NX, NY);
    auto start = std::chrono::steady_clock::now();
    Image image;
    if (USE_BVH) {
      std::fprintf(stderr, "Generating bounding volume hierarchy.\n");
      Scene top{boxed(bvh::from_scene(world, exposure))};
      image = par_cast(NX, NY, NS, camera, top);
    } else {
      std::fprintf(stderr, "Testing every ray against every object.\n");
      image = par_cast(NX, NY, NS, camera, world);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::fprintf(stderr, "Took %.3fs wall time\n", secs);
    print_ppm(image);
  } catch (const Error& e) {
    std::fprintf(stderr, "rtiow error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
