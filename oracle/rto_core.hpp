// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
//
// PARITY UNPINNED: the reference (cbiffle/rtiow-rust) is Rust, cannot be compiled in this image
// (no rustc/cargo, crates not vendored) and holds no golden vectors / known-answer tests for the
// color() hot path.  This oracle is a line-by-line CPU restatement of the
// reference arithmetic; each function cites the reference file:line it follows.  What IS pinned against the reference:
// the two pictures it publishes (img/demo-scene.jpg, img/rttnw-final.jpg) correlate with this oracle's renders of the
// same seeded scenes at 0.99 / 0.92 (tests/test_reference_image.py) -- scene construction, RNG stream, camera and
// geometry, not low-order bits.
//
// rto_core.hpp: Vec3 / Ray / Aabb math, RNG front-ends and the shared libm restatements.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "rto_libm.hpp"

namespace rto {

// ------------------------------------------------------------------------------------------------
// Vec3 -- reference src/vec3.rs:13 `pub struct Vec3(pub f32, pub f32, pub f32)`.
// Operation ORDER is part of the contract (f32 is not associative): every operator below performs
// exactly the scalar operations, in the order, the Rust operator impl performs.
// ------------------------------------------------------------------------------------------------
struct Vec3 {
  float x = 0.f, y = 0.f, z = 0.f;  // vec3.rs:12 derive(Default) -> (0,0,0)
  Vec3() = default;
  Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  // vec3.rs:106-111 From<f32>: broadcast
  static Vec3 from(float v) { return Vec3(v, v, v); }
  float operator[](int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }  // vec3.rs:287-298
  float& at(int axis) { return axis == 0 ? x : (axis == 1 ? y : z); }               // vec3.rs:300-309
};

// vec3.rs:115-122  Vec3 * Vec3 (element-wise)
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
// vec3.rs:125-132  f32 * Vec3 == Vec3::from(self) * rhs
inline Vec3 operator*(float s, Vec3 v) { return Vec3::from(s) * v; }
// vec3.rs:135-142  Vec3 / Vec3
inline Vec3 operator/(Vec3 a, Vec3 b) { return Vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
// vec3.rs:145-152  Vec3 / f32 : per-component DIVIDE (never a reciprocal multiply)
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
// vec3.rs:155-162  Vec3 + Vec3
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
// vec3.rs:165-172  f32 + Vec3 : self + x
inline Vec3 operator+(float s, Vec3 v) { return Vec3(s + v.x, s + v.y, s + v.z); }
// vec3.rs:175-182  Vec3 - Vec3
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
// vec3.rs:185-192  -Vec3
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }

// vec3.rs:43-46,100-102  dot = reduce(add) over zip_with(mul) = (x*x' + y*y') + z*z'
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vec3.rs:49-55
inline Vec3 cross(Vec3 a, Vec3 b) {
  return Vec3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
// vec3.rs:59-61
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// vec3.rs:66-68
inline Vec3 into_unit(Vec3 a) { return a / length(a); }

// Rust `f32::max` / `f32::min` (used at aabb.rs:10-11,26-27; object.rs:385,553-554; bvh.rs:30-32):
// "if one of the arguments is NaN, the other is returned".  Sign of zero on equal inputs is
// unspecified in Rust/LLVM (maxnum); every call site only feeds comparisons or non-zero values.
inline float rs_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
inline float rs_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }

// vec3.rs:313-315  v - 2. * v.dot(n) * n   ==  v - ((2*dot) * n)
inline Vec3 reflect(Vec3 v, Vec3 n) { return v - (2.f * dot(v, n)) * n; }

// vec3.rs:321-330
inline bool refract(Vec3 v, Vec3 n, float ni_over_nt, Vec3* out) {
  Vec3 uv = into_unit(v);
  float dt = dot(uv, n);
  float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
  if (discriminant > 0.f) {
    *out = ni_over_nt * (uv - dt * n) - std::sqrt(discriminant) * n;
    return true;
  }
  return false;
}

// ray.rs:5-17
struct Ray {
  Vec3 origin, direction;
  float time = 0.f;
  Vec3 point_at_parameter(float t) const { return origin + t * direction; }
};

constexpr float F32_MAX = std::numeric_limits<float>::max();
constexpr float F32_MIN = -std::numeric_limits<float>::max();  // Rust std::f32::MIN == -MAX

struct Range {
  float start, end;
};

// ------------------------------------------------------------------------------------------------
// Instrumentation shared by all hit functions: the counters SURVEY.md 8(d) prices the path with.
// ------------------------------------------------------------------------------------------------
struct Counters {
  uint64_t aabb_tests = 0;   // N: Aabb::hit calls
  uint64_t prim_tests = 0;   // P: Sphere::hit + Rect::hit calls (incl. medium boundary calls)
  uint64_t shaded_hits = 0;  // H: hit_top() results that reach Material::{emitted,scatter}
  uint64_t rays = 0;         // hit_top() calls
  uint64_t draws = 0;        // RNG u32 draws
  void add(const Counters& o) {
    aabb_tests += o.aabb_tests;
    prim_tests += o.prim_tests;
    shaded_hits += o.shaded_hits;
    rays += o.rays;
    draws += o.draws;
  }
};

// aabb.rs:4-44
struct Aabb {
  Vec3 min, max;
  Aabb merge(Aabb o) const {  // aabb.rs:9-14
    return Aabb{Vec3(rs_min(min.x, o.min.x), rs_min(min.y, o.min.y), rs_min(min.z, o.min.z)),
                Vec3(rs_max(max.x, o.max.x), rs_max(max.y, o.max.y), rs_max(max.z, o.max.z))};
  }
  bool hit(const Ray& ray, Range t_range, Counters* c) const {  // aabb.rs:16-27
    if (c) c->aabb_tests++;
    Vec3 inv_d(1.f / ray.direction.x, 1.f / ray.direction.y, 1.f / ray.direction.z);
    Vec3 t0 = (min - ray.origin) * inv_d;
    Vec3 t1 = (max - ray.origin) * inv_d;
    Vec3 a(inv_d.x < 0.f ? t1.x : t0.x, inv_d.y < 0.f ? t1.y : t0.y, inv_d.z < 0.f ? t1.z : t0.z);
    Vec3 b(inv_d.x < 0.f ? t0.x : t1.x, inv_d.y < 0.f ? t0.y : t1.y, inv_d.z < 0.f ? t0.z : t1.z);
    float start = rs_max(t_range.start, rs_max(rs_max(a.x, a.y), a.z));
    float end = rs_min(t_range.end, rs_min(rs_min(b.x, b.y), b.z));
    return end > start;
  }
  Vec3 corner(int i) const {  // aabb.rs:29-43: x outermost, z innermost
    return Vec3((i & 4) ? max.x : min.x, (i & 2) ? max.y : min.y, (i & 1) ? max.z : min.z);
  }
};

inline uint32_t f32_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float f32_from_bits(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// ------------------------------------------------------------------------------------------------
// RNG front-ends.  Third-party arithmetic: rand 0.6.5 / rand_core 0.4.2 / rand_pcg 0.1.2
// (Cargo.lock), NOT under /root/reference -> restated from the published algorithms.
// Call sites: lib.rs:41,53,368,369,389,390; material.rs:97; vec3.rs:21,34,212; camera.rs:55.
// ------------------------------------------------------------------------------------------------
struct Rng {
  Counters* counters = nullptr;
  virtual ~Rng() = default;
  virtual uint32_t next_u32_impl() = 0;
  // Counter-stream hook (see SampleRng): a new path event starts.  A sequential generator such as
  // SmallRng -- the reference's own `impl Rng` -- has no such notion and ignores it.
  virtual void set_event(uint32_t) {}
  uint32_t next_u32() {
    if (counters) counters->draws++;
    return next_u32_impl();
  }
  // rand 0.6.5 Standard for f32: 24 high bits -> [0,1):  (u32 >> 8) * 2^-24
  float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
  // rand 0.6.5 UniformFloat::<f32>::sample_single(low, high): value0_1 = bits(u32>>9 | 1.0) - 1;
  // res = value0_1 * (high-low) + low; retry while res >= high.   (camera.rs:55)
  float gen_range_f32(float low, float high) {
    float scale = high - low;
    for (;;) {
      float value1_2 = f32_from_bits((next_u32() >> 9) | 0x3f800000u);
      float value0_1 = value1_2 - 1.0f;
      float res = value0_1 * scale + low;
      if (res < high) return res;
    }
  }
};

// rand::rngs::SmallRng on 64-bit targets == rand_pcg::Pcg64Mcg (Mcg128Xsl64):
//   state' = state * 0x2360ED051FC65DA44385DF649FCCF645 (mod 2^128)
//   out    = rotr64((state' >> 64) ^ state', state' >> 122)
// next_u32 = low 32 bits of next_u64.  seed_from_u64 (rand_core 0.4.2) expands the u64 with a
// PCG32 stream (mul 6364136223846793005, inc 11634580027462260723, XSH-RR) into 16 LE seed bytes;
// from_seed ORs the state with 1.
struct SmallRng final : Rng {
  unsigned __int128 state;
  explicit SmallRng(uint64_t seed) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    uint32_t w[4];
    uint64_t s = seed;
    for (int i = 0; i < 4; i++) {
      s = s * MUL + INC;
      uint32_t xorshifted = (uint32_t)(((s >> 18) ^ s) >> 27);
      uint32_t rot = (uint32_t)(s >> 59);
      w[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    unsigned __int128 st = 0;
    for (int i = 3; i >= 0; i--) st = (st << 32) | w[i];
    state = st | 1;
  }
  uint64_t next_u64() {
    const unsigned __int128 M =
        ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | (unsigned __int128)0x4385DF649FCCF645ull;
    state = state * M;
    uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    unsigned rot = (unsigned)(hi >> 58);
    uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
  }
  uint32_t next_u32_impl() override { return (uint32_t)next_u64(); }
};

// The determinism contract of this build: one counter-based stream per
// (seed, pixel, sample, event).  Philox4x32-10 (Salmon et al., SC'11; Random123), key = seed lo/hi,
// counter = (block, sample, pixel, event); a block yields 4 draws, consumed in order x,y,z,w.
// event 0 = the camera ray of the sample (u, v, lens disc, shutter time; lib.rs:368-370);
// event k >= 1 = the k-th hit_top() of the path plus the scatter that follows it (lib.rs:73-84).
// Starting every event on a fresh block keeps a GPU wave's lanes in lock-step through the
// rejection loops (no lane-divergent block generation) and makes a path's RNG state just (k).
struct Philox4x32 {
  static void block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]) {
    for (int round = 0; round < 10; round++) {
      uint64_t p0 = (uint64_t)0xD2511F53u * c0;
      uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
      uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
      uint32_t n1 = (uint32_t)p1;
      uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
      uint32_t n3 = (uint32_t)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

struct SampleRng final : Rng {
  uint32_t k0, k1, sample, pixel, event = 0, blk = 0, idx = 4;
  uint32_t buf[4];
  SampleRng(uint64_t seed, uint32_t pixel_, uint32_t sample_)
      : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), sample(sample_), pixel(pixel_) {}
  void set_event(uint32_t e) override { event = e, blk = 0, idx = 4; }
  uint32_t next_u32_impl() override {
    if (idx == 4) {
      Philox4x32::block(k0, k1, blk, sample, pixel, event, buf);
      blk++;
      idx = 0;
    }
    return buf[idx++];
  }
};

// vec3.rs:209-214  Standard -> Vec3: x, y, z drawn in order
inline Vec3 gen_vec3(Rng& rng) {
  float a = rng.gen_f32();
  float b = rng.gen_f32();
  float c = rng.gen_f32();
  return Vec3(a, b, c);
}
// vec3.rs:19-26
inline Vec3 in_unit_sphere(Rng& rng) {
  for (;;) {
    Vec3 v = 2.f * gen_vec3(rng) - Vec3::from(1.f);
    if (dot(v, v) < 1.f) return v;
  }
}
// vec3.rs:32-39  (z term is 2*0 - 0)
inline Vec3 in_unit_disc(Rng& rng) {
  for (;;) {
    float a = rng.gen_f32();
    float b = rng.gen_f32();
    Vec3 v = 2.f * Vec3(a, b, 0.f) - Vec3(1.f, 1.f, 0.f);
    if (dot(v, v) < 1.f) return v;
  }
}

}  // namespace rto
