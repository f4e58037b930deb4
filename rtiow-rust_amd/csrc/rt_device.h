// rt_device.h -- scalar building blocks of the hot path as gfx950 device code:
//   V3 arithmetic in the reference's exact operation order (src/vec3.rs),
//   the per-(pixel,sample) counter RNG (Philox4x32-10) with rand-0.6.5's float conversions,
//   (the libm restatements -- ln / powf(., 5) / sin, pinned to glibc -- live in rt_libm.h).
//
// Compile with -ffp-contract=off: rustc never fuses a*b+c, hipcc would (v_fmac_f32).  f32 divide and
// sqrt stay correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt is the hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_libm.h"

#define RT_DEV __device__ __forceinline__

namespace rtg {

struct V3 {
  float x, y, z;
};

RT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 splat(float s) { return V3{s, s, s}; }                                   // vec3.rs:106
RT_DEV V3 vadd(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }          // vec3.rs:155
RT_DEV V3 vsub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }          // vec3.rs:175
RT_DEV V3 vmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }          // vec3.rs:115
RT_DEV V3 vdiv(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }          // vec3.rs:135
RT_DEV V3 smul(float s, V3 v) { return V3{s * v.x, s * v.y, s * v.z}; }             // vec3.rs:125
RT_DEV V3 sdiv(V3 v, float s) { return V3{v.x / s, v.y / s, v.z / s}; }             // vec3.rs:145
RT_DEV V3 vneg(V3 v) { return V3{-v.x, -v.y, -v.z}; }                               // vec3.rs:185
RT_DEV float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }       // vec3.rs:43,100
RT_DEV float vlen(V3 a) { return __builtin_sqrtf(vdot(a, a)); }                     // vec3.rs:59
RT_DEV V3 vunit(V3 a) { return sdiv(a, vlen(a)); }                                  // vec3.rs:66
RT_DEV float vget(V3 v, uint32_t axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// Rust f32::max / f32::min: NaN-ignoring (maxnum); lowers to v_max_f32 / v_min_f32 (IEEE mode).
RT_DEV float rs_max(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float rs_min(float a, float b) { return __builtin_fminf(a, b); }

// vec3.rs:313  v - ((2*(v.n)) * n)
RT_DEV V3 reflect(V3 v, V3 n) { return vsub(v, smul(2.f * vdot(v, n), n)); }

constexpr float F32_MAX = 3.402823466e+38f;

// ---- counter RNG ---------------------------------------------------------------------------------
// Determinism contract: one Philox4x32-10 stream per (seed, pixel, sample, event); key =
// seed, counter = (block, sample, pixel, event); each block hands out its 4 words in order.  event 0 =
// camera ray, event k = k-th hit_top + scatter: every event starts on a fresh block, so the lanes of a
// wave walk the rejection loops in lock-step (block generation is never lane-divergent).
struct SampleRng {
  uint32_t k0, k1, sample, pixel, event, blk;
  uint32_t b0, b1, b2, b3;  // unread words of the current block, b0 next
  uint32_t left;            // words left in b0..b3
  uint32_t draws;

  RT_DEV void init(uint64_t seed, uint32_t pixel_, uint32_t sample_) {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
    pixel = pixel_;
    sample = sample_;
    event = 0;
    blk = 0;
    left = 0;
    draws = 0;
  }
  RT_DEV void set_event(uint32_t e) {
    event = e;
    blk = 0;
    left = 0;
  }
  RT_DEV void refill() {
    uint32_t c0 = blk, c1 = sample, c2 = pixel, c3 = event;
    uint32_t key0 = k0, key1 = k1;
#ifndef RT_PHILOX_ROUNDS
#define RT_PHILOX_ROUNDS 10
#endif
#pragma unroll
    for (int round = 0; round < RT_PHILOX_ROUNDS; round++) {
      // one 32x32->64 multiply each (v_mad_u64_u32) instead of separate mul_hi + mul_lo
      const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
      const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
      uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
      uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      uint32_t n0 = hi1 ^ c1 ^ key0;
      uint32_t n2 = hi0 ^ c3 ^ key1;
      c0 = n0;
      c1 = lo1;
      c2 = n2;
      c3 = lo0;
      key0 += 0x9E3779B9u;
      key1 += 0xBB67AE85u;
    }
    b0 = c0, b1 = c1, b2 = c2, b3 = c3;
    blk++;
    left = 4;
  }
  // continue the current event's stream after `k` words were already consumed elsewhere (the medium
  // draws of this event's traversal); does not count as draws of this generator
  RT_DEV void seek(uint32_t k) {
    blk = k >> 2;
    left = 0;
    const uint32_t skip = k & 3u;
    if (skip) {
      refill();
      if (skip >= 1u) b0 = b1, b1 = b2, b2 = b3;
      if (skip >= 2u) b0 = b1, b1 = b2;
      if (skip >= 3u) b0 = b1;
      left = 4u - skip;
    }
  }
  RT_DEV uint32_t next_u32() {
    if (left == 0) refill();
    uint32_t r = b0;
    b0 = b1, b1 = b2, b2 = b3;
    left--;
    draws++;
    return r;
  }
  // rand 0.6.5 Standard for f32: (u32 >> 8) * 2^-24
  RT_DEV float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
  // rand 0.6.5 UniformFloat::sample_single (camera.rs:55)
  RT_DEV float gen_range(float low, float high) {
    float scale = high - low;
    for (;;) {
      float value1_2 = __uint_as_float((next_u32() >> 9) | 0x3f800000u);
      float res = (value1_2 - 1.0f) * scale + low;
      if (res < high) return res;
    }
  }
};

// vec3.rs:19-26
RT_DEV V3 in_unit_sphere(SampleRng& rng) {
  for (;;) {
    float a = rng.gen_f32();
    float b = rng.gen_f32();
    float c = rng.gen_f32();
    V3 v = vsub(smul(2.f, mk(a, b, c)), splat(1.f));
    if (vdot(v, v) < 1.f) return v;
  }
}
// the same loop cut off after `max_tries` attempts: false = no direction yet, the stream stands behind the last attempt
RT_DEV bool in_unit_sphere_tries(SampleRng& rng, uint32_t max_tries, V3& out) {
  for (uint32_t k = 0; k < max_tries; k++) {
    float a = rng.gen_f32();
    float b = rng.gen_f32();
    float c = rng.gen_f32();
    V3 v = vsub(smul(2.f, mk(a, b, c)), splat(1.f));
    if (vdot(v, v) < 1.f) {
      out = v;
      return true;
    }
  }
  return false;
}
// vec3.rs:32-39
RT_DEV V3 in_unit_disc(SampleRng& rng) {
  for (;;) {
    float a = rng.gen_f32();
    float b = rng.gen_f32();
    V3 v = vsub(smul(2.f, mk(a, b, 0.f)), mk(1.f, 1.f, 0.f));
    if (vdot(v, v) < 1.f) return v;
  }
}

}  // namespace rtg
