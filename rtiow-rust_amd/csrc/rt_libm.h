// rt_libm.h -- the platform libm's f32 ln / powf(., 5) / sin as gfx950 device code.
// The reference calls f32::ln (object.rs:562), f32::powf(x, 5.) (material.rs:145) and f32::sin (texture.rs:14), which
// lower to the PLATFORM libm (llvm.log / pow / sin.f32 -> logf / powf / sinf).  A GPU cannot call glibc, so these are
// restatements of glibc's own algorithms (glibc 2.28+ sysdeps/ieee754/flt-32/{e_logf,e_powf,s_sinf}.c = ARM
// optimized-routines: table + double-precision polynomial, one final rounding), with every fused multiply-add written
// out the way GCC contracts that source for glibc's FMA ifunc variants (__logf_fma / __powf_fma / __sinf_fma, what an
// x86-64 CPU with FMA -- this container's and the GPU box's -- dispatches to).
// PINNED: bit-identical to glibc 2.35's logf(x), powf(x, 5.0f) and sinf(x) on ALL 2^32 inputs
// (tests/test_libm.py::test_restatements_equal_the_platform_libm_on_every_float, the oracle's copy against the host libm;
// the GPU's copy against the oracle's in tests/test_parity_gpu.py and against the host libm in tools/libm_exhaustive_gpu.py).
// Without FMA contraction (glibc's __powf_sse2 / __sinf_sse2 on pre-Haswell CPUs) powf(x, 5) differs on 6 and sinf on 12
// of the 2^32 inputs by one ulp; logf is identical either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtg {
__device__ __forceinline__ uint32_t gl_asuint(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float gl_asfloat(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint64_t gl_asuint64(double f) { return (uint64_t)__double_as_longlong(f); }
__device__ __forceinline__ double gl_asdouble(uint64_t u) { return __longlong_as_double((long long)u); }
// ln and sin out of line: rare on the hot path (one ln per medium evaluation, sin only under a checker texture)
#define GL_FN __device__ __attribute__((noinline))
#define GL_TAB __device__
#define GL_FMA(a, b, c) __builtin_fma((a), (b), (c))
struct GlLogTab { double invc, logc; };
GL_TAB const GlLogTab kGlLogfTab[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};
GL_TAB const GlLogTab kGlPowLog2Tab[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
  { 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 },  { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
  { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
  { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 },  { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 },
  { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
  { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },  { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};
GL_TAB const uint64_t kGlExp2fTab[32] = {
  0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
  0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
  0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
  0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
  0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
  0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
GL_TAB const uint32_t kGlInvPio4[24] = { 0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27,
  0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
  0x993c4390, 0x3c439041 };

// glibc 2.28+ sysdeps/ieee754/flt-32/e_logf.c (ARM optimized-routines logf): 16-entry table, degree-3 polynomial in double
__device__ __attribute__((always_inline)) float rt_logf_inline(float x) {
  uint32_t ix = gl_asuint(x);
  if (ix == 0x3f800000u) return 0.f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0u) return -__builtin_inff();
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
    ix = gl_asuint(x * 0x1p23f);  // subnormal: normalise
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const int32_t k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = kGlLogfTab[i].invc, logc = kGlLogfTab[i].logc;
  const double z = (double)gl_asfloat(iz);
  const double r = GL_FMA(z, invc, -1.0);
  const double y0 = GL_FMA((double)k, 0x1.62e42fefa39efp-1, logc);
  const double r2 = r * r;
  double y = GL_FMA(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = GL_FMA(-0x1.00ea348b88334p-2, r2, y);
  y = GL_FMA(y, r2, y0 + r);
  return (float)y;
}

GL_FN float rt_logf(float x) { return rt_logf_inline(x); }

// glibc e_powf.c with y = 5.0f (schlick, material.rs:145): log2 via a 16-entry table + degree-5 polynomial, exp2 via a 32-entry table
// (a template so that the lean kernel's SCATTER pass can take it inline -- C2 7.97 -> 7.90 ms -- while the full-feature kernels,
// which sit at the 128-VGPR limit, call the out-of-line copy)
template <bool INLINE>
__device__ __attribute__((always_inline)) float rt_pow5f_body(float x) {
  uint32_t sign_bias = 0u, ix = gl_asuint(x);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {  // x is +-0, +-inf or nan; y = 5 is a positive odd integer
      float x2 = x * x;
      if (ix & 0x80000000u) x2 = -x2;
      return x2;
    }
    if (ix & 0x80000000u) sign_bias = 1u << 16, ix &= 0x7fffffffu;  // x < 0: the sign comes back through the exp2 table shift
    if (ix < 0x00800000u) {
      ix = gl_asuint(x * 0x1p23f);
      ix &= 0x7fffffffu;
      ix -= 23u << 23;
    }
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int32_t k = (int32_t)top >> 23;
  const double invc = kGlPowLog2Tab[i].invc, logc = kGlPowLog2Tab[i].logc, z = (double)gl_asfloat(iz);
  const double r = GL_FMA(z, invc, -1.0), y0 = logc + (double)k;
  const double r2 = r * r;
  double y = GL_FMA(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
  const double p = GL_FMA(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
  const double r4 = r2 * r2;
  double q = GL_FMA(0x1.71547652ab82bp0, r, y0);
  q = GL_FMA(p, r2, q);
  y = GL_FMA(y, r4, q);
  const double ylogx = 5.0 * y;
  if (((gl_asuint64(ylogx) >> 47) & 0xffffu) >= (gl_asuint64(126.0) >> 47)) {
    if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
    if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
  }
  double kd = ylogx + 0x1.8p+52 / 32;
  const uint64_t ki = gl_asuint64(kd);
  kd -= 0x1.8p+52 / 32;
  const double rr = ylogx - kd;
  uint64_t t = kGlExp2fTab[ki & 31u];
  t += (ki + sign_bias) << 47;
  const double s = gl_asdouble(t);
  const double zz = GL_FMA(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3), rr2 = rr * rr;
  double yy = GL_FMA(0x1.62e42ff0c52d6p-1, rr, 1.0);
  yy = GL_FMA(zz, rr2, yy);
  yy = yy * s;
  return (float)yy;
}

GL_FN float rt_pow5f(float x) { return rt_pow5f_body<false>(x); }
__device__ __forceinline__ float rt_pow5f_inline(float x) { return rt_pow5f_body<true>(x); }

// glibc s_sinf.c (ARM optimized-routines sinf): double-precision polynomials, table-driven reduction for |x| >= 120
GL_FN float gl_sinf_poly(double x, double x2, uint32_t neg, uint32_t n) {  // neg: the coefficient set of the negated cosine
  const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
  if ((n & 1u) == 0u) {
    const double x3 = x * x2, s1 = GL_FMA(x2, s3c, s2c), x7 = x3 * x2, s = GL_FMA(x3, s1c, x);
    return (float)GL_FMA(x7, s1, s);
  }
  const double sg = neg ? -1.0 : 1.0;
  const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3c = sg * -0x1.6c087e89a359dp-10,
               c4c = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2, c2 = GL_FMA(x2, c4c, c3c), c1 = GL_FMA(x2, c1c, c0), x6 = x4 * x2, c = GL_FMA(x4, c2c, c1);
  return (float)GL_FMA(x6, c2, c);
}
GL_FN float rt_sinf(float y) {
  const uint32_t top = (gl_asuint(y) >> 20) & 0x7ffu;  // abstop12
  double x = (double)y;
  if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {  // |y| < pi/4
    if (top < ((0x39800000u >> 20) & 0x7ffu)) return y;  // |y| < 2^-12
    return gl_sinf_poly(x, x * x, 0u, 0u);
  }
  if (top < ((0x42f00000u >> 20) & 0x7ffu)) {  // |y| < 120: quadrant from a scaled float -> int conversion
    const double r = x * 0x1.45F306DC9C883p+23;
    const int32_t n = ((int32_t)r + 0x800000) >> 24;
    x = GL_FMA(-(double)n, 0x1.921FB54442D18p0, x);
    const double s = (((uint32_t)n + 1u) & 2u) ? -1.0 : 1.0;  // sign[n & 3] of {1, -1, -1, 1}
    return gl_sinf_poly(x * s, x * x, (uint32_t)(n & 2), (uint32_t)n);
  }
  if (top < 0x7f8u) {  // finite: 192-bit fixed-point reduction with the bits of 4/pi
    uint32_t xi = gl_asuint(y);
    const uint32_t sign = xi >> 31;
    const uint32_t* arr = &kGlInvPio4[(xi >> 26) & 15u];
    const uint32_t shift = (xi >> 23) & 7u;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4], res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t nn = (res0 + (1ull << 61)) >> 62;
    res0 -= nn << 62;
    x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
    const uint32_t n = (uint32_t)nn;
    const double s = ((n + sign + 1u) & 2u) ? -1.0 : 1.0;
    return gl_sinf_poly(x * s, x * x, (n + sign) & 2u, n);
  }
  return __builtin_nanf("");  // inf, nan
}
#undef GL_FN
#undef GL_TAB
#undef GL_FMA
}  // namespace rtg
