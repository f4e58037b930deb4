// rt_persistent.h -- the tuned kernel for "lean" scenes (spheres under Bvhs, constant textures: the
// book-1 north-star workload).  Same arithmetic as rt_trace.h, different SCHEDULE:
//
//  * persistent wavefronts: the grid is sized to the chip (CUs x resident workgroups); every lane
//    owns one pixel at a time and pulls the next one from a global work counter with ONE
//    wave-aggregated atomicAdd per refill (ballot + mbcnt), so no wave idles in a tail;
//  * ray regeneration: sample loop and bounce loop are flattened into one state machine
//    (NEED_PIXEL -> GEN -> TRAV -> SHADE -> ...); a lane whose path ended starts its next sample at
//    once instead of waiting for the wave's longest path.  Wave-ballots decide, wave-uniformly, when
//    enough lanes wait in SHADE to make a shading pass worthwhile;
//  * the flat program is staged into LDS once per workgroup (ds_read_b128 gathers instead of L1
//    misses); box steps and sphere steps are separate wave-uniform phases -- lanes that reached a
//    SPHERE record park until enough of them wait (or no box lane is left), so the expensive
//    sqrt/divide sequence never runs for a handful of lanes;
//  * the hit point / normal are NOT computed during traversal: only (t, pc of the winning record)
//    is kept, and p, n are rebuilt once in SHADE from the same operands in the same order
//    (bit-identical to doing it at hit time).
//
// Scheduling never changes results: each (pixel, sample) is an independent, deterministic
// computation keyed by its own RNG stream, and a pixel's samples are still folded in order by the
// single lane that owns the pixel.
#pragma once
#include "rt_trace.h"

namespace rtg {

constexpr uint32_t ST_NEED_PIXEL = 0, ST_GEN = 1, ST_TRAV = 2, ST_SHADE = 3, ST_DEAD = 4;
constexpr uint32_t NO_HIT = 0xffffffffu;

// Scheduling knobs (never affect results): lanes waiting in SHADE before traversal yields to a shading
// pass, and lanes parked at a SPHERE record before a sphere pass runs.
struct Tuning {
  uint32_t regen_min;
  uint32_t sphere_min;
  uint32_t box_leave;  // a box run re-evaluates the schedule after this many lanes left the BOX state
};

RT_DEV uint32_t lane_rank(uint64_t mask) {  // number of set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// work item -> pixel.  Work items enumerate this rank's tiles (tile % nranks == rank) in order, each
// tile as 8x8 blocks, so the 64 items a wave grabs at start form one coherent 8x8 block.
RT_DEV bool work_to_pixel(const DevParams& P, uint32_t w, uint32_t& x, uint32_t& row) {
  const uint32_t px_per_tile = P.tile_w * P.tile_h;
  const uint32_t tiles_x = (P.nx + P.tile_w - 1u) / P.tile_w;
  uint32_t k = w / px_per_tile, r = w - k * px_per_tile;
  uint32_t tile = P.rank + k * P.nranks;
  uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  uint32_t blocks_x = P.tile_w >> 3;
  uint32_t b = r >> 6, l = r & 63u;
  uint32_t bx = b % blocks_x, by = b / blocks_x;
  x = tx * P.tile_w + bx * 8u + (l & 7u);
  row = ty * P.tile_h + by * 8u + (l >> 3);
  return x < P.nx && row < P.ny;
}

template <bool USE_LDS, bool COUNT>
__global__ __launch_bounds__(512) void render_lean_persistent(DevScene sc, DevCamera cam, DevParams P,
                                                              float* __restrict__ out, uint32_t total_work,
                                                              uint32_t* __restrict__ queue,
                                                              unsigned long long* counters, Tuning tune) {
  extern __shared__ uint4 s_prog[];  // [0, n) = lo packets, [n, 2n) = hi packets, [2n, 2n + 2m) = materials
  const uint32_t n_prog = sc.n_prog;
  if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < n_prog; i += blockDim.x) {
      s_prog[i] = sc.lo[i];
      s_prog[n_prog + i] = sc.hi[i];
    }
    for (uint32_t i = threadIdx.x; i < 2u * sc.n_mat; i += blockDim.x) s_prog[2u * n_prog + i] = sc.mat[i];
    __syncthreads();
  }
#define RT_FETCH_LO(pc_) (USE_LDS ? s_prog[(pc_)] : sc.lo[(pc_)])
#define RT_FETCH_HI(pc_) (USE_LDS ? s_prog[n_prog + (pc_)] : sc.hi[(pc_)])
#define RT_FETCH_MAT(i_) (USE_LDS ? s_prog[2u * n_prog + (i_)] : sc.mat[(i_)])

  const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
  const float t_near = P.t_near;

  // ---- per-lane state ----------------------------------------------------------------------------
  uint32_t st = ST_NEED_PIXEL;
  uint32_t x = 0, row = 0, s = 0;        // pixel (x, row from the top) and next sample index
  V3 col = mk(0.f, 0.f, 0.f);            // ordered sum of this pixel's sample colours
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  float time = 0.f;
  V3 accum = o, strength = o;
  uint32_t bounces = 0;
  SampleRng rng;
  rng.init(seed, 0, 0);
  uint32_t pc = 0, best_pc = NO_HIT;
  float best = F32_MAX;
  uint4 cur_lo = make_uint4(0, 0, 0, 0), cur_hi = make_uint4(0, 0, 0, OP_END);
  Counts cnt = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  // schedule statistics (COUNT variant only; wave-uniform): passes and active lanes per phase
  uint32_t n_box_it = 0, n_box_lanes = 0, n_sph_it = 0, n_sph_lanes = 0, n_regen = 0, n_shade_lanes = 0, n_gen_lanes = 0;
  unsigned long long t_shade = 0, t_gen = 0, t_box = 0, t_sph = 0, t_mark = 0;
#define RT_TICK() (COUNT ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)

  for (;;) {
    if (COUNT) {
      n_regen++;
      n_shade_lanes += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == ST_SHADE));
    }
    if (COUNT) t_mark = RT_TICK();
    // ================================ SHADE: finished traversals ================================
    if (st == ST_SHADE) {
      bool ended = true;               // path ends here?
      V3 result = mk(0.f, 0.f, 0.f);   // lib.rs:100: a miss returns black and discards accum
      if (best_pc != NO_HIT) {
        if (COUNT) cnt.shaded++;
        // rebuild the hit record (object.rs:100-106 + Translate/FlipNormals) from the winning record
        const uint4 plo = RT_FETCH_LO(best_pc), phi = RT_FETCH_HI(best_pc);
        V3 off = mk(u2f(plo.x), u2f(plo.y), u2f(plo.z));
        V3 lo_o = o;
        if (phi.w & F_TRANSLATE) lo_o = vsub(o, off);
        V3 hp = vadd(lo_o, smul(best, d));
        V3 hn = sdiv(hp, u2f(plo.w));
        if (phi.w & F_TRANSLATE) hp = vadd(hp, off);
        if (phi.w & F_FLIP) hn = vneg(hn);
        const uint4 mlo = RT_FETCH_MAT(2u * phi.z), mhi = RT_FETCH_MAT(2u * phi.z + 1u);
        const uint32_t kind = mhi.w & 0xffu;
        const float param = u2f(mlo.w);
        const V3 mcol = mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
        V3 emitted = mk(0.f, 0.f, 0.f);
        if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, mcol);  // material.rs:120-128
        accum = vadd(accum, vmul(strength, emitted));                // lib.rs:76
        V3 nd = mk(0.f, 0.f, 0.f), att = mcol;
        bool scattered = true;
        // Lambertian, Metal and Isotropic each draw exactly one in_unit_sphere, before any other draw of
        // this event (reflect() consumes no randomness): ONE rejection loop serves all three.
        V3 rs = mk(0.f, 0.f, 0.f);
        if (kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC) rs = in_unit_sphere(rng);
        if (kind == MAT_LAMBERTIAN) {  // material.rs:57-65
          V3 target = vadd(vadd(hp, hn), rs);
          nd = vsub(target, hp);
        } else if (kind == MAT_METAL) {  // material.rs:66-80
          V3 refl = reflect(vunit(d), hn);
          nd = vadd(refl, smul(param, rs));
          scattered = vdot(nd, hn) > 0.f;
        } else if (kind == MAT_DIELECTRIC) {  // material.rs:81-107
          V3 outward;
          float ni_over_nt, cosine;
          float dn = vdot(d, hn);
          if (dn > 0.f) {
            outward = vneg(hn);
            ni_over_nt = param;
            cosine = param * dn / vlen(d);
          } else {
            outward = hn;
            ni_over_nt = 1.0f / param;
            cosine = -dn / vlen(d);
          }
          V3 uv = vunit(d);  // refract, vec3.rs:321-330
          float dt = vdot(uv, outward);
          float disc = 1.0f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
          bool refracted = disc > 0.f;
          if (refracted) {
            nd = vsub(smul(ni_over_nt, vsub(uv, smul(dt, outward))), smul(__builtin_sqrtf(disc), outward));
            refracted = rng.gen_f32() >= schlick(cosine, param);
          }
          if (!refracted) nd = reflect(d, hn);
          att = splat(1.f);
        } else if (kind == MAT_DIFFUSE_LIGHT) {
          scattered = false;
        } else {  // Isotropic
          nd = rs;
        }
        result = accum;  // both early returns of color() yield accum (lib.rs:90,94)
        if (scattered) {
          o = hp, d = nd;
          strength = vmul(strength, att);
          if (bounces != P.max_bounces) {  // lib.rs:93-97
            bounces += 1;
            ended = false;
          }
        }
      }
      if (ended) {
        col = vadd(col, result);  // ordered fold (vec3.rs:195-203)
        if (COUNT) total_draws += rng.draws;
        s++;
        if (s == P.ns) {
          V3 px = sdiv(col, (float)P.ns);  // lib.rs:374
          float* op = out + 3ull * ((size_t)row * P.nx + x);
          op[0] = px.x, op[1] = px.y, op[2] = px.z;
          st = ST_NEED_PIXEL;
        } else {
          st = ST_GEN;
        }
      } else {
        st = ST_TRAV;
        rng.set_event(bounces + 1u);  // event k = k-th hit_top + scatter
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
        pc = 0, best = F32_MAX, best_pc = NO_HIT;
        cur_lo = RT_FETCH_LO(0), cur_hi = RT_FETCH_HI(0);
        if (COUNT) cnt.rays++;
      }
    }
    if (COUNT) {
      unsigned long long now = RT_TICK();
      t_shade += now - t_mark, t_mark = now;
    }
    // ================================ NEED_PIXEL: pull work ====================================
    for (;;) {
      uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
      if (need == 0) break;
      uint32_t base = 0;
      if ((uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(need))
        base = atomicAdd(queue, (uint32_t)__builtin_popcountll(need));  // one atomic per wave per refill
      base = __builtin_amdgcn_readlane(base, __builtin_ctzll(need));
      if (st == ST_NEED_PIXEL) {
        uint32_t w = base + lane_rank(need);
        if (w >= total_work) {
          st = ST_DEAD;
        } else if (work_to_pixel(P, w, x, row)) {
          s = 0;
          col = mk(0.f, 0.f, 0.f);
          st = ST_GEN;
        }  // else: a tile pixel outside the image, ask again
      }
    }
    if (COUNT) n_gen_lanes += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == ST_GEN));
    // ================================ GEN: next sample's camera ray ==============================
    if (st == ST_GEN) {  // par_cast closure, lib.rs:366-371
      const uint32_t y = P.ny - 1u - row;
      rng.init(seed, y * P.nx + x, s);
      float u = ((float)x + rng.gen_f32()) / (float)P.nx;
      float v = ((float)y + rng.gen_f32()) / (float)P.ny;
      get_ray(cam, u, v, rng, o, d, time);
      accum = mk(0.f, 0.f, 0.f), strength = splat(1.f), bounces = 0;  // lib.rs:62-67
      rng.set_event(1u);
      st = ST_TRAV;
      inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      pc = 0, best = F32_MAX, best_pc = NO_HIT;
      cur_lo = RT_FETCH_LO(0), cur_hi = RT_FETCH_HI(0);
      if (COUNT) cnt.rays++;
    }
    if (COUNT) {
      unsigned long long now = RT_TICK();
      t_gen += now - t_mark, t_mark = now;
    }
    if (__builtin_amdgcn_ballot_w64(st != ST_DEAD) == 0) break;

    // ================================ TRAV: hit_top over the flat program =======================
    for (;;) {
      uint32_t op = (st == ST_TRAV) ? (cur_hi.w & 0xffu) : 0xffu;
      if (op == OP_END) {
        st = ST_SHADE;
        op = 0xffu;
      }
      const uint64_t m_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
      const uint64_t m_sph = __builtin_amdgcn_ballot_w64(op == OP_SPHERE);
      if ((m_box | m_sph) == 0) break;
      if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == ST_SHADE)) >= tune.regen_min) break;
      if (m_box != 0 && (uint32_t)__builtin_popcountll(m_sph) < tune.sphere_min) {
        // ---- box run: tight loop, schedule re-evaluated once `box_leave` lanes have left the BOX state
        if (COUNT) t_mark = RT_TICK();
        const uint32_t n0 = (uint32_t)__builtin_popcountll(m_box);
        const uint32_t floor_lanes = n0 > tune.box_leave ? n0 - tune.box_leave : 0u;
        uint32_t n_now;
        do {
          if (COUNT) n_box_it++;
          if (op == OP_BOX) {  // Aabb::hit, aabb.rs:16-27
            if (COUNT) cnt.aabb++;
            float t0x = (u2f(cur_lo.x) - o.x) * inv.x, t1x = (u2f(cur_lo.y) - o.x) * inv.x;
            float t0y = (u2f(cur_lo.z) - o.y) * inv.y, t1y = (u2f(cur_lo.w) - o.y) * inv.y;
            float t0z = (u2f(cur_hi.x) - o.z) * inv.z, t1z = (u2f(cur_hi.y) - o.z) * inv.z;
            float ax = inv.x < 0.f ? t1x : t0x, bx = inv.x < 0.f ? t0x : t1x;
            float ay = inv.y < 0.f ? t1y : t0y, by = inv.y < 0.f ? t0y : t1y;
            float az = inv.z < 0.f ? t1z : t0z, bz = inv.z < 0.f ? t0z : t1z;
            float start = rs_max(t_near, rs_max(rs_max(ax, ay), az));
            float end = rs_min(best, rs_min(rs_min(bx, by), bz));
            pc = (end > start) ? pc + 1u : cur_hi.z;
            cur_lo = RT_FETCH_LO(pc), cur_hi = RT_FETCH_HI(pc);
            op = cur_hi.w & 0xffu;
          }
          n_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_BOX));
          if (COUNT) n_box_lanes += n_now;
        } while (n_now > floor_lanes);
        if (COUNT) t_box += RT_TICK() - t_mark;
      } else {
        if (COUNT) n_sph_it++, n_sph_lanes += (uint32_t)__builtin_popcountll(m_sph), t_mark = RT_TICK();
        if (op == OP_SPHERE) {  // Sphere::hit, object.rs:84-111 (+ Translate :275)
          if (COUNT) cnt.prim++;
          V3 lo_o = o;
          if (cur_hi.w & F_TRANSLATE) lo_o = vsub(o, mk(u2f(cur_lo.x), u2f(cur_lo.y), u2f(cur_lo.z)));
          float t;
          if (sphere_hit_t(lo_o, d, u2f(cur_lo.w), t_near, best, t)) {
            best = t;
            best_pc = pc;
          }
          pc++;
          cur_lo = RT_FETCH_LO(pc), cur_hi = RT_FETCH_HI(pc);
        }
        if (COUNT) t_sph += RT_TICK() - t_mark;
      }
    }
  }
  if (COUNT) {
    atomicAdd(&counters[0], (unsigned long long)cnt.aabb);
    atomicAdd(&counters[1], (unsigned long long)cnt.prim);
    atomicAdd(&counters[2], (unsigned long long)cnt.shaded);
    atomicAdd(&counters[3], (unsigned long long)cnt.rays);
    atomicAdd(&counters[4], (unsigned long long)total_draws);
    if ((threadIdx.x & 63u) == 0) {
      unsigned long long* sched = counters + 8;
      atomicAdd(&sched[0], (unsigned long long)n_box_it), atomicAdd(&sched[1], (unsigned long long)n_box_lanes);
      atomicAdd(&sched[2], (unsigned long long)n_sph_it), atomicAdd(&sched[3], (unsigned long long)n_sph_lanes);
      atomicAdd(&sched[4], (unsigned long long)n_regen), atomicAdd(&sched[5], (unsigned long long)n_shade_lanes);
      atomicAdd(&sched[6], (unsigned long long)n_gen_lanes);
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], t_gen), atomicAdd(&counters[18], t_box),
          atomicAdd(&counters[19], t_sph);
    }
  }
#undef RT_FETCH_LO
#undef RT_FETCH_HI
#undef RT_FETCH_MAT
}

}  // namespace rtg
