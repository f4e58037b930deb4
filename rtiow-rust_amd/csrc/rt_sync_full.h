// rt_sync_full.h -- the full-feature kernel for LIST WORLDS WITHOUT A BVH (no BOX record in the flat program: the Cornell box of
// configs[0], volume_test, the smoke boxes ...).  Every ray of such a world executes the same records in the same order, so there
// is nothing for ray pools to compact: every lane owns one path from camera ray to booked sample, and the wave alternates between a
// SHADE / GEN phase (all 64 lanes) and a TRAVERSE phase in which the lanes walk the record list in lock-step.  No slots, no lists,
// no service: path state never leaves the registers and the kernel spills nothing (107-126 VGPRs, 0 SGPR spills).
// Cornell 300x300x100: 4.4 -> 3.8 ms; smoke boxes 9.1 -> 7.6; volume_test 5.3 -> 4.6.  On programs WITH Bvhs it loses (book-2 48 ->
// 60-71 ms: rays leave a Bvh at different times and the box loop runs 7 lanes wide), so those stay on the pool kernel.
#pragma once
#include "rt_pool_full.h"

namespace rtg {

template <int PROG, bool TEX, bool COUNT, bool GENB = false>
__global__ __launch_bounds__(TEX ? RT_FULL_TEX_THREADS : 1024) void render_full_sync(DevScene sc, const LaunchConsts* __restrict__ lc, float* __restrict__ out,
                                                        uint32_t total_work, uint32_t* __restrict__ queue,
                                                        unsigned long long* counters, PoolTuning tune,
                                                        float* __restrict__ g_stack, uint32_t window) {
  constexpr uint32_t FEAT = FEAT_XFORM | FEAT_MEDIUM | FEAT_RECT | (TEX ? FEAT_TEXTURE : 0u);
  extern __shared__ uint4 s_mem[];
  constexpr uint32_t OP_SLOW_LAST = (uint32_t)OP_SEG;  // SPHERE .. SEG (this kernel steps over OP_SEG: it hoists nothing)
  constexpr uint32_t STACK_LEVELS = GENB ? 2 * MAX_XFORM_DEPTH : MAX_XFORM_DEPTH;
  constexpr bool USE_LDS = PROG != 0;
  const uint32_t win_bytes = RSZ * window;
  if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < window; i += blockDim.x) {
      uint4 h = sc.hi[i];
      if ((h.w & 0xffu) == OP_BOX) h.z *= RSZ;
      if ((h.w & 0xffu) == OP_MEDIUM) h.x *= RSZ;
      s_mem[2u * i] = sc.lo[i];
      s_mem[2u * i + 1u] = h;
    }
  }
  const char* s_bytes = reinterpret_cast<const char*>(s_mem);
#define RT_IN_LDS(pc_) (PROG == 1 || (PROG == 2 && (pc_) < win_bytes))
#define RT_FETCH_LO(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + (pc_)) : sc.lo[(pc_) >> 5])
#define RT_FETCH_HI(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + (pc_) + 16u) : fetch_hi_global(sc, (pc_) >> 5))
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const size_t gwave = (size_t)blockIdx.x * n_waves + wave;
  float* stack = g_stack + gwave * (STACK_LEVELS * 6 * 64);  // [level][component][lane]
  __syncthreads();

  const float t_near = load_const(&lc->P.t_near);
  uint32_t w_next = 0, w_end = 0, w_chunk = 0, w_delta = 0;
  bool w_lpt_ready = false;
  bool exhausted = false;

  // ---- per-lane state: ONE path ---------------------------------------------------------------------
  uint32_t st = ST_NEED_PIXEL;
  uint32_t px = 0, prow = 0, ps = 0, bounces = 0;  // pixel, sample, bounce count of the path (lib.rs:62-67)
  V3 strength = mk(0.f, 0.f, 0.f);
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  float time = 0.f, best = F32_MAX;
  uint32_t pc = 0;
  uint4 cur_lo = make_uint4(0, 0, 0, 0), cur_hi = make_uint4(0, 0, 0, OP_END);
  V3 hp = o, hn = o;
  uint32_t hmat = NO_HIT;
  uint32_t depth = 0, tag = 0, nhits = 0, root_hits = 0, ev_draws = 0;
  V3 sv_o = mk(0.f, 0.f, 0.f), sv_d = sv_o;  // level 0 of the transform stack (rt_full_ops.inc)
  uint32_t r_pixel = 0, r_sample = 0, r_event = 0;
  float seg_t = F32_MAX;  // (OP_SEG of rt_full_ops.inc: compiled out here, RT_HOIST = 0)
  uint32_t seg_pc = 0;
  uint32_t bmode = 0;
  float t_lo = t_near, b_saved = 0.f, b_t1 = 0.f;
  Counts cnt = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  uint32_t* tr_out = nullptr;
  uint32_t tr_draws = 0, tr_a0 = 0, tr_p0 = 0;  // per-sample trace accumulators of the lane's path
  if (COUNT) tr_out = reinterpret_cast<uint32_t*>(counters[30]);
  uint32_t n_box_it = 0, n_box_lanes = 0, n_slow_it = 0, n_slow_lanes = 0, n_shade = 0, n_shade_lanes = 0, n_refill = 0;
  unsigned long long t_shade = 0, t_box = 0, t_slow = 0, t_mark = 0;

#define RT_DEFER_HIT 0  // this kernel keeps the hit record in registers from the hit to its SHADE phase
#define RT_HOIST 0
#define RT_REG_STACK0 0  // measured: six more live registers cost this kernel 7 % on sphere lists, Cornell's wrappers gain nothing
#define RT_SAME_KIND_RUN 1  // consecutive SPHERE / RECT records in one go: simple_light (200 spheres) 16.0 -> 9.4 ms, Cornell 3.7 -> 3.6, smoke boxes 7.5 -> 7.15
#include "rt_full_ops.inc"
#undef RT_SAME_KIND_RUN
#undef RT_REG_STACK0
#undef RT_HOIST
#undef RT_DEFER_HIT

  for (;;) {
    // ============================== SHADE / GEN (every lane, its own path) ==========================
    {
      __builtin_amdgcn_s_setprio(RT_FULL_SERVICE_PRIO);  // (Cornell 3.37 -> 3.21 ms, smoke boxes 6.7 -> 6.35, volume_test 4.15 -> 4.0: r04s_priority_sync_ab.txt)
      if (COUNT) t_mark = RT_TICK();
      const DevParams P = load_const(&lc->P);
      const ChunkMode cm = load_const(&lc->cm);
      const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
      if (COUNT) n_shade++, n_shade_lanes += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == ST_SHADE));
      bool lpt_on = false;
      if (st == ST_SHADE) {  // color() loop body, lib.rs:73-97, for the ray that just finished (d is the original ray's: depth 0)
        const uint32_t hm = hmat == NO_HIT ? NO_HIT : (hmat & 0x7fffffffu);
        const V3 p = hp;
        uint4 mlo = make_uint4(0, 0, 0, 0), mhi = make_uint4(0, 0, 0, 0);
        V3 texval = mk(0.f, 0.f, 0.f);
        if (hm != NO_HIT) {
          mlo = sc.mat[2 * hm], mhi = sc.mat[2 * hm + 1];
          texval = material_texture<FEAT>(sc, mlo, mhi, p);
        }
        const V3 sd = d;
        SampleRng rng;
        rng.init(seed, (P.ny - 1u - prow) * P.nx + px, ps);
        rng.set_event(bounces + 1u);
        rng.seek(ev_draws);
        if (COUNT) tr_draws += ev_draws, total_draws += 0u;
        bool ended = true;
        V3 result = mk(0.f, 0.f, 0.f);
        if (hm != NO_HIT) {
          if (COUNT) cnt.shaded++;
          const V3 n = hn;
          const uint32_t kind = mhi.w & 0xffu;
          const float param = u2f(mlo.w);
          V3 emitted = mk(0.f, 0.f, 0.f);
          if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, texval);
          V3 accum = vadd(mk(0.f, 0.f, 0.f), vmul(strength, emitted));  // accum is +0 whenever it is read (rt_pool.h PoolField)
          V3 nd = mk(0.f, 0.f, 0.f), att = texval;
          bool scattered = true;
          V3 rs = mk(0.f, 0.f, 0.f);
          if (kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC) rs = in_unit_sphere(rng);
          float sd_len = 0.f;
          V3 sd_unit = sd;
          if (kind == MAT_METAL || kind == MAT_DIELECTRIC) sd_len = vlen(sd), sd_unit = sdiv(sd, sd_len);
          if (kind == MAT_LAMBERTIAN) {
            V3 target = vadd(vadd(p, n), rs);
            nd = vsub(target, p);
          } else if (kind == MAT_METAL) {
            V3 refl = reflect(sd_unit, n);
            nd = vadd(refl, smul(param, rs));
            att = mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
            scattered = vdot(nd, n) > 0.f;
          } else if (kind == MAT_DIELECTRIC) {
            V3 outward;
            float ni_over_nt, cosine;
            float dn = vdot(sd, n);
            if (dn > 0.f) {
              outward = vneg(n);
              ni_over_nt = param;
              cosine = param * dn / sd_len;
            } else {
              outward = n;
              ni_over_nt = 1.0f / param;
              cosine = -dn / sd_len;
            }
            V3 uv = sd_unit;
            float dt = vdot(uv, outward);
            float disc = 1.0f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
            bool refracted = disc > 0.f;
            if (refracted) {
              nd = vsub(smul(ni_over_nt, vsub(uv, smul(dt, outward))), smul(__builtin_sqrtf(disc), outward));
              refracted = rng.gen_f32() >= schlick(cosine, param);
            }
            if (!refracted) nd = reflect(sd, n);
            att = splat(1.f);
          } else if (kind == MAT_DIFFUSE_LIGHT) {
            scattered = false;
          } else {  // Isotropic
            nd = rs;
          }
          result = accum;
          if (scattered) {
            o = p, d = nd;  // time is carried over by every material
            strength = vmul(strength, att);
            if (bounces != P.max_bounces) {
              bounces += 1;
              ended = false;
              lpt_on = ps < cm.lpt_samples && bounces == cm.lpt_deep;
            }
          }
        }
        if (COUNT) total_draws += rng.draws, tr_draws += rng.draws;
        if (ended) {
          const size_t at = (size_t)ps * cm.pix_work + pixel_to_work(P, load_const(&lc->pm), px, prow);
          RT_SCRATCH_STORE(cm.scratch + 3ull * at, result);
          if (COUNT && tr_out) {
            uint32_t* tp = tr_out + 4ull * at;
            tp[0] = bounces, tp[1] = tr_draws, tp[2] = cnt.aabb - tr_a0, tp[3] = cnt.prim - tr_p0;
          }
          ps++;
          st = (ps == P.ns || ps % cm.chunk == 0u) ? ST_NEED_PIXEL : ST_GEN;
        } else {
          st = ST_TRAV;
        }
      }
      if (cm.lpt_samples) lpt_count(cm, lpt_on, lpt_on ? pixel_to_work(P, load_const(&lc->pm), px, prow) >> 8 : 0u);
      for (;;) {  // next work item (rt_pool.h)
        const uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
        if (need == 0) break;
        if (w_next == w_end && !exhausted) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(queue, cm.work_block);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= total_work) {
            exhausted = true;
          } else {
            w_chunk = lpt_reservation(cm, base, lane, w_delta, w_lpt_ready);
            w_next = base;
            w_end = (total_work - base < cm.work_block) ? total_work : base + cm.work_block;
          }
        }
        const uint32_t avail = w_end - w_next;
        if (st == ST_NEED_PIXEL) {
          const uint32_t r = lane_rank(need);
          if (r < avail) {
            const uint32_t w = w_next + r + w_delta;
            const uint32_t first = cm.s_begin + w_chunk * cm.chunk;
            if (work_to_pixel(P, load_const(&lc->pm), w, px, prow) && first < P.ns) {
              ps = first;
              st = ST_GEN;
            }
          } else if (exhausted) {
            st = ST_DEAD;
          }
        }
        const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
        w_next += n_need < avail ? n_need : avail;
      }
      if (st == ST_GEN) {  // par_cast closure, lib.rs:366-371 (event 0)
        const DevCamera cam = load_const(&lc->cam);
        const uint32_t y = P.ny - 1u - prow;
        SampleRng rng;
        rng.init(seed, y * P.nx + px, ps);
        float u = ((float)px + rng.gen_f32()) / (float)P.nx;
        float v = ((float)y + rng.gen_f32()) / (float)P.ny;
        get_ray(cam, u, v, rng, o, d, time);
        strength = splat(1.f), bounces = 0;
        if (COUNT) total_draws += rng.draws, tr_draws = rng.draws, tr_a0 = cnt.aabb, tr_p0 = cnt.prim;
        st = ST_TRAV;
      }
      if (st == ST_TRAV) {  // World::hit_top for the new ray
        r_pixel = (P.ny - 1u - prow) * P.nx + px, r_sample = ps, r_event = bounces + 1u;
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
        pc = 0, best = F32_MAX, hmat = NO_HIT;
        depth = 0, tag = 0, nhits = 0, root_hits = 0, ev_draws = 0;
        bmode = 0, t_lo = t_near;
        cur_lo = RT_FETCH_LO(0), cur_hi = RT_FETCH_HI(0);
        if (COUNT) cnt.rays++;
      }
      if (COUNT) t_shade += RT_TICK() - t_mark;
    }
    if (__builtin_amdgcn_ballot_w64(st == ST_TRAV) == 0) break;  // every lane retired
    // ============================== TRAVERSE: until the last of the wave's rays reaches END ==========
    const bool have_ray = st == ST_TRAV;
    uint32_t op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    for (;;) {
#define RT_R_PIXEL r_pixel
#define RT_R_EVENT r_event
#define RT_PHASE_PRIO 1  // wave priorities per phase (rt_pool_full.h RT_FULL_SLOW_PRIO): slow passes 3, box runs 1, the SHADE / GEN phase 0
#include "rt_full_traverse.inc"
#undef RT_PHASE_PRIO
#undef RT_R_PIXEL
#undef RT_R_EVENT
    op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    if (__builtin_amdgcn_ballot_w64(op >= OP_BOX && op <= OP_SLOW_LAST) == 0) break;
    }
    if (have_ray) st = ST_SHADE;
  }
  if (COUNT) {
    atomicAdd(&counters[0], (unsigned long long)cnt.aabb);
    atomicAdd(&counters[1], (unsigned long long)cnt.prim);
    atomicAdd(&counters[2], (unsigned long long)cnt.shaded);
    atomicAdd(&counters[3], (unsigned long long)cnt.rays);
    atomicAdd(&counters[4], (unsigned long long)total_draws);
    if (lane == 0) {
      unsigned long long* sched = counters + 8;
      atomicAdd(&sched[0], (unsigned long long)n_box_it), atomicAdd(&sched[1], (unsigned long long)n_box_lanes);
      atomicAdd(&sched[2], (unsigned long long)n_slow_it), atomicAdd(&sched[3], (unsigned long long)n_slow_lanes);
      atomicAdd(&sched[4], (unsigned long long)n_shade), atomicAdd(&sched[5], (unsigned long long)n_shade_lanes);
      atomicAdd(&sched[6], (unsigned long long)n_refill);
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], 0ull), atomicAdd(&counters[18], t_box), atomicAdd(&counters[19], t_slow);
    }
  }
#undef RT_IN_LDS
#undef RT_FETCH_LO
#undef RT_FETCH_HI
}

}  // namespace rtg
