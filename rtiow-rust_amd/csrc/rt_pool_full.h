// rt_pool_full.h -- the ray-pool schedule of rt_pool.h for EVERY feature of the hot path: Rect,
// PUSH/POP transform wrappers (Translate / RotateY / Scale / LinearMove / FlipNormals over subtrees),
// ConstantMedium (RNG draws during traversal), checker / Perlin textures, Isotropic.
//
// Differences from the lean kernel:
//  * the hit record (p, normal, material) is built at hit time and carried in registers, because it
//    has to travel back through the enclosing POPs (object.rs:279-282,365-369); at END it is written
//    to the path slot (7 dwords) instead of (best, best_pc);
//  * every non-BOX record (SPHERE, RECT, PUSH, POP, MEDIUM) is a "slow op": lanes park on it and a slow
//    pass executes one record per parked lane once enough lanes wait (or no BOX lane is left);
//  * the transform stack (<= 4 saved rays) lives in a per-wave, lane-interleaved global scratch;
//  * a medium draws from the event's RNG stream by index (event_draw), and the number of draws made
//    during traversal travels with the path so that Material::scatter continues the stream where
//    traversal left it (draw order of SURVEY 8a);
//  * always one sample per work item + ordered fold (the host falls back to render_kernel when the
//    sample scratch would not fit).
#pragma once
#include "rt_pool.h"

namespace rtg {

#ifndef RT_FULL_POOL_SLOTS
#define RT_FULL_POOL_SLOTS 160  // 128: gather points and slow passes run short of waiting lanes (book-2 +15 %); 192: +2.5 %
#endif
constexpr uint32_t FPOOL = RT_FULL_POOL_SLOTS;  // path slots per wave of the full-feature kernel (21 dwords each)
constexpr uint32_t FPOOL_FIELDS = 21;
enum FullPoolField : uint32_t {
  FF_O = 0, FF_D = 3, FF_TIME = 6, FF_HITMAT = 7, FF_P = 8, FF_N = 11, FF_STRENGTH = 14, FF_BOUNCES = 17, FF_SAMPLE = 18, FF_XY = 19,
  FF_EVDRAWS = 20,
};

// LDS = the first `window` program records (all of them when the program fits, 0 = none) + the lists
inline size_t full_pool_lds_bytes(uint32_t window, uint32_t waves) {
  return (size_t)window * 32 + (((size_t)waves * FPOOL * 3 * 2 + 15) & ~(size_t)15);  // T-, S- and X-list (u16 slot ids)
}

// draw `idx` (0-based) of the stream (seed, pixel, sample, event): word idx%4 of Philox block idx/4
__device__ __attribute__((noinline)) float event_draw_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t event, uint32_t idx) {
  SampleRng r;
  r.init(seed, pixel, sample);
  r.set_event(event);
  r.blk = idx >> 2;
  r.refill();
  uint32_t w = idx & 3u;
  uint32_t u = w == 0 ? r.b0 : (w == 1 ? r.b1 : (w == 2 ? r.b2 : r.b3));
  return (float)(u >> 8) * (1.0f / 16777216.0f);
}

// PROG: 0 = program fetched from global memory (L1/L2), 1 = whole program staged in LDS, 2 = an LDS
// window over the first `window` records (depth-first order, so it holds whole leading subtrees: book-2's
// 199 KB program keeps its floor Bvh and most top-level objects in LDS) and global memory for the rest.
#ifndef RT_FULL_TEX_THREADS
#define RT_FULL_TEX_THREADS 1024  // the textured variant wants ~142 VGPRs; capped at 128 it spills ~25 of them to scratch but
                                  // runs 16 instead of 12 waves per CU: measured 4 % faster on book-2 (768 = no spills)
#endif
// GENB: the scene holds a ConstantMedium whose boundary is an object graph (F_GENERAL_BOUNDARY, e.g. the book's smoke
// boxes: ConstantMedium<Translate<RotateY<And<...>>>>, object.rs:533-575).  Its two boundary queries
// (`boundary.hit(f32::MIN..f32::MAX)`, then `boundary.hit(t1 + 0.0001..f32::MAX)`, object.rs:551-552) run THROUGH THE SAME
// WALK: at such a MEDIUM record the lane enters the boundary's own record stream in "boundary mode" (bmode 1, then 2) --
// the range (t_lo, best) becomes the query's, hits only shrink `best` (no hit record), the main walk's `best` waits in
// b_saved -- and the stream's closing OP_BEND record finishes the query: restart for query 2, or compute the medium's hit
// and resume the main walk behind the stream.  The boundary's BOX records run in the box loop and its primitives in the
// slow passes like any others, so a complex boundary is scheduled as well as the rest of the scene.  A template variant:
// 5 more VGPRs, paid only by scenes that need it.
template <int PROG, bool TEX, bool COUNT, bool GENB = false>
__global__ __launch_bounds__(TEX ? RT_FULL_TEX_THREADS : 1024) void render_full_pool(DevScene sc, const LaunchConsts* __restrict__ lc, float* __restrict__ out,
                                                        uint32_t total_work, uint32_t* __restrict__ queue,
                                                        unsigned long long* counters, PoolTuning tune,
                                                        uint32_t* __restrict__ g_slots, float* __restrict__ g_stack,
                                                        uint32_t window) {
  // TEX = the scene references a checker / Perlin texture: only then is texture_eval (and its register
  // footprint) compiled in
  constexpr uint32_t FEAT = FEAT_XFORM | FEAT_MEDIUM | FEAT_RECT | (TEX ? FEAT_TEXTURE : 0u);
  extern __shared__ uint4 s_mem[];
  constexpr uint32_t OP_SLOW_LAST = GENB ? (uint32_t)OP_BEND : (uint32_t)OP_PRISM;  // records a slow pass executes
  constexpr uint32_t STACK_LEVELS = GENB ? 2 * MAX_XFORM_DEPTH : MAX_XFORM_DEPTH;   // a boundary stream nests below the medium's own wrappers
  constexpr bool USE_LDS = PROG != 0;
  const uint32_t staged = USE_LDS ? 2u * window : 0u;  // uint4 units; pc = 16 r, hi[] of the window at +16 window
  const uint32_t win_bytes = 16u * window;
  if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < window; i += blockDim.x) {
      uint4 h = sc.hi[i];
      if ((h.w & 0xffu) == OP_BOX) h.z *= 16u;
      if ((h.w & 0xffu) == OP_MEDIUM) h.x *= 16u;
      s_mem[i] = sc.lo[i];
      s_mem[window + i] = h;
    }
  }
  const char* s_bytes = reinterpret_cast<const char*>(s_mem);
#define RT_IN_LDS(pc_) (PROG == 1 || (PROG == 2 && (pc_) < win_bytes))
#define RT_FETCH_LO(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + (pc_)) : sc.lo[(pc_) >> 4])
#define RT_FETCH_HI(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + win_bytes + (pc_)) : fetch_hi_global(sc, (pc_) >> 4))
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const size_t gwave = (size_t)blockIdx.x * n_waves + wave;
  uint32_t* slot = g_slots + gwave * (FPOOL * FPOOL_FIELDS);
  float* slotf = reinterpret_cast<float*>(slot);
  float* stack = g_stack + gwave * (STACK_LEVELS * 6 * 64);  // [level][component][lane]
  uint16_t* tlist = reinterpret_cast<uint16_t*>(s_mem + staged) + wave * (3u * FPOOL);
  uint16_t* slist = tlist + FPOOL;  // finished rays whose material needs no texture lookup (and slots without a ray)
  uint16_t* xlist = slist + FPOOL;  // finished rays that hit a checker / Perlin textured material
#define SLOT_U(f_, j_) slot[(f_)*FPOOL + (j_)]
#define SLOT_F(f_, j_) slotf[(f_)*FPOOL + (j_)]
  for (uint32_t j = lane; j < FPOOL; j += 64u) {
    SLOT_U(FF_HITMAT, j) = SLOT_NEED_PIXEL;
    slist[j] = (uint16_t)j;
  }
  __syncthreads();

  const float t_near = load_const(&lc->P.t_near);
  uint32_t t_count = 0, s_count = FPOOL, x_count = 0, n_dead = 0;
  uint32_t w_next = 0, w_end = 0, w_chunk = 0, w_delta = 0;
  bool w_lpt_ready = false;
  bool exhausted = false;

  // ---- per-lane traversal state ---------------------------------------------------------------
  uint32_t my_slot = 0;
  bool have_ray = false;
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  float time = 0.f, best = F32_MAX;
  uint32_t pc = 0;
  uint4 cur_lo = make_uint4(0, 0, 0, 0), cur_hi = make_uint4(0, 0, 0, OP_END);
  V3 hp = o, hn = o;                 // hit record (object.rs:61-71), in the space of wrapper depth `tag`
  uint32_t hmat = NO_HIT;            // NO_HIT = None
  uint32_t depth = 0, tag = 0, nhits = 0, root_hits = 0, ev_draws = 0;
  uint32_t r_pixel = 0, r_sample = 0, r_event = 0;  // RNG stream of this ray's event (media)
  uint32_t bmode = 0;                                // GENB: 0 = main walk, 1 / 2 = inside a boundary stream, query 1 / 2
  float t_lo = t_near, b_saved = 0.f, b_t1 = 0.f;    // GENB: lower end of the current range; the main walk's best; query 1's t
  Counts cnt = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  uint32_t* tr_out = nullptr;  // per-sample trace of the instrumented variant (rt_pool.h)
  uint32_t* tr_slot = nullptr;
  uint32_t tr_a0 = 0, tr_p0 = 0;
  if (COUNT) {
    tr_out = reinterpret_cast<uint32_t*>(counters[30]);
    if (tr_out) tr_slot = reinterpret_cast<uint32_t*>(counters[31]) + gwave * (FPOOL * 3u);
  }
  uint32_t n_box_it = 0, n_box_lanes = 0, n_slow_it = 0, n_slow_lanes = 0, n_shade = 0, n_shade_lanes = 0, n_refill = 0;
  unsigned long long t_shade = 0, t_serv = 0, t_box = 0, t_slow = 0, t_mark = 0, t_mark2 = 0;  // COUNT: s_memtime shares

#include "rt_full_ops.inc"

  for (;;) {
    uint32_t op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    const uint64_t m_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    const uint64_t m_slow = __builtin_amdgcn_ballot_w64(op >= OP_SPHERE && op <= OP_SLOW_LAST);
    const uint32_t n_busy = (uint32_t)__builtin_popcountll(m_box | m_slow);
    // ============================== SERVICE ======================================================
    if (64u - n_busy >= tune.refill_min || n_busy == 0) {
      if (COUNT) t_mark = RT_TICK();
      {  // (1) finish
        const bool fin = have_ray && op == OP_END;
        // hmat bit 31 = the hit material reads a non-constant texture (F_TEXTURED of the winning record):
        // such hits are shaded in their own passes, so the Perlin / checker code is issued for 64 textured
        // hits at a time instead of in every pass that happens to hold one
        const bool to_x = TEX && fin && hmat != NO_HIT && (hmat >> 31) != 0u;
        const uint64_t m_fin = __builtin_amdgcn_ballot_w64(fin && !to_x), m_x = __builtin_amdgcn_ballot_w64(to_x);
        if (fin) {
          SLOT_U(FF_HITMAT, my_slot) = hmat == NO_HIT ? NO_HIT : (hmat & 0x7fffffffu);
          SLOT_F(FF_P, my_slot) = hp.x, SLOT_F(FF_P + 1, my_slot) = hp.y, SLOT_F(FF_P + 2, my_slot) = hp.z;
          SLOT_F(FF_N, my_slot) = hn.x, SLOT_F(FF_N + 1, my_slot) = hn.y, SLOT_F(FF_N + 2, my_slot) = hn.z;
          SLOT_U(FF_EVDRAWS, my_slot) = ev_draws;
          if (COUNT && tr_slot)
            tr_slot[my_slot] += ev_draws, tr_slot[FPOOL + my_slot] += cnt.aabb - tr_a0, tr_slot[2u * FPOOL + my_slot] += cnt.prim - tr_p0;
          if (to_x) xlist[x_count + lane_rank(m_x)] = (uint16_t)my_slot;
          else slist[s_count + lane_rank(m_fin)] = (uint16_t)my_slot;
          have_ray = false;
        }
        s_count += (uint32_t)__builtin_popcountll(m_fin);
        x_count += (uint32_t)__builtin_popcountll(m_x);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      // (2) shade
      auto shade_pass = [&](auto textured_tag, uint16_t* list, uint32_t& count) {
        constexpr bool TEXTURED = decltype(textured_tag)::value;
        constexpr uint32_t PASS_FEAT = TEXTURED ? FEAT : (FEAT & ~FEAT_TEXTURE);
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        const uint32_t take = count < 64u ? count : 64u;
        count -= take;
        if (COUNT) n_shade++, n_shade_lanes += take, t_mark2 = RT_TICK();
        uint32_t st = ST_DEAD, j = 0;
        V3 so = mk(0.f, 0.f, 0.f), sd = so, strength = so;
        // no accum field: it is +0 whenever it is read (rt_pool.h PoolField; the host only routes scenes
        // here whose path strength stays finite and non-negative)
        V3 accum = so;
        float stime = 0.f;
        uint32_t bounces = 0, s = 0, x = 0, row = 0;
        bool lpt_on = false;
        if (lane < take) {
          j = list[count + lane];
          const uint32_t hm = SLOT_U(FF_HITMAT, j);
          if (hm == SLOT_NEED_PIXEL) {
            st = ST_NEED_PIXEL;
          } else {
            // The texture value is fetched FIRST, while almost nothing of this pass is live: texture_eval
            // (Perlin turbulence / checker) is an out-of-line call and everything live across it adds to
            // the kernel's register count.
            const V3 p = mk(SLOT_F(FF_P, j), SLOT_F(FF_P + 1, j), SLOT_F(FF_P + 2, j));
            uint4 mlo = make_uint4(0, 0, 0, 0), mhi = make_uint4(0, 0, 0, 0);
            V3 texval = mk(0.f, 0.f, 0.f);
            if (hm != NO_HIT) {
              mlo = sc.mat[2 * hm], mhi = sc.mat[2 * hm + 1];
              texval = material_texture<PASS_FEAT>(sc, mlo, mhi, p);  // = albedo / emission colour for constant textures
            }
            so = mk(SLOT_F(FF_O, j), SLOT_F(FF_O + 1, j), SLOT_F(FF_O + 2, j));
            sd = mk(SLOT_F(FF_D, j), SLOT_F(FF_D + 1, j), SLOT_F(FF_D + 2, j));
            stime = SLOT_F(FF_TIME, j);
            strength = mk(SLOT_F(FF_STRENGTH, j), SLOT_F(FF_STRENGTH + 1, j), SLOT_F(FF_STRENGTH + 2, j));
            bounces = SLOT_U(FF_BOUNCES, j), s = SLOT_U(FF_SAMPLE, j);
            const uint32_t xy = SLOT_U(FF_XY, j);
            x = xy & 0xffffu, row = xy >> 16;
            // ---------------- color() loop body, lib.rs:73-97 ----------------
            SampleRng rng;
            rng.init(seed, (P.ny - 1u - row) * P.nx + x, s);
            rng.set_event(bounces + 1u);
            rng.seek(SLOT_U(FF_EVDRAWS, j));  // continue after the medium draws of this event's traversal
            bool ended = true;
            V3 result = mk(0.f, 0.f, 0.f);
            if (hm != NO_HIT) {
              if (COUNT) cnt.shaded++;
              const V3 n = mk(SLOT_F(FF_N, j), SLOT_F(FF_N + 1, j), SLOT_F(FF_N + 2, j));
              const uint32_t kind = mhi.w & 0xffu;
              const float param = u2f(mlo.w);
              V3 emitted = mk(0.f, 0.f, 0.f);
              if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, texval);  // material.rs:120-128
              accum = vadd(accum, vmul(strength, emitted));
              V3 nd = mk(0.f, 0.f, 0.f), att = texval;  // Lambertian / Isotropic: albedo(p)
              bool scattered = true;
              V3 rs = mk(0.f, 0.f, 0.f);
              if (kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC) rs = in_unit_sphere(rng);
              float sd_len = 0.f;  // |d| and unit(d) once for the Metal and the Dielectric lanes (rt_pool.h)
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
                so = p, sd = nd;  // time is carried over by every material
                strength = vmul(strength, att);
                if (bounces != P.max_bounces) {
                  bounces += 1;
                  ended = false;
                  lpt_on = s < cm.lpt_samples && bounces == cm.lpt_deep;  // phase 1 of the cost-ordered queue (rt_pool.h)
                }
              }
            }
            if (COUNT) total_draws += rng.draws;
            if (COUNT && tr_slot) tr_slot[j] += rng.draws;
            if (ended) {
              float* sp = cm.scratch + 3ull * ((size_t)s * cm.pix_work + pixel_to_work(P, x, row));
              RT_SCRATCH_STORE(sp, result);
              if (COUNT && tr_out) {
                uint32_t* tp = tr_out + 4ull * ((size_t)s * cm.pix_work + pixel_to_work(P, x, row));
                tp[0] = bounces, tp[1] = tr_slot[j], tp[2] = tr_slot[FPOOL + j], tp[3] = tr_slot[2u * FPOOL + j];
              }
              s++;
              st = (s == P.ns || s % cm.chunk == 0u) ? ST_NEED_PIXEL : ST_GEN;
            } else {
              st = ST_TRAV;
            }
          }
        }
        if (cm.lpt_samples) lpt_count(cm, lpt_on, lpt_on ? pixel_to_work(P, x, row) >> 8 : 0u);
        for (;;) {  // next work item (see rt_pool.h)
          const uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
          if (need == 0) break;
          if (w_next == w_end && !exhausted) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(queue, WORK_BLOCK);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total_work) {
              exhausted = true;
            } else {
              w_chunk = lpt_reservation(cm, base, lane, w_delta, w_lpt_ready);
              w_next = base;
              w_end = (total_work - base < WORK_BLOCK) ? total_work : base + WORK_BLOCK;
            }
          }
          const uint32_t avail = w_end - w_next;
          if (st == ST_NEED_PIXEL) {
            const uint32_t r = lane_rank(need);
            if (r < avail) {
              uint32_t w = w_next + r;
              w += w_delta;
              const uint32_t first = w_chunk * cm.chunk;
              if (work_to_pixel(P, w, x, row) && first < P.ns) {
                s = first;
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
          const uint32_t y = P.ny - 1u - row;
          SampleRng rng;
          rng.init(seed, y * P.nx + x, s);
          float u = ((float)x + rng.gen_f32()) / (float)P.nx;
          float v = ((float)y + rng.gen_f32()) / (float)P.ny;
          const DevCamera cam = load_const(&lc->cam);
          get_ray(cam, u, v, rng, so, sd, stime);
          accum = mk(0.f, 0.f, 0.f), strength = splat(1.f), bounces = 0;
          if (COUNT) total_draws += rng.draws;
          if (COUNT && tr_slot) tr_slot[j] = rng.draws, tr_slot[FPOOL + j] = 0u, tr_slot[2u * FPOOL + j] = 0u;
          st = ST_TRAV;
        }
        const bool live = st == ST_TRAV;
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live);
        if (live) {
          SLOT_F(FF_O, j) = so.x, SLOT_F(FF_O + 1, j) = so.y, SLOT_F(FF_O + 2, j) = so.z;
          SLOT_F(FF_D, j) = sd.x, SLOT_F(FF_D + 1, j) = sd.y, SLOT_F(FF_D + 2, j) = sd.z;
          SLOT_F(FF_TIME, j) = stime;
          SLOT_F(FF_STRENGTH, j) = strength.x, SLOT_F(FF_STRENGTH + 1, j) = strength.y, SLOT_F(FF_STRENGTH + 2, j) = strength.z;
          SLOT_U(FF_BOUNCES, j) = bounces, SLOT_U(FF_SAMPLE, j) = s;
          SLOT_U(FF_XY, j) = x | (row << 16);
          tlist[t_count + lane_rank(m_live)] = (uint16_t)j;
          if (COUNT) cnt.rays++;
        }
        t_count += (uint32_t)__builtin_popcountll(m_live);
        n_dead += take - (uint32_t)__builtin_popcountll(m_live);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_shade += RT_TICK() - t_mark2;
      };
      const bool starving = t_count == 0 && n_busy == 0;
      while (s_count >= 64u || (s_count > 0 && starving)) shade_pass(std::false_type{}, slist, s_count);
      if (TEX)
        while (x_count >= 64u || (x_count > 0 && starving && t_count == 0)) shade_pass(std::true_type{}, xlist, x_count);
      {  // (3) refill
        const uint64_t m_idle = __builtin_amdgcn_ballot_w64(!have_ray);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(m_idle);
        const uint32_t got = n_idle < t_count ? n_idle : t_count;
        if (got) {
          const uint32_t r = lane_rank(m_idle);
          if (!have_ray && r < got) {
            my_slot = tlist[t_count - 1u - r];
            o = mk(SLOT_F(FF_O, my_slot), SLOT_F(FF_O + 1, my_slot), SLOT_F(FF_O + 2, my_slot));
            d = mk(SLOT_F(FF_D, my_slot), SLOT_F(FF_D + 1, my_slot), SLOT_F(FF_D + 2, my_slot));
            time = SLOT_F(FF_TIME, my_slot);
            const uint32_t xy = SLOT_U(FF_XY, my_slot);
            r_pixel = (load_const(&lc->P.ny) - 1u - (xy >> 16)) * load_const(&lc->P.nx) + (xy & 0xffffu);
            r_sample = SLOT_U(FF_SAMPLE, my_slot);
            r_event = SLOT_U(FF_BOUNCES, my_slot) + 1u;
            inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
            pc = 0, best = F32_MAX, hmat = NO_HIT;
            depth = 0, tag = 0, nhits = 0, root_hits = 0, ev_draws = 0;
            bmode = 0, t_lo = t_near;
            if (COUNT) tr_a0 = cnt.aabb, tr_p0 = cnt.prim;
            cur_lo = RT_FETCH_LO(0), cur_hi = RT_FETCH_HI(0);
            have_ray = true;
          }
          t_count -= got;
          if (COUNT) n_refill++;
        }
      }
      if (COUNT) t_serv += RT_TICK() - t_mark;
      if (n_dead == FPOOL) break;
      if (__builtin_amdgcn_ballot_w64(have_ray) == 0) continue;
      op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    }
    // ============================== TRAVERSE ======================================================
    // box runs and slow passes alternate in this inner loop until a service is due (rt_pool.h)
    for (;;) {
#include "rt_full_traverse.inc"
    op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    const uint32_t busy = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op >= OP_BOX && op <= OP_SLOW_LAST));
    if (64u - busy >= tune.refill_min || busy == 0) break;
    }
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
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], t_serv - t_shade), atomicAdd(&counters[18], t_box),
          atomicAdd(&counters[19], t_slow);
    }
  }
#undef RT_IN_LDS
#undef RT_FETCH_LO
#undef RT_FETCH_HI
#undef SLOT_U
#undef SLOT_F
}

}  // namespace rtg
