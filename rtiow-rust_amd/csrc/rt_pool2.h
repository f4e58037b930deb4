// rt_pool2.h -- the SECOND full-feature ray-pool kernel ("pool 2"): the lean kernel's structure (rt_pool.h: home slots, u16 id
// lists in LDS, hits as (t, record pc)) for list worlds with Bvhs, media, textures and wrapped Bvhs -- book-2's shape (main.rs:161-319)
// -- with the WHOLE LIST LEVEL HOISTED to ray creation (flat_scene.h "the list level, hoisted"; SceneBuilder::flatten_pool2).
//
// What differs from render_full_pool (rt_pool_full.h), which stays the fallback for every other program shape:
//  * a path has a HOME SLOT: one 128-byte line of global memory (array of structs: a lane reads or writes its slot with four
//    16-byte accesses; the first kernel moved 15- and 16-dword records through four dense stacks, one dword per instruction).
//    The slot holds the ray, the path's sample / pixel / strength / bounce count, what the walk found (t, winning record pc, medium
//    draws made) and the path's SIDE RECORD: two dwords per list-level item + the first two words of the event's Philox block;
//  * four lists of u16 slot ids per wave in LDS -- T (to traverse) and E (ended: next sample / work item) share one array from
//    both ends, S and X (to shade without / with a texture lookup) another; a slot is in at most one list, so neither overflows;
//  * the walk only meets Bvh streams and OP_LIST records.  Everything a list-level object needs from the ray -- the closest
//    candidate of every run of plain primitives, the two boundary roots of every medium, the root-box distances of every wrapped
//    Bvh for the transformed ray, the event's first two random words -- is evaluated by the pass that CREATES the ray (60 lanes
//    wide, every lane on the same record) and committed by one OP_LIST record of ~60 instructions that waits for no company: no
//    gather point holds 26 of 64 lanes back (profiles/r05_experiments/r05a_book2_lane_census.txt);
//  * the walk carries no strength / sample / pixel / bounce count, no transform stack and no hit counters: the slot keeps the
//    original ray, a POP reloads it, media sit at list level only and draw from the hoisted words.
// Same arithmetic as every other interpreter of the flat program (operation order cited at the operations); results, counters and
// per-sample traces equal the oracle's.
#pragma once
#include "rt_pool_full.h"

namespace rtg {

#ifndef RT_P2_EXP
#define RT_P2_EXP 0  // cost probes (WRONG pictures): bit 0 / 1 / 2 / 3 = hoist_eval without its primitive runs / media / wrapped Bvhs / random words
#endif
#ifndef RT_P2_SERVICE_PRIO
#define RT_P2_SERVICE_PRIO 2  // (services at 2: 1.6 % faster than at the first kernel's 0 -- this kernel's waves wait more and issue less)
#define RT_P2_BOX_PRIO RT_FULL_BOX_PRIO
#define RT_P2_SLOW_PRIO RT_FULL_SLOW_PRIO
#endif
#ifndef RT_P2_PREFETCH
#define RT_P2_PREFETCH 0  // 1: a finishing lane asks for its slot's line again so that the pass that takes the slot next finds it in the L2 -- measured 3 % SLOWER (r06d)
#endif
#ifndef RT_P2_RELOAD_SIDE
#define RT_P2_RELOAD_SIDE 0  // 1: every lane re-reads its side record from the slot after a service (12 registers dead across the passes: 117 -> 107 VGPRs, and 1 % slower)
#endif
#ifndef RT_P2_POOL
#define RT_P2_POOL 224  // paths in flight per wave (64 in the lanes + the ones that wait for company in S, X, E and T)
#endif
constexpr uint32_t P2POOL = RT_P2_POOL;
constexpr uint32_t P2_SLOT_BYTES = 128;
// byte offsets inside a slot; every group starts on 16 bytes
enum P2SlotField : uint32_t {
  PS_O = 0, PS_TIME = 12,                       // o.xyz, time
  PS_D = 16, PS_SAMPLE = 28,                    // d.xyz, sample
  PS_BEST = 32, PS_HMAT = 36, PS_EV = 40, PS_BOUNCES = 44,  // what the walk found: t, record pc | face | textured; medium draws made | scatter tries << 28; bounces
  PS_STRENGTH = 48, PS_XY = 60,                 // strength.xyz, x | row << 16
  PS_SIDE = 64,                                 // P2_MAX_ITEMS x (a, b)
  PS_LN = 104,                                  // ln(u0), ln(u1): u = draws 0 and 1 of the ray's event (rand 0.6.5 f32 of words 0, 1 of Philox block 0) -- what a medium takes (object.rs:562)
  PS_BLK = 112,                                 // (spare: 16 bytes)
};
static_assert(PS_BLK + 16u <= P2_SLOT_BYTES && PS_SIDE + 8u * P2_MAX_ITEMS == PS_LN && PS_LN + 8u == PS_BLK, "slot layout");
// E-list entries carry in their two top bits what the camera pass has to do for the slot
constexpr uint32_t E_MISS = 0x0000u;  // the ray missed: book black (lib.rs:100), then the next sample of the work item / the next item
constexpr uint32_t E_NEXT = 0x4000u;  // a shade pass booked the sample: next sample of the same work item
constexpr uint32_t E_ITEM = 0x8000u;  // the slot wants a new work item (initial state; last sample of an item booked)
constexpr uint32_t E_ID = 0x3fffu;
static_assert(P2POOL <= E_ID, "slot ids are 14 bits");
constexpr float P2_NOT_CROSSED = __builtin_inff();  // side.a of a medium whose boundary the ray does not cross twice (a root is < f32::MAX)

// dynamic LDS of a workgroup: [program, 32 B per record][control words][P2Table][materials, when they fit][per wave: TE ids, SX ids]
constexpr uint32_t P2_TABLE_BYTES = 128;
static_assert(sizeof(P2Table) <= P2_TABLE_BYTES, "P2Table");
inline size_t pool2_lds_bytes(uint32_t n_prog, uint32_t n_mat_in_lds, uint32_t waves) {
  return (size_t)n_prog * 32 + DC_WORDS * 4 + P2_TABLE_BYTES + (size_t)n_mat_in_lds * 32 + (size_t)waves * P2POOL * 2 * sizeof(uint16_t);
}
// slot space of a workgroup: its waves' slots, then its hand-over buffer (drain-phase work sharing: DQ_CAP slot lines)
inline size_t pool2_slot_words(uint32_t waves) { return ((size_t)waves * P2POOL + DQ_CAP) * (P2_SLOT_BYTES / 4); }

// Schedule thresholds (options p2_*; none of them changes a result)
struct Pool2Tuning {
  uint32_t refill_min;  // idle lanes before the wave services (finish / shade / camera rays / refill)
  uint32_t box_leave;   // lanes leaving the BOX state before a box run re-evaluates the schedule
  uint32_t park_max;    // lanes parked below their kinds' thresholds before the fullest kind runs anyway
  uint32_t t_sphere, t_prism, t_list, t_push;  // lanes that must wait on a kind of record before a slow pass runs it (>= 1)
};

typedef uint32_t p2_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t p2_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t p2_u32x2 __attribute__((ext_vector_type(2)));
RT_DEV uint4 p2_record(uint32_t lds_addr) {  // 16 bytes at a 16-byte aligned absolute LDS address
  const p2_u32x4 v = RT_AS3(p2_u32x4, lds_addr);
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <bool TEX, bool COUNT>
__global__ __launch_bounds__(TEX ? RT_FULL_TEX_THREADS : 1024) void render_full_pool2(DevScene sc, const LaunchConsts* __restrict__ lc, float* __restrict__ out,
                                                         uint32_t total_work, uint32_t* __restrict__ queue,
                                                         unsigned long long* counters, Pool2Tuning tune,
                                                         uint32_t* __restrict__ g_slots, const P2Table* __restrict__ g_table) {
  constexpr uint32_t FEAT = FEAT_XFORM | FEAT_MEDIUM | FEAT_RECT | (TEX ? FEAT_TEXTURE : 0u);
  extern __shared__ uint4 s_mem[];
  const uint32_t n_prog = sc.n_prog;
  for (uint32_t i = threadIdx.x; i < n_prog; i += blockDim.x) {  // the whole program lives in LDS (else the launcher takes the first kernel)
    uint4 h = sc.hi[i];
    if ((h.w & 0xffu) == OP_BOX) h.z *= RSZ;  // skip pointers as byte offsets
    s_mem[2u * i] = sc.lo[i];
    s_mem[2u * i + 1u] = h;
  }
  uint32_t* ctl = reinterpret_cast<uint32_t*>(s_mem + 2u * n_prog);
  if (threadIdx.x < DC_WORDS) ctl[threadIdx.x] = 0u;
  uint32_t* s_table = ctl + DC_WORDS;  // P2Table, as words
  if (threadIdx.x < sizeof(P2Table) / 4u) s_table[threadIdx.x] = reinterpret_cast<const uint32_t*>(g_table)[threadIdx.x];
  const uint32_t mat_lds = load_const(&lc->mat_lds);
  if (mat_lds)
    for (uint32_t i = threadIdx.x; i < 2u * sc.n_mat; i += blockDim.x) s_mem[(mat_lds >> 4) + i] = sc.mat[i];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const size_t gwave = (size_t)blockIdx.x * n_waves + wave;
  uint16_t* te = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(s_mem) + load_const(&lc->p2_lists)) + (size_t)wave * (2u * P2POOL);
  uint16_t* sx = te + P2POOL;
  // T grows from te[0] upwards, E from te[P2POOL - 1] downwards; S from sx[0] upwards, X from sx[P2POOL - 1] downwards
  for (uint32_t j = lane; j < P2POOL; j += 64u) te[P2POOL - 1u - j] = (uint16_t)(j | E_ITEM);
  const size_t wg_words = ((size_t)n_waves * P2POOL + DQ_CAP) * (P2_SLOT_BYTES / 4u);
  uint32_t* const wg_slots = g_slots + (size_t)blockIdx.x * wg_words;
  const QueueRsrc qr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg_slots + (size_t)wave * (P2POOL * (P2_SLOT_BYTES / 4u))), 0, P2POOL * P2_SLOT_BYTES, 0x00020000);
  bool hungry = false;  // this wave has run dry and is counted in ctl[DC_HUNGRY] (wave-uniform; rt_pool_full.h DrainCtl)
#ifdef RT_DRAIN_WATCHDOG
  uint32_t drain_polls = 0;
#endif
  __syncthreads();

  // A record's byte offset IS its LDS address: the kernel has no static __shared__ data, so the dynamic segment starts at 0 (checked
  // here) -- as `s_bytes + pc` every record fetch carried a `v_add_u32 v, 0, pc` for the segment's base.
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)s_mem != 0u) __builtin_trap();
#define P2_LO(pc_) p2_record(pc_)
#define P2_HI(pc_) p2_record((pc_) + 16u)
#define SL_OFF(id_, f_) ((id_) * P2_SLOT_BYTES + (f_))
#define SL_LD1(id_, f_) __builtin_amdgcn_raw_buffer_load_b32(qr, SL_OFF(id_, f_), 0, 0)
#define SL_LD2(id_, f_) __builtin_amdgcn_raw_buffer_load_b64(qr, SL_OFF(id_, f_), 0, 0)
#define SL_LD3(id_, f_) __builtin_amdgcn_raw_buffer_load_b96(qr, SL_OFF(id_, f_), 0, 0)
#define SL_LD4(id_, f_) __builtin_amdgcn_raw_buffer_load_b128(qr, SL_OFF(id_, f_), 0, 0)
#define SL_ST1(id_, f_, v_) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(v_), qr, SL_OFF(id_, f_), 0, 0)
#define SL_ST2(id_, f_, a_, b_) __builtin_amdgcn_raw_buffer_store_b64(p2_u32x2{(a_), (b_)}, qr, SL_OFF(id_, f_), 0, 0)
#define SL_ST3(id_, f_, a_, b_, c_) __builtin_amdgcn_raw_buffer_store_b96(p2_u32x3{(a_), (b_), (c_)}, qr, SL_OFF(id_, f_), 0, 0)
#define SL_ST4(id_, f_, a_, b_, c_, d_) __builtin_amdgcn_raw_buffer_store_b128(p2_u32x4{(a_), (b_), (c_), (d_)}, qr, SL_OFF(id_, f_), 0, 0)
  // side (a, b) of item k, k wave-uniform: the soffset operand carries 8 k
#define SL_ST_SIDE(id_, k_, a_, b_) __builtin_amdgcn_raw_buffer_store_b64(p2_u32x2{(a_), (b_)}, qr, (id_) * P2_SLOT_BYTES, PS_SIDE + 8u * (k_), 0)

  const float t_near = load_const(&lc->P.t_near);
  uint32_t t_count = 0, e_count = P2POOL, s_count = 0, x_count = 0, n_dead = 0;  // wave-uniform list sizes / retired slots
  uint32_t w_next = 0, w_end = 0, w_chunk = 0, w_delta = 0;
  bool w_lpt_ready = false;
  bool exhausted = false;
  const unsigned long long t_start = RT_TICK();
  unsigned long long t_exhausted = 0;

  // ---- per-lane traversal state ---------------------------------------------------------------
  bool have_ray = false;
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  float time = 0.f, best = F32_MAX;
  uint32_t pc = 0, hmat = NO_HIT;
  uint32_t ev = 0;  // medium draws the walk has made (low byte) | the lane's slot << 8: one register (a thirteenth live across the passes was spilled)
#define my_slot (ev >> 8)
  uint4 cur_lo = make_uint4(0, 0, 0, 0), cur_hi = make_uint4(0, 0, 0, OP_END);
  p2_u32x4 sdA = {0, 0, 0, 0}, sdB = sdA, sdC = sdA;  // the side record: items 0-1, 2-3, 4 + (ln u0, ln u1)
  [[maybe_unused]] uint32_t prefetch_sink = 0;  // (keeps the prefetch loads alive: stored at the end under a condition that never holds)
  Counts cnt = {0, 0, 0, 0};
  // tests a pass makes for the rays it CREATES: kept apart from cnt -- the lane may hold a ray of its own at that moment, whose
  // per-sample trace is the difference of cnt between its refill and its finish
  Counts hoisted = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  uint32_t* tr_out = nullptr;   // per-sample trace (instrumented variant; rt_pool.h): counters[30] = the table, counters[31] = per-slot accumulators
  uint32_t* tr_slot = nullptr;  // rows draws | aabb | prim of this wave's slots
  uint32_t tr_a0 = 0, tr_p0 = 0;
  if (COUNT) {
    tr_out = reinterpret_cast<uint32_t*>(counters[30]);
    if (tr_out) tr_slot = reinterpret_cast<uint32_t*>(counters[31]) + gwave * (P2POOL * 3u);
  }
  uint32_t n_box_it = 0, n_box_lanes = 0, n_slow_it = 0, n_slow_lanes = 0, n_shade = 0, n_shade_lanes = 0, n_refill = 0;
  uint32_t n_gen = 0, n_gen_lanes = 0;
  unsigned long long t_gen = 0, t_shade = 0, t_serv = 0, t_box = 0, t_slow = 0, t_refill = 0, t_fin = 0, n_serv = 0, t_mark = 0, t_mark2 = 0;

  // One plain primitive record `q` (SPHERE incl. F_MOVE, RECT, PRISM) against (t_lo .. t_hi): object.rs:84-111, 185-218, 420-473.
  // Returns the hit's t and `tag` = face | textured bit (what a hit's pc carries); n_tests += the reference's primitive tests.
  // q is wave-uniform where the passes call it (scalar branches), per lane in the replay below.
  auto prim_test = [&](auto uniform_tag, const uint32_t q, const V3 ro, const V3 rd, const float rtime, const float t_hi, float& t, uint32_t& tag,
                       uint32_t& n_tests, uint32_t& q_next) -> bool {
    const uint4 q_lo = P2_LO(q), q_hi = P2_HI(q);
    const uint32_t w = decltype(uniform_tag)::value ? __builtin_amdgcn_readfirstlane(q_hi.w) : q_hi.w, q_op = w & 0xffu;
    tag = (w >> 16) & 8u;
    q_next = q + RSZ;
    if (q_op == OP_SPHERE) {
      V3 lo_o = ro;
      if (w & F_TRANSLATE) lo_o = vsub(ro, mk(u2f(q_lo.x), u2f(q_lo.y), u2f(q_lo.z)));
      if (w & F_MOVE) {
        const uint4 mv = P2_LO(q + RSZ);
        lo_o = vsub(lo_o, smul(rtime, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
        q_next = q + 2u * RSZ;
      }
      n_tests += 1u;
      return sphere_hit_t(lo_o, rd, u2f(q_lo.w), t_near, t_hi, t);
    }
    if (q_op == OP_RECT) {
      n_tests += 1u;
      return rect_hit_t(ro, rd, (w >> F_AXIS_SHIFT) & 3u, u2f(q_lo.x), u2f(q_lo.y), u2f(q_lo.z), u2f(q_lo.w), u2f(q_hi.x), t_near, t_hi, t);
    }
    uint32_t face = 0;
    n_tests += 6u;
    const bool h = prism_hit_t(q_lo, q_hi, ro, rd, t_near, t_hi, t, face) != 0u;
    tag |= face;
    return h;
  };

  // PUSH of a wrapper (object.rs:275-278, 357-361, 309-313, 505-508), on the ray in (ro, rd)
  auto push_xform = [&](const uint4 x_lo, const uint4 x_hi, const float rtime, V3& ro, V3& rd) {
    const uint32_t kind = (x_hi.w >> F_KIND_SHIFT) & 7u;
    const V3 a = mk(u2f(x_lo.x), u2f(x_lo.y), u2f(x_lo.z));
    if (x_hi.w & F_PRE_TRANSLATE) ro = vsub(ro, mk(u2f(x_lo.w), u2f(x_hi.x), u2f(x_hi.y)));
    if (kind == XF_TRANSLATE) ro = vsub(ro, a);
    else if (kind == XF_ROTATE_Y) ro = rot_y(ro, -a.x, a.y), rd = rot_y(rd, -a.x, a.y);
    else if (kind == XF_SCALE) ro = vdiv(ro, a), rd = vdiv(rd, a);
    else if (kind == XF_MOVE) ro = vsub(ro, smul(rtime, a));
  };

  // The list level for a ray that is being created (flat_scene.h P2Kind): every item's side pair goes straight to the path's slot.
  // `id` = the slot, (ro, rd, rtime) the new ray, (pixel, sample, event) its RNG stream (event = the hit_top the ray is for).
  auto hoist_eval = [&](const uint32_t id, const V3 ro, const V3 rd, const float rtime, const uint64_t seed, const uint32_t pixel,
                        const uint32_t sample, const uint32_t event, uint32_t& n_aabb, uint32_t& n_prim) {
    n_aabb = 0u, n_prim = 0u;
    const uint32_t n_items = __builtin_amdgcn_readfirstlane(s_table[4u * P2_MAX_ITEMS]);
    for (uint32_t k = 0; k < n_items; k++) {
      const uint32_t kind = __builtin_amdgcn_readfirstlane(s_table[4u * k]), ia = __builtin_amdgcn_readfirstlane(s_table[4u * k + 1u]) * RSZ;
      if (kind == P2_PRIMS) {  // a run of plain primitives against a shrinking t, as the walk would with t_range.end = f32::MAX
        const uint32_t q_end = __builtin_amdgcn_readfirstlane(s_table[4u * k + 2u]) * RSZ;
        float c_t = F32_MAX;
        uint32_t c_pc = 0u;
        for (uint32_t q = ia; q < q_end && !(RT_P2_EXP & 1);) {
          float t;
          uint32_t tag, q_next;
          if (prim_test(std::true_type{}, q, ro, rd, rtime, c_t, t, tag, n_prim, q_next)) c_t = t, c_pc = q | tag;
          q = __builtin_amdgcn_readfirstlane(q_next);
        }
        SL_ST_SIDE(id, k, f2u(c_t), c_pc);
      } else if (kind == P2_MEDIUM) {  // the two boundary queries of ConstantMedium::hit (object.rs:551-552): independent of t_range
        const uint4 blo = P2_LO(ia + RSZ), bhi = P2_HI(ia + RSZ);
        float t1 = 0.f, t2 = 0.f;
        uint32_t n_tests;
        const bool crossed = (RT_P2_EXP & 2) ? (n_tests = 0, false) : boundary_pair_t(blo, bhi, ro, rd, t1, t2, n_tests);
        n_prim += n_tests;
        SL_ST_SIDE(id, k, f2u(crossed ? t1 : P2_NOT_CROSSED), f2u(t2));
      } else {  // P2_WRAPPED: the wrapper's PUSH on a copy of the ray, then Aabb::hit's two distances for the Bvh's root box (aabb.rs:16-27)
        V3 to = ro, td = rd;
        if (!(RT_P2_EXP & 4)) push_xform(P2_LO(ia), P2_HI(ia), rtime, to, td);
        const V3 ti = mk(1.f / td.x, 1.f / td.y, 1.f / td.z);
        const uint4 b_lo = P2_LO(ia + RSZ), b_hi = P2_HI(ia + RSZ);
        const float t0x = (u2f(b_lo.x) - to.x) * ti.x, t1x = (u2f(b_lo.y) - to.x) * ti.x;
        const float t0y = (u2f(b_lo.z) - to.y) * ti.y, t1y = (u2f(b_lo.w) - to.y) * ti.y;
        const float t0z = (u2f(b_hi.x) - to.z) * ti.z, t1z = (u2f(b_hi.y) - to.z) * ti.z;
        const float ax = ti.x < 0.f ? t1x : t0x, bx = ti.x < 0.f ? t0x : t1x;
        const float ay = ti.y < 0.f ? t1y : t0y, by = ti.y < 0.f ? t0y : t1y;
        const float az = ti.z < 0.f ? t1z : t0z, bz = ti.z < 0.f ? t0z : t1z;
        const float start = rs_max(t_near, rs_max(rs_max(ax, ay), az));
        const float far = rs_min(rs_min(bx, by), bz);  // the commit takes rs_min(best, far): aabb.rs:26 with the range's end of that moment
        n_aabb += 1u;
        SL_ST_SIDE(id, k, f2u(start), f2u(far));
      }
    }
    if (!(RT_P2_EXP & 8) && __builtin_amdgcn_readfirstlane(s_table[4u * P2_MAX_ITEMS + 1u]) != 0u) {  // media: draws 0 and 1 of the ray's event (object.rs:562)
      SampleRng r;
      r.init(seed, pixel, sample);
      r.set_event(event);
      r.refill();
      SL_ST2(id, PS_LN, f2u(rt_logf((float)(r.b0 >> 8) * (1.0f / 16777216.0f))), f2u(rt_logf((float)(r.b1 >> 8) * (1.0f / 16777216.0f))));
    }
  };

  // The hit record (object.rs:61-71) of a finished ray from the ray, t and the winning record (rt_pool_full.h rebuild_hit): at
  // most ONE wrapper surrounds a record of the second program -- the W item whose records (a, c) hold it.
  auto rebuild_hit = [&](const uint32_t bpc, V3 ro, V3 rd, const float rtime, const float t, V3& p, V3& n, uint32_t& mat) {
    const uint32_t rpc = bpc & ~(RSZ - 1u);
    uint32_t wpc = 0xffffffffu;
    const uint32_t n_items = __builtin_amdgcn_readfirstlane(s_table[4u * P2_MAX_ITEMS]);
    if (__builtin_amdgcn_readfirstlane(s_table[4u * P2_MAX_ITEMS + 2u]) != 0u)
      for (uint32_t k = 0; k < n_items; k++)
        if (__builtin_amdgcn_readfirstlane(s_table[4u * k]) == P2_WRAPPED) {
          const uint32_t ia = __builtin_amdgcn_readfirstlane(s_table[4u * k + 1u]) * RSZ, ic = __builtin_amdgcn_readfirstlane(s_table[4u * k + 3u]) * RSZ;
          if (rpc > ia && rpc < ic) wpc = ia;
        }
    uint4 x_lo = make_uint4(0, 0, 0, 0), x_hi = x_lo;
    if (wpc != 0xffffffffu) {
      x_lo = P2_LO(wpc), x_hi = P2_HI(wpc);
      push_xform(x_lo, x_hi, rtime, ro, rd);
    }
    const uint4 r_lo = P2_LO(rpc), r_hi = P2_HI(rpc);
    const uint32_t r_op = r_hi.w & 0xffu;
    mat = r_hi.z;
    if (r_op == OP_SPHERE) {
      const V3 off = mk(u2f(r_lo.x), u2f(r_lo.y), u2f(r_lo.z));
      V3 lo_o = ro;
      if (r_hi.w & F_TRANSLATE) lo_o = vsub(ro, off);
      if (r_hi.w & F_MOVE) {
        const uint4 mv = P2_LO(rpc + RSZ);
        lo_o = vsub(lo_o, smul(rtime, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
      }
      p = vadd(lo_o, smul(t, rd));
      n = sdiv(p, u2f(r_lo.w));
      if (r_hi.w & F_TRANSLATE) p = vadd(p, off);
      if (r_hi.w & F_FLIP) n = vneg(n);
    } else if (r_op == OP_RECT) {
      const uint32_t axis = (r_hi.w >> F_AXIS_SHIFT) & 3u;
      n = mk(axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f);
      if (r_hi.w & F_FLIP) n = vneg(n);
      p = vadd(ro, smul(t, rd));
    } else if (r_op == OP_PRISM) {
      p = vadd(ro, smul(t, rd)), n = prism_normal(bpc & 7u);
    } else {  // MEDIUM: object.rs:567-571
      p = vadd(ro, smul(t, rd)), n = mk(1.f, 0.f, 0.f);
    }
    if (wpc != 0xffffffffu) {  // back out through the wrapper (object.rs:279-282, 365-369, 314-318, 249-252)
      const uint32_t kind = (x_hi.w >> F_KIND_SHIFT) & 7u;
      const V3 a = mk(u2f(x_lo.x), u2f(x_lo.y), u2f(x_lo.z));
      if (kind == XF_TRANSLATE) p = vadd(p, a);
      else if (kind == XF_ROTATE_Y) p = rot_y(p, a.x, a.y), n = rot_y(n, a.x, a.y);
      else if (kind == XF_SCALE) p = vmul(p, a), n = vdiv(n, a);
      else if (kind == XF_FLIP) n = vneg(n);
      if (x_hi.w & F_PRE_TRANSLATE) p = vadd(p, mk(u2f(x_lo.w), u2f(x_hi.x), u2f(x_hi.y)));
    }
  };

  for (;;) {
    uint32_t op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    const uint64_t m_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    const uint64_t m_slow = __builtin_amdgcn_ballot_w64(op >= OP_SPHERE && op != 0xffu);
    const uint32_t n_busy = (uint32_t)__builtin_popcountll(m_box | m_slow);
    // ============================== SERVICE ======================================================
    const uint32_t n_fin = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_END));
    const bool can_serve = t_count != 0u || s_count + n_fin >= 64u || (TEX && x_count + n_fin >= 64u) || e_count + n_fin >= 64u;
    if ((64u - n_busy >= tune.refill_min && can_serve) || n_busy == 0) {
      __builtin_amdgcn_s_setprio(RT_P2_SERVICE_PRIO);
      if (COUNT) t_mark = RT_TICK();
      {  // (1) finish: what the walk found goes to the slot, the slot's id to S, X or E
        const bool fin = have_ray && op == OP_END;
        const bool to_e = fin && hmat == NO_HIT;
        const bool to_x = TEX && fin && !to_e && (hmat & 8u) != 0u;
        const bool to_s = fin && !to_e && !to_x;
        const uint64_t m_e = __builtin_amdgcn_ballot_w64(to_e), m_x = __builtin_amdgcn_ballot_w64(to_x), m_s = __builtin_amdgcn_ballot_w64(to_s);
        if (fin) {
#if RT_P2_PREFETCH
          // the pass that takes this slot next reads its whole line, which has left the L2 since the refill (the working set of an XCD's
          // waves is ~14 MB): ask for it now -- one dword, nobody waits for it -- so that the line is back when the pass starts
          prefetch_sink ^= SL_LD1(my_slot, to_e ? PS_SAMPLE : PS_O);
#endif
          if (!to_e) SL_ST3(my_slot, PS_BEST, f2u(best), hmat, ev & 0xffu);
          if (COUNT && tr_slot) tr_slot[my_slot] += ev & 0xffu, tr_slot[P2POOL + my_slot] += cnt.aabb - tr_a0, tr_slot[2u * P2POOL + my_slot] += cnt.prim - tr_p0;
          if (to_e) te[P2POOL - 1u - (e_count + lane_rank(m_e))] = (uint16_t)(my_slot | E_MISS);
          else if (to_x) sx[P2POOL - 1u - (x_count + lane_rank(m_x))] = (uint16_t)my_slot;
          else sx[s_count + lane_rank(m_s)] = (uint16_t)my_slot;
          have_ray = false;
        }
        e_count += (uint32_t)__builtin_popcountll(m_e);
        x_count += (uint32_t)__builtin_popcountll(m_x);
        s_count += (uint32_t)__builtin_popcountll(m_s);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (COUNT) t_fin += RT_TICK() - t_mark, n_serv++;
      // (2) shade: Material::scatter for 64 finished rays (color() loop body, lib.rs:73-97)
      auto shade_pass = [&](auto textured_tag) {
        constexpr bool TEXTURED = decltype(textured_tag)::value;
        uint32_t& count = TEXTURED ? x_count : s_count;
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        const uint32_t take = count < 64u ? count : 64u;
        count -= take;
        if (COUNT) n_shade++, n_shade_lanes += take, t_mark2 = RT_TICK();
        bool lpt_on = false, live = false, ended = false, deferred = false;
        uint32_t id = 0, x = 0, row = 0, s = 0;
        if (lane < take) {
          id = TEXTURED ? sx[P2POOL - 1u - (count + lane)] : sx[count + lane];  // pop: the top `take` entries
          const p2_u32x4 a0 = SL_LD4(id, PS_O), a1 = SL_LD4(id, PS_D), a2 = SL_LD4(id, PS_BEST), a3 = SL_LD4(id, PS_STRENGTH);
          const V3 ray_o = mk(u2f(a0.x), u2f(a0.y), u2f(a0.z));
          V3 sd = mk(u2f(a1.x), u2f(a1.y), u2f(a1.z));
          const float stime = u2f(a0.w), hit_t = u2f(a2.x);
          const uint32_t hpc = a2.y, evw = a2.z, tries = evw >> 28;
          uint32_t bounces = a2.w;
          s = a1.w;
          V3 strength = mk(u2f(a3.x), u2f(a3.y), u2f(a3.z));
          x = a3.w & 0xffffu, row = a3.w >> 16;
          // the hit record and the texture value first, while little of this pass is live (rt_pool_full.h)
          V3 p, n;
          uint32_t hm;
          rebuild_hit(hpc, ray_o, sd, stime, hit_t, p, n, hm);
          uint4 mlo, mhi;
          if (mat_lds) mlo = s_mem[(mat_lds >> 4) + 2u * hm], mhi = s_mem[(mat_lds >> 4) + 2u * hm + 1u];
          else mlo = sc.mat[2 * hm], mhi = sc.mat[2 * hm + 1];
          // (= albedo / emission colour for constant textures; checker / Perlin code only in passes over the X list)
          const V3 texval = (TEX && TEXTURED) ? material_texture<FEAT>(sc, mlo, mhi, p) : material_texture<(FEAT & ~FEAT_TEXTURE)>(sc, mlo, mhi, p);
          const uint32_t pixel = (P.ny - 1u - row) * P.nx + x;
          SampleRng rng;
          rng.init(seed, pixel, s);
          rng.set_event(bounces + 1u);
          rng.seek(evw & 0x0fffffffu);  // behind the medium draws of this event's walk and the attempts earlier passes made (rt_pool.h RT_SCATTER_TRIES)
          ended = true;
          const uint32_t kind = mhi.w & 0xffu;
          const float param = u2f(mlo.w);
          V3 emitted = mk(0.f, 0.f, 0.f);
          if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, texval);  // material.rs:120-128
          // no accum field: it is +0 whenever it is read (rt_pool.h PoolField)
          const V3 accum = vadd(mk(0.f, 0.f, 0.f), vmul(strength, emitted));
          V3 nd = mk(0.f, 0.f, 0.f), att = texval;  // Lambertian / Isotropic: albedo(p)
          bool scattered = true;
          V3 rs = mk(0.f, 0.f, 0.f);
          if (kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC)
            deferred = !in_unit_sphere_tries(rng, (RT_SCATTER_TRIES && tries < 3u) ? (uint32_t)RT_SCATTER_TRIES : 0xffffffffu, rs);
          if (COUNT && !deferred) cnt.shaded++;
          float sd_len = 0.f;  // |d| and unit(d) once for the Metal and the Dielectric lanes (rt_pool.h)
          V3 sd_unit = sd;
          if (kind == MAT_METAL || kind == MAT_DIELECTRIC) sd_len = vlen(sd), sd_unit = sdiv(sd, sd_len);
          if (kind == MAT_LAMBERTIAN) {  // material.rs:57-65
            V3 target = vadd(vadd(p, n), rs);
            nd = vsub(target, p);
          } else if (kind == MAT_METAL) {  // material.rs:66-80
            V3 refl = reflect(sd_unit, n);
            nd = vadd(refl, smul(param, rs));
            att = mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
            scattered = vdot(nd, n) > 0.f;
          } else if (kind == MAT_DIELECTRIC) {  // material.rs:81-107
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
          } else {  // Isotropic, material.rs:109-116
            nd = rs;
          }
          if (COUNT) total_draws += rng.draws;
          if (COUNT && tr_slot) tr_slot[id] += rng.draws;
          if (deferred) {  // no direction yet: the slot stays as it is but for the stream position (rt_pool.h)
            ended = false;
            SL_ST1(id, PS_EV, ((evw & 0x0fffffffu) + rng.draws) | ((tries + 1u) << 28));
          } else if (scattered) {
            strength = vmul(strength, att);  // lib.rs:87
            if (bounces != P.max_bounces) {  // lib.rs:93-97
              bounces += 1;
              ended = false, live = true;
              lpt_on = s < cm.lpt_samples && bounces == cm.lpt_deep;  // phase 1 of the cost-ordered queue (rt_pool.h)
            }
          }
          if (live) {  // the new ray: slot (time, sample, pixel stay), list level evaluated, onto T
            SL_ST3(id, PS_O, f2u(p.x), f2u(p.y), f2u(p.z));
            SL_ST3(id, PS_D, f2u(nd.x), f2u(nd.y), f2u(nd.z));
            SL_ST1(id, PS_BOUNCES, bounces);
            SL_ST3(id, PS_STRENGTH, f2u(strength.x), f2u(strength.y), f2u(strength.z));
            uint32_t h_aabb, h_prim;
            if (RT_P2_EXP & 16) {
              V3 p2_ = p;
              asm volatile("" : "+v"(p2_.x));
              hoist_eval(id, p2_, nd, stime, seed, pixel, s, bounces + 1u, h_aabb, h_prim);
            }
            hoist_eval(id, p, nd, stime, seed, pixel, s, bounces + 1u, h_aabb, h_prim);
            if (COUNT) hoisted.aabb += h_aabb, hoisted.prim += h_prim, cnt.rays++;
            if (COUNT && tr_slot) tr_slot[P2POOL + id] += h_aabb, tr_slot[2u * P2POOL + id] += h_prim;
          }
          if (ended) {  // both early returns of color() yield accum (lib.rs:90,94)
            const uint32_t w = pixel_to_work(P, load_const(&lc->pm), x, row);
            float* sp = cm.scratch + 3ull * ((size_t)s * cm.pix_work + w);
            RT_SCRATCH_STORE(sp, accum);
            if (COUNT && tr_out) {
              uint32_t* tp = tr_out + 4ull * ((size_t)s * cm.pix_work + w);
              tp[0] = bounces, tp[1] = tr_slot[id], tp[2] = tr_slot[P2POOL + id], tp[3] = tr_slot[2u * P2POOL + id];
            }
            s++;
            SL_ST1(id, PS_SAMPLE, s);
          }
        }
        if (cm.lpt_samples) lpt_count(cm, lpt_on, lpt_on ? pixel_to_work(P, load_const(&lc->pm), x, row) >> 8 : 0u);
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live), m_end = __builtin_amdgcn_ballot_w64(ended), m_def = __builtin_amdgcn_ballot_w64(deferred);
        if (live) te[t_count + lane_rank(m_live)] = (uint16_t)id;
        if (ended) te[P2POOL - 1u - (e_count + lane_rank(m_end))] = (uint16_t)(id | ((s == P.ns || s % cm.chunk == 0u) ? E_ITEM : E_NEXT));
        if (deferred) {  // back onto the list it came from, over entries this pass has consumed
          if (TEXTURED) sx[P2POOL - 1u - (count + lane_rank(m_def))] = (uint16_t)id;
          else sx[count + lane_rank(m_def)] = (uint16_t)id;
        }
        t_count += (uint32_t)__builtin_popcountll(m_live);
        e_count += (uint32_t)__builtin_popcountll(m_end);
        count += (uint32_t)__builtin_popcountll(m_def);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_shade += RT_TICK() - t_mark2;
      };
      // (2b) camera rays for 64 paths that begin (par_cast closure, lib.rs:366-371: event 0): a miss is booked first (lib.rs:100)
      auto gen_pass = [&]() {
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        const uint32_t take = e_count < 64u ? e_count : 64u;
        e_count -= take;
        if (COUNT) n_gen++, n_gen_lanes += take, t_mark2 = RT_TICK();
        uint32_t st = ST_DEAD, id = 0, s = 0, x = 0, row = 0;
        if (lane < take) {
          const uint32_t ent = te[P2POOL - 1u - (e_count + lane)];
          id = ent & E_ID;
          st = ST_NEED_PIXEL;
          if ((ent & E_ITEM) == 0u) {
            s = SL_LD1(id, PS_SAMPLE);
            const uint32_t xy = SL_LD1(id, PS_XY);
            x = xy & 0xffffu, row = xy >> 16;
            st = ST_GEN;
            if ((ent & E_NEXT) == 0u) {  // E_MISS: the path's colour is black, accum is discarded (lib.rs:100)
              const uint32_t w = pixel_to_work(P, load_const(&lc->pm), x, row);
              float* sp = cm.scratch + 3ull * ((size_t)s * cm.pix_work + w);
              RT_SCRATCH_STORE(sp, mk(0.f, 0.f, 0.f));
              if (COUNT && tr_out) {
                uint32_t* tp = tr_out + 4ull * ((size_t)s * cm.pix_work + w);
                tp[0] = SL_LD1(id, PS_BOUNCES), tp[1] = tr_slot[id], tp[2] = tr_slot[P2POOL + id], tp[3] = tr_slot[2u * P2POOL + id];
              }
              s++;
              if (s == P.ns || s % cm.chunk == 0u) st = ST_NEED_PIXEL;
            }
          }
        }
        for (;;) {  // next work item (rt_pool.h)
          const uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
          if (need == 0) break;
          if (w_next == w_end && !exhausted) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(queue, cm.work_block);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total_work) {
              exhausted = true;
              if (COUNT) t_exhausted = RT_TICK();
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
              if (work_to_pixel(P, load_const(&lc->pm), w, x, row) && first < P.ns) {
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
        const bool live = st == ST_GEN;
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live);
        if (live) {
          const uint32_t y = P.ny - 1u - row;
          SampleRng rng;
          rng.init(seed, y * P.nx + x, s);
          float u = ((float)x + rng.gen_f32()) / (float)P.nx;
          float v = ((float)y + rng.gen_f32()) / (float)P.ny;
          const DevCamera cam = load_const(&lc->cam);
          V3 so, sd;
          float stime;
          get_ray(cam, u, v, rng, so, sd, stime);
          if (COUNT) total_draws += rng.draws;
          SL_ST4(id, PS_O, f2u(so.x), f2u(so.y), f2u(so.z), f2u(stime));
          SL_ST4(id, PS_D, f2u(sd.x), f2u(sd.y), f2u(sd.z), s);
          SL_ST1(id, PS_BOUNCES, 0u);
          SL_ST4(id, PS_STRENGTH, f2u(1.f), f2u(1.f), f2u(1.f), x | (row << 16));  // lib.rs:63
          uint32_t h_aabb, h_prim;
          hoist_eval(id, so, sd, stime, seed, y * P.nx + x, s, 1u, h_aabb, h_prim);
          if (COUNT) hoisted.aabb += h_aabb, hoisted.prim += h_prim, cnt.rays++;
          if (COUNT && tr_slot) tr_slot[id] = rng.draws, tr_slot[P2POOL + id] = h_aabb, tr_slot[2u * P2POOL + id] = h_prim;
          te[t_count + lane_rank(m_live)] = (uint16_t)id;
        }
        t_count += (uint32_t)__builtin_popcountll(m_live);
        n_dead += take - (uint32_t)__builtin_popcountll(m_live);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_gen += RT_TICK() - t_mark2;
      };
      // full passes first; partial ones only when the lanes would have nothing to traverse, and then for every list that holds something
      for (;;) {
        const bool any_part = n_busy == 0 && t_count == 0;
        uint32_t which = s_count >= 64u ? 1u : (TEX && x_count >= 64u) ? 2u : e_count >= 64u ? 3u : 0u;
        if (which == 0u && any_part) which = e_count ? 3u : s_count ? 1u : (TEX && x_count) ? 2u : 0u;
        if (which == 0u) break;
        if (which == 1u) shade_pass(std::false_type{});
        else if (which == 3u) gen_pass();
        else if (TEX) shade_pass(std::true_type{});
      }
      if (COUNT) t_mark2 = RT_TICK();
      if (exhausted && load_const(&lc->cm.drain_share) != 0u) {  // ---- drain-phase work sharing (rt_pool_full.h DrainCtl: same protocol, whole slot lines) ----
        p2_u32x4* dq = reinterpret_cast<p2_u32x4*>(uniform_ptr(wg_slots + (size_t)n_waves * (P2POOL * (P2_SLOT_BYTES / 4u))));
        if (n_dead == P2POOL) {  // run dry (every slot is free, every list empty): register as hungry, adopt what waits in the workgroup's buffer
          if (!hungry || __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT)) != 0u) {
            pool_lock(ctl + DC_LOCK, lane);
            const uint32_t n = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT));
            const uint32_t h = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) + (hungry ? 0u : 1u);  // hungry waves, this one included
            uint32_t k = (n + h - 1u) / h;  // an even share of what waits
            k = k < 64u ? k : 64u;
            if (lane < k) {  // the adopted paths take slots 0 .. k - 1
              const uint32_t src = n - k + lane;
              for (uint32_t f = 0; f < P2_SLOT_BYTES / 16u; f++) {
                const p2_u32x4 v = dq[src * (P2_SLOT_BYTES / 16u) + f];
                if (COUNT && tr_slot && f == PS_BLK / 16u) tr_slot[lane] = v.x, tr_slot[P2POOL + lane] = v.y, tr_slot[2u * P2POOL + lane] = v.z;
                __builtin_amdgcn_raw_buffer_store_b128(v, qr, lane * P2_SLOT_BYTES + 16u * f, 0, 0);
              }
              te[lane] = (uint16_t)lane;
            }
            if (lane == 0u) {
              __hip_atomic_store(ctl + DC_COUNT, n - k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(ctl + DC_HUNGRY, k ? h - 1u : h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            pool_unlock(ctl + DC_LOCK, lane);
            hungry = k == 0u;
            t_count = k, n_dead -= k;
          }
        } else if (w_next == w_end && t_count >= 2u && __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) != 0u) {
          pool_lock(ctl + DC_LOCK, lane);  // holds rays while a wave of the workgroup has none: hand half of them over
          const uint32_t n = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT));
          uint32_t k = t_count / 2u;
          k = k < 64u ? k : 64u;
          k = k < DQ_CAP - n ? k : DQ_CAP - n;
          if (lane < k) {
            const uint32_t id = te[t_count - k + lane], dst = n + lane;
            for (uint32_t f = 0; f < P2_SLOT_BYTES / 16u; f++) {
              p2_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(qr, id * P2_SLOT_BYTES + 16u * f, 0, 0);
              if (COUNT && tr_slot && f == PS_BLK / 16u) v.x = tr_slot[id], v.y = tr_slot[P2POOL + id], v.z = tr_slot[2u * P2POOL + id];  // (the trace accumulators ride in the spare words)
              dq[dst * (P2_SLOT_BYTES / 16u) + f] = v;
            }
          }
          if (lane == 0u) __hip_atomic_store(ctl + DC_COUNT, n + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          pool_unlock(ctl + DC_LOCK, lane);
          t_count -= k, n_dead += k;
        }
      }
      {  // (3) refill idle lanes from T
        const uint64_t m_idle = __builtin_amdgcn_ballot_w64(!have_ray);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(m_idle);
        const uint32_t got = n_idle < t_count ? n_idle : t_count;
        if (got) {
          const uint32_t r = lane_rank(m_idle);
          if (!have_ray && r < got) {
            ev = (uint32_t)te[t_count - 1u - r] << 8;
            const p2_u32x4 a0 = SL_LD4(my_slot, PS_O);
            const p2_u32x3 a1 = SL_LD3(my_slot, PS_D);
#if !RT_P2_RELOAD_SIDE
            sdA = SL_LD4(my_slot, PS_SIDE), sdB = SL_LD4(my_slot, PS_SIDE + 16u), sdC = SL_LD4(my_slot, PS_SIDE + 32u);
#endif
            o = mk(u2f(a0.x), u2f(a0.y), u2f(a0.z)), time = u2f(a0.w);
            d = mk(u2f(a1.x), u2f(a1.y), u2f(a1.z));
            pc = 0, best = F32_MAX, hmat = NO_HIT;
            if (COUNT) tr_a0 = cnt.aabb, tr_p0 = cnt.prim;
            have_ray = true;
          }
          t_count -= got;
          if (COUNT) n_refill++;
        }
      }
      // 1/d and the current record of EVERY lane are (re)derived here, so that these registers are dead across the passes above
#if RT_P2_RELOAD_SIDE
      // ... and so is the side record: every lane that holds a ray reads it (again) from its slot -- 12 registers that the passes need
      sdA = SL_LD4(my_slot, PS_SIDE), sdB = SL_LD4(my_slot, PS_SIDE + 16u), sdC = SL_LD4(my_slot, PS_SIDE + 32u);
#endif
      inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      cur_lo = P2_LO(pc), cur_hi = P2_HI(pc);
      if (COUNT) t_refill += RT_TICK() - t_mark2;
      if (COUNT) t_serv += RT_TICK() - t_mark;
      __builtin_amdgcn_s_setprio(RT_P2_BOX_PRIO);
      if (n_dead == P2POOL) {  // every slot retired
        if (load_const(&lc->cm.drain_share) != 0u) {  // leave when every wave of the workgroup has run dry and nothing waits to be adopted
          if (hungry && __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) == n_waves &&
              __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT)) == 0u) break;
          __builtin_amdgcn_s_sleep(127);  // (poll every ~3.5 us)
#ifdef RT_DRAIN_WATCHDOG
          if (++drain_polls > (uint32_t)(RT_DRAIN_WATCHDOG)) __builtin_trap();
#endif
          continue;
        }
        break;
      }
      if (__builtin_amdgcn_ballot_w64(have_ray) == 0) continue;
      op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    }
    // ============================== TRAVERSE ======================================================
    for (;;) {
      // Box runs and slow passes alternate.  A slow pass executes ONE record for the lanes parked on the kinds it runs; a kind runs
      // when enough lanes wait on it (its own threshold: the cheaper and the more frequent a kind, the fewer lanes it needs --
      // n_k ~ sqrt(f_k c_k), HISTORY.md) -- or when nothing else can run.
      const uint64_t b_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
      const uint32_t n_sph = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_SPHERE));
      const uint32_t n_pri = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_PRISM || op == OP_RECT));
      const uint32_t n_lst = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_LIST));
      const uint32_t n_psh = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_PUSH || op == OP_POP));
      bool r_sph = n_sph >= tune.t_sphere, r_pri = n_pri >= tune.t_prism, r_lst = n_lst >= tune.t_list, r_psh = n_psh >= tune.t_push;  // (thresholds >= 1: the launcher clamps)
      const uint32_t n_parked = n_sph + n_pri + n_lst + n_psh;
      if (!(r_sph || r_pri || r_lst || r_psh) && n_parked != 0u && (b_box == 0 || n_parked >= tune.park_max)) {
        if (b_box == 0) {  // nothing to traverse: whatever waits, runs
          r_sph = n_sph != 0u, r_pri = n_pri != 0u, r_lst = n_lst != 0u, r_psh = n_psh != 0u;
        } else {  // too many lanes parked below their thresholds: the fullest kind runs
          const uint32_t m = n_sph > n_pri ? (n_sph > n_lst ? (n_sph > n_psh ? n_sph : n_psh) : (n_lst > n_psh ? n_lst : n_psh))
                                           : (n_pri > n_lst ? (n_pri > n_psh ? n_pri : n_psh) : (n_lst > n_psh ? n_lst : n_psh));
          r_lst = n_lst == m, r_pri = !r_lst && n_pri == m, r_sph = !r_lst && !r_pri && n_sph == m, r_psh = !r_lst && !r_pri && !r_sph;
        }
      }
      if (!(r_sph || r_pri || r_lst || r_psh)) {
        if (b_box == 0) break;  // (idle or finished lanes only)
        // ---- box run: tight loop, schedule re-evaluated once `box_leave` lanes have left the BOX state ----
        __builtin_amdgcn_s_setprio(RT_P2_BOX_PRIO);
        const uint32_t n0 = (uint32_t)__builtin_popcountll(b_box);
        const uint32_t floor_lanes = n0 > tune.box_leave ? n0 - tune.box_leave : 0u;
        uint32_t n_now;
        if (COUNT) t_mark = RT_TICK();
        do {
          if (COUNT) n_box_it++;
#define P2_BOX_STEP() \
          if (op == OP_BOX) {  /* Aabb::hit, aabb.rs:16-27 */ \
            if (COUNT) cnt.aabb++; \
            f32x2 tx, ty, tz; \
            tx.x = (u2f(cur_lo.x) - o.x) * inv.x, tx.y = (u2f(cur_lo.y) - o.x) * inv.x; \
            ty.x = (u2f(cur_lo.z) - o.y) * inv.y, ty.y = (u2f(cur_lo.w) - o.y) * inv.y; \
            tz.x = (u2f(cur_hi.x) - o.z) * inv.z, tz.y = (u2f(cur_hi.y) - o.z) * inv.z; \
            const float ax = inv.x < 0.f ? tx.y : tx.x, bx = inv.x < 0.f ? tx.x : tx.y; \
            const float ay = inv.y < 0.f ? ty.y : ty.x, by = inv.y < 0.f ? ty.x : ty.y; \
            const float az = inv.z < 0.f ? tz.y : tz.x, bz = inv.z < 0.f ? tz.x : tz.y; \
            const float start = rs_max(t_near, rs_max(rs_max(ax, ay), az)); \
            const float end = rs_min(best, rs_min(rs_min(bx, by), bz)); \
            pc = (end > start) ? pc + RSZ : cur_hi.z; \
            cur_lo = P2_LO(pc), cur_hi = P2_HI(pc); \
            op = cur_hi.w & 0xffu; \
          }
          P2_BOX_STEP();
          P2_BOX_STEP();  // lanes that left the BOX state sit this one out; the schedule check runs every other step
#undef P2_BOX_STEP
          n_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_BOX));
          if (COUNT) n_box_lanes += n_now;
        } while (n_now > floor_lanes);
        if (COUNT) t_box += RT_TICK() - t_mark;
      } else {
        // ---- slow pass ----
        if (COUNT) t_mark = RT_TICK();
        __builtin_amdgcn_s_setprio(RT_P2_SLOW_PRIO);
        const bool mine = (op == OP_SPHERE && r_sph) || ((op == OP_PRISM || op == OP_RECT) && r_pri) || (op == OP_LIST && r_lst) ||
                          ((op == OP_PUSH || op == OP_POP) && r_psh);
        if (COUNT) n_slow_it++, n_slow_lanes += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
        if (r_sph && op == OP_SPHERE) {  // Sphere::hit, object.rs:84-111 (+ fused Translate / LinearMove)
          if (COUNT) cnt.prim++;
          const uint32_t self_pc = pc;
          V3 lo_o = o;
          if (cur_hi.w & F_TRANSLATE) lo_o = vsub(o, mk(u2f(cur_lo.x), u2f(cur_lo.y), u2f(cur_lo.z)));
          if (cur_hi.w & F_MOVE) {
            const uint4 mv = P2_LO(pc + RSZ);
            lo_o = vsub(lo_o, smul(time, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
            pc += RSZ;
          }
          float t;
          if (sphere_hit_t(lo_o, d, u2f(cur_lo.w), t_near, best, t)) best = t, hmat = self_pc | ((cur_hi.w >> 16) & 8u);
          pc += RSZ;
        }
        if (r_pri && op == OP_PRISM) {  // rect_prism: six Rect::hit in one record
          if (COUNT) cnt.prim += 6;
          float t;
          uint32_t face = 0;
          if (prism_hit_t(cur_lo, cur_hi, o, d, t_near, best, t, face)) best = t, hmat = pc | face | ((cur_hi.w >> 16) & 8u);
          pc += RSZ;
        } else if (r_pri && op == OP_RECT) {  // Rect::hit, object.rs:185-218
          if (COUNT) cnt.prim++;
          float t;
          if (rect_hit_t(o, d, (cur_hi.w >> F_AXIS_SHIFT) & 3u, u2f(cur_lo.x), u2f(cur_lo.y), u2f(cur_lo.z), u2f(cur_lo.w), u2f(cur_hi.x), t_near, best, t))
            best = t, hmat = pc | ((cur_hi.w >> 16) & 8u);
          pc += RSZ;
        }
        if (r_lst) {
          // ---- commit the list-level items [first, first + count) of the lane's OP_LIST record, in order (lib.rs:40-45) ----
          const bool at = op == OP_LIST;
          const uint32_t first = cur_hi.x, last = cur_hi.x + cur_hi.y;
          uint32_t next_pc = pc + RSZ;
          // (the lanes usually stand on the SAME record: then the loop runs over its items only)
          const uint32_t l_at = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(at));  // (r_lst: at least one lane stands on an OP_LIST record)
          const uint32_t f0 = __builtin_amdgcn_readlane(first, l_at), l0 = __builtin_amdgcn_readlane(last, l_at);
          const bool same = __builtin_amdgcn_ballot_w64(at && (first != f0 || last != l0)) == 0;
          const uint32_t k_lo = same ? f0 : 0u, k_hi = same ? l0 : __builtin_amdgcn_readfirstlane(s_table[4u * P2_MAX_ITEMS]);
          for (uint32_t k = k_lo; k < k_hi; k++) {
            const bool in = at && k >= first && k < last;
            uint32_t sa, sb;
            switch (k) {
              case 0: sa = sdA.x, sb = sdA.y; break;
              case 1: sa = sdA.z, sb = sdA.w; break;
              case 2: sa = sdB.x, sb = sdB.y; break;
              case 3: sa = sdB.z, sb = sdB.w; break;
              default: sa = sdC.x, sb = sdC.y; break;
            }
            const uint32_t kind = __builtin_amdgcn_readfirstlane(s_table[4u * k]), ia = __builtin_amdgcn_readfirstlane(s_table[4u * k + 1u]) * RSZ;
            if (kind == P2_PRIMS) {
              // the run's closest candidate against what was found before it: `t < t_range.end` (object.rs:99,195) for the one
              // primitive that can still win.  NaN (a Rect met with 0 / 0, object.rs:194-196: `t < start || t >= end` lets it pass)
              // is not ordered: such a ray replays the run's records against its own best, as the reference walks them.
              const float c_t = u2f(sa);
              const bool odd = in && (c_t != c_t || best != best);
              if (in && !odd && c_t < best) best = c_t, hmat = sb;
              if (__builtin_amdgcn_ballot_w64(odd) != 0) {
                const uint32_t q_end = __builtin_amdgcn_readfirstlane(s_table[4u * k + 2u]) * RSZ;
                if (odd)
                  for (uint32_t q = ia; q < q_end;) {
                    float t;
                    uint32_t tag, q_next, n_unused = 0;
                    if (prim_test(std::false_type{}, q, o, d, time, best, t, tag, n_unused, q_next)) best = t, hmat = q | tag;
                    q = q_next;
                  }
              }
            } else if (kind == P2_MEDIUM) {  // ConstantMedium::hit once both boundary queries hit at t1, t2 (object.rs:553-574)
              float t1 = u2f(sa), t2 = u2f(sb);
              if (in && t1 != P2_NOT_CROSSED) {
                t1 = rs_max(t1, t_near);
                t2 = rs_min(t2, best);
                if (!(t1 >= t2)) {
                  const uint4 m_lo = P2_LO(ia), m_hi = P2_HI(ia);
                  const float len = vlen(d);
                  const float distance_inside = (t2 - t1) * len;
                  const float ln_u = u2f((ev & 0xffu) == 0u ? sdC.z : sdC.w);  // ln of draw `ev` of the event's stream (evaluated when the ray was created)
                  const float hit_distance = -u2f(m_lo.y) * ln_u;    // -(1. / density) * rng().ln(), object.rs:562 (m_lo.y = 1 / density)
                  ev++;
                  if (COUNT) total_draws++;
                  if (hit_distance < distance_inside) {  // a list world: the later hit replaces (lib.rs:41-44)
                    best = t1 + hit_distance / len;
                    hmat = ia | ((m_hi.w >> 16) & 8u);
                  }
                }
              }
            } else {  // P2_WRAPPED: Aabb::hit of the wrapped Bvh's root box with the range's end of this moment (aabb.rs:26-27)
              if (in) {
                const float end = rs_min(best, u2f(sb));
                next_pc = (end > u2f(sa)) ? ia : __builtin_amdgcn_readfirstlane(s_table[4u * k + 3u]) * RSZ;
              }
            }
          }
          if (at) pc = next_pc;
        }
        if (r_psh && op == OP_PUSH) {  // into a wrapped Bvh whose root box the OP_LIST record in front found open: PUSH, then the root's first child
          push_xform(cur_lo, cur_hi, time, o, d);
          inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
          pc += 2u * RSZ;
        } else if (r_psh && op == OP_POP) {  // the ray outside the wrapper: the slot still holds it
          if (!(cur_hi.w & F_P2_DEAD_POP)) {
            const p2_u32x3 a0 = SL_LD3(my_slot, PS_O), a1 = SL_LD3(my_slot, PS_D);
            o = mk(u2f(a0.x), u2f(a0.y), u2f(a0.z)), d = mk(u2f(a1.x), u2f(a1.y), u2f(a1.z));
            inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
          }
          pc += RSZ;
        }
        if (mine) cur_lo = P2_LO(pc), cur_hi = P2_HI(pc), op = cur_hi.w & 0xffu;
        if (COUNT) t_slow += RT_TICK() - t_mark;
      }
      op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
      const uint32_t busy = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op >= OP_BOX && op != 0xffu));
      if (64u - busy >= tune.refill_min || busy == 0) break;
    }
  }
#if RT_P2_PREFETCH
  if (prefetch_sink == 0x9e3779b9u && total_work == 0xffffffffu) counters[63] = prefetch_sink;
#endif
  if (COUNT) {
    atomicAdd(&counters[0], (unsigned long long)cnt.aabb + hoisted.aabb);
    atomicAdd(&counters[1], (unsigned long long)cnt.prim + hoisted.prim);
    atomicAdd(&counters[2], (unsigned long long)cnt.shaded);
    atomicAdd(&counters[3], (unsigned long long)cnt.rays);
    atomicAdd(&counters[4], (unsigned long long)total_draws);
    if (lane == 0) {
      unsigned long long* sched = counters + 8;
      atomicAdd(&sched[0], (unsigned long long)n_box_it), atomicAdd(&sched[1], (unsigned long long)n_box_lanes);
      atomicAdd(&sched[2], (unsigned long long)n_slow_it), atomicAdd(&sched[3], (unsigned long long)n_slow_lanes);
      atomicAdd(&sched[4], (unsigned long long)n_shade), atomicAdd(&sched[5], (unsigned long long)n_shade_lanes);
      atomicAdd(&sched[6], (unsigned long long)n_refill);
      atomicAdd(&sched[15], t_refill);
      atomicAdd(&sched[12], (unsigned long long)n_gen), atomicAdd(&sched[13], (unsigned long long)n_gen_lanes), atomicAdd(&sched[14], t_gen);
      atomicAdd(&counters[6], t_fin), atomicAdd(&counters[5], n_serv);
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], t_serv - t_shade), atomicAdd(&counters[18], t_box), atomicAdd(&counters[19], t_slow);
      const unsigned long long dur = RT_TICK() - t_start, exh = t_exhausted - t_start;
      atomicMax(&counters[24], dur), atomicAdd(&counters[25], dur), atomicAdd(&counters[26], 1ull);
      atomicMax(&counters[27], (1ull << 62) - exh), atomicAdd(&counters[28], exh), atomicMax(&counters[29], exh);
    }
  }
#undef P2_LO
#undef P2_HI
#undef SL_OFF
#undef SL_LD1
#undef SL_LD2
#undef SL_LD3
#undef SL_LD4
#undef SL_ST1
#undef SL_ST2
#undef SL_ST3
#undef SL_ST4
#undef SL_ST_SIDE
#undef my_slot
}

}  // namespace rtg
