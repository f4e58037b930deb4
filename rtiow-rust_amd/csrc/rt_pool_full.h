// rt_pool_full.h -- the ray-pool schedule of rt_pool.h for EVERY feature of the hot path: Rect,
// PUSH/POP transform wrappers (Translate / RotateY / Scale / LinearMove / FlipNormals over subtrees),
// ConstantMedium (RNG draws during traversal), checker / Perlin textures, Isotropic.
//
// Differences from the lean kernel:
//  * the hit record (p, normal, material) is built at hit time and carried in registers, because it
//    has to travel back through the enclosing POPs (object.rs:279-282,365-369);
//  * paths have no home slot: their state MOVES through four dense per-wave stacks in global memory -- T (rays to
//    traverse: origin, direction, time, strength, bounces, sample, pixel), S and X (finished rays to shade without / with
//    a texture lookup: hit record, direction, time, draws made, strength, bounces, sample, pixel) and N (paths that ended
//    and ask for their successor: sample, pixel).  Every push goes to consecutive positions ([field][position] rows, so a
//    wave's push is whole cache lines) and every pop takes the top: what was written last is read next, from L2.  The
//    lanes carry strength / bounces / sample / pixel in registers while they traverse;
//  * three kinds of 64-wide passes: scatter (S), textured scatter (X), camera rays (N: next sample / next work item);
//  * every non-BOX record (SPHERE, RECT, PUSH, POP, MEDIUM) is a "slow op": lanes park on it and a slow
//    pass executes one record per parked lane once enough lanes wait (or no BOX lane is left);
//  * the transform stack (<= 4 saved rays): level 0 in registers, deeper levels in a per-wave, lane-interleaved global
//    scratch;
//  * a medium draws from the event's RNG stream by index (event_draw), and the number of draws made
//    during traversal travels with the path so that Material::scatter continues the stream where
//    traversal left it (draw order of SURVEY 8a);
//  * always one sample per work item + ordered fold (frames whose sample colours exceed the scratch budget are rendered
//    in several sample passes: rtg_launch.inc samples_per_pass).
#pragma once
#include "rt_pool.h"

namespace rtg {

#ifndef RT_FULL_POOL_SLOTS
#define RT_FULL_POOL_SLOTS 224  // 64 in the lanes + rays that wait for company in S, X and N (up to 63 each) + T to refill from;
                                // book-2: 160 52.6 ms (the lanes starve), 192 45.3, 224 43.9, 256 44.6, 320 45.4
#endif
#ifndef RT_RUN_AHEAD_NOBOX
#define RT_RUN_AHEAD_NOBOX 1  // rt_full_traverse.inc: a slow pass runs ahead while no lane waits at a BOX record
#endif
#ifndef RT_DRAIN_ALL
#define RT_DRAIN_ALL 1  // a starved wave runs the partial passes of ALL its stacks before it refills
#endif
// Wave priorities (s_setprio; the arbiter of a SIMD prefers the wave with the higher one).  The four waves of a SIMD are in
// different phases at any moment; what a phase is worth to the others differs: a SLOW PASS holds parked lanes back (every lane
// of the wave that stands at a non-BOX record waits for it), a box run feeds them, a service (shade / camera passes, stack
// traffic: long dependent chains with trips to L2 in them) has the slack.  Slow passes first, box runs next, services last:
// book-2 800x800x100 40.0 -> 38.7 ms, 400 spp 146.6 -> 143.3, the reference's 300x300x100 frame 10.4 -> 9.8, book2_bvh 43.3 ->
// 42.2 (profiles/r04_experiments/r04s_priority_*; services first: 0 on book-2, -2 % on the lean kernel, which loses with every
// assignment tried and keeps none).
#ifndef RT_FULL_SERVICE_PRIO
#define RT_FULL_SERVICE_PRIO 0
#define RT_FULL_BOX_PRIO 1
#define RT_FULL_SLOW_PRIO 3
#endif
#ifndef RT_FAT_LEAF
#define RT_FAT_LEAF 0  // rt_full_traverse.inc
#endif
#ifndef RT_FULL_BOX_UNROLL
#define RT_FULL_BOX_UNROLL 2  // box steps per schedule check (book-2 43.8 -> 43.0 ms)
#endif
constexpr uint32_t FPOOL = RT_FULL_POOL_SLOTS;  // paths in flight per wave of the full-feature kernel = capacity of each stack
enum FullTField : uint32_t {  // T stack: a ray ready to traverse (the *_TRACE rows only exist for the instrumented variant)
  TQ_O = 0, TQ_D = 3, TQ_TIME = 6, TQ_STRENGTH = 7, TQ_BOUNCES = 10, TQ_SAMPLE = 11, TQ_XY = 12,
  TQ_SEG_T = 13, TQ_SEG_PC = 14,  // the hoisted segment's closest candidate for this ray (hoist_eval; flat_scene.h OP_SEG)
  TQ_TRACE = 15, TQ_FIELDS = 18,
};
enum FullSField : uint32_t {  // S / X stacks: a finished ray with what the walk found -- t and the winning record (RT_DEFER_HIT: the hit
                               // record itself is rebuilt by the shade pass, rebuild_hit)
  SQ_O = 0, SQ_D = 3, SQ_TIME = 6, SQ_T = 7, SQ_HITPC = 8, SQ_EVDRAWS = 9, SQ_STRENGTH = 10, SQ_BOUNCES = 13, SQ_SAMPLE = 14,
  SQ_XY = 15, SQ_TRACE = 16, SQ_FIELDS = 19,
};
enum FullNField : uint32_t { NQ_SAMPLE = 0, NQ_XY = 1, NQ_FIELDS = 2 };  // N stack: a path that ended asks for its successor
constexpr uint32_t NQ_NEED_ITEM = 0xffffffffu;  // NQ_SAMPLE: "the next work item" instead of "sample s of the same pixel"
constexpr uint32_t FPOOL_FIELDS = TQ_FIELDS + 2 * SQ_FIELDS + NQ_FIELDS;  // dwords of stack space per path in flight (T, S, X, N)
// stack space of a workgroup: its waves' stacks, then its hand-over stack (RT_DRAIN_SHARE: [field][position] rows of T records)
inline size_t full_pool_wg_words(uint32_t waves) { return (size_t)waves * FPOOL * FPOOL_FIELDS + (size_t)TQ_FIELDS * 256u; }

// Drain-phase work sharing (RT_DRAIN_SHARE).  A launch ends with a DRAIN: the work queue is empty and every wave finishes the paths
// it holds.  What is left are long bounce chains (book-2: 2.3 % of the samples run into the bounce cap, 51 DEPENDENT generations of
// traverse + shade), spread unevenly: the waves of a book-2 launch drain for 0.6 .. 4.7 ms (median 2.2), and a generation takes a wave
// the longer the more rays it holds (~35 us for one ray, ~55 us for twelve: every ray adds record visits of its own).  So a wave
// that has run dry does not leave: it registers as HUNGRY, and a wave that still holds rays hands half of its T stack -- whole
// records, a path has no other state -- to a per-workgroup stack in global memory (DQ_CAP records behind an LDS spin lock; the
// waves of a workgroup share a CU, hence an L1) whenever a hungry wave waits; hungry waves adopt from it.  The chains of a CU end
// up one or two per wave and finish together; the workgroup leaves when all its waves are hungry and the stack is empty.  A
// path's RNG streams are keyed by (pixel, sample, event): which wave runs it changes no bit.  Measured (r04g_share_*): book-2
// 800x800x100 40.7 -> 40.0 ms, the reference's own 300x300x100 frame 11.9 -> 10.7, fixed cost per launch 8.1 -> 6.0 ms, every
// wave of a launch done within 0.7 ms of the mean (before: 2.3).  What is left of the drain is ONE chain: a path that starts when
// the queue runs dry and bounces 50 times, ~70 us a generation.
// The lean kernel (rt_pool.h) does NOT share: built there too (slot ids made workgroup-wide, a 64-entry mailbox of ids in LDS:
// profiles/r04_experiments/r04i_lean_work_sharing.patch) it gained nothing on book-1 at 50 or 500 spp -- the waves that end a
// book-1 launch hold ONE chain each -- 5 % at 10 spp, and the code cost the steady state 1.7 % (r04i_lean_share_ab*.txt).
// (First built the other way round -- the first wave of every SIMD to see the queue empty COLLECTS the rays of the other twelve,
// because a launch with 4 waves per CU has half the fixed cost -- and measured: no gain, the collectors' generations grow with the
// rays they hold; profiles/r04_experiments/r04f_consolidate_*.)
#ifndef RT_DRAIN_SHARE
#define RT_DRAIN_SHARE 1
#endif
constexpr uint32_t DQ_CAP = 256;                                   // records of a workgroup's hand-over stack
// Control words in LDS.  INVARIANT the lock-free exit test relies on (it reads DC_HUNGRY and DC_COUNT without the lock):
//  * a wave registers as hungry only under the lock, and only when it adopted nothing (k == 0, i.e. DC_COUNT was 0 then);
//    a wave that adopts stores DC_COUNT before DC_HUNGRY, so a reader that sees the smaller DC_HUNGRY also sees the smaller count;
//  * only a wave that still holds rays adds to DC_COUNT, and such a wave is not hungry: DC_HUNGRY == n_waves therefore means no
//    wave can add records any more, and with DC_COUNT == 0 this is the terminal state -- it is reached exactly once, by the last
//    non-hungry wave registering under the lock;
//  * every wave of the workgroup takes part until then: a kernel variant in which a wave returns early, or skips the hand-over
//    block, would leave the others polling for ever.  Debug builds (-DRT_DRAIN_WATCHDOG=<polls>) trap instead of hanging the GPU.
enum DrainCtl : uint32_t { DC_LOCK = 0, DC_COUNT = 1, DC_HUNGRY = 2, DC_WORDS = 16 };
// LDS = the first `window` program records (all of them when the program fits, 0 = none) + the hand-over control words
inline size_t full_pool_lds_bytes(uint32_t window, uint32_t /*waves*/) { return (size_t)window * 32 + DC_WORDS * 4; }  // (+ the materials when they fit: rtg_launch.inc)

// draw `idx` (0-based) of the stream (seed, pixel, sample, event): word idx%4 of Philox block idx/4
__device__ __attribute__((always_inline)) float event_draw_f32_inline(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t event, uint32_t idx) {
  SampleRng r;
  r.init(seed, pixel, sample);
  r.set_event(event);
  r.blk = idx >> 2;
  r.refill();
  uint32_t w = idx & 3u;
  uint32_t u = w == 0 ? r.b0 : (w == 1 ? r.b1 : (w == 2 ? r.b2 : r.b3));
  return (float)(u >> 8) * (1.0f / 16777216.0f);
}

__device__ __attribute__((noinline)) float event_draw_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t event, uint32_t idx) {
  return event_draw_f32_inline(seed, pixel, sample, event, idx);
}
// PROG: 0 = program fetched from global memory (L1/L2), 1 = whole program staged in LDS, 2 = an LDS
// window over the first `window` records (depth-first order, so it holds whole leading subtrees) and global memory for the
// rest (book-2's 4 213 records = 134.8 KB fit whole).
#ifndef RT_FULL_TEX_THREADS
#define RT_FULL_TEX_THREADS 1024  // 16 waves per CU; the textured variant allocates 120-128 VGPRs without a spill (round 1: ~142
                                  // wanted, 25 spilled -- and still 4 % faster than 12 waves at 768 threads)
#endif
// ---- stack access: `qr` = the wave's stack space
RT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
typedef __amdgpu_buffer_rsrc_t QueueRsrc;
constexpr uint32_t SQ_BASE = TQ_FIELDS, NQ_BASE = TQ_FIELDS + 2u * SQ_FIELDS;  // first rows of the S (+X) and N stacks
RT_DEV QueueRsrc make_queue_rsrc(uint32_t* wave_base) {
  // raw buffer (stride 0, bounds = the wave's stack space in bytes; word 3 = gfx9 raw-buffer format bits)
  return __builtin_amdgcn_make_buffer_rsrc(wave_base, 0, FPOOL * FPOOL_FIELDS * 4u, 0x00020000);
}
#define RT_IN_LDS(pc_) (PROG == 1 || (PROG == 2 && (pc_) < win_bytes))
#define RT_FETCH_LO(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + (pc_)) : sc.lo[(pc_) >> 5])
#define RT_FETCH_HI(pc_) (RT_IN_LDS(pc_) ? *reinterpret_cast<const uint4*>(s_bytes + (pc_) + 16u) : fetch_hi_global(sc, (pc_) >> 5))
// Stack records are addressed through ONE buffer resource per wave (`qr`: base = the wave's stack space, in SGPRs) with the
// position as the only VGPR of an access and the row as a constant offset -- no 64-bit address arithmetic, no address registers.
#define Q_ROW(base_, f_) (((base_) + (f_)) * FPOOL * 4u)  /* byte offset of a row */
#define Q_LD(row_, i_) __builtin_amdgcn_raw_buffer_load_b32(qr, (i_) * 4u + ((row_) & 4095u), (row_) & ~4095u, 0)
#define Q_ST(row_, i_, v_) __builtin_amdgcn_raw_buffer_store_b32((v_), qr, (i_) * 4u + ((row_) & 4095u), (row_) & ~4095u, 0)
#define TQ_LD_U(f_, i_) Q_LD(Q_ROW(0u, f_), i_)
#define TQ_LD_F(f_, i_) u2f(TQ_LD_U(f_, i_))
#define TQ_ST_U(f_, i_, v_) Q_ST(Q_ROW(0u, f_), i_, (uint32_t)(v_))
#define TQ_ST_F(f_, i_, v_) Q_ST(Q_ROW(0u, f_), i_, f2u(v_))
#define SQ_LD_U(f_, i_) Q_LD(Q_ROW(SQ_BASE, f_), i_)
#define SQ_LD_F(f_, i_) u2f(SQ_LD_U(f_, i_))
#define SQ_ST_U(f_, i_, v_) Q_ST(Q_ROW(SQ_BASE, f_), i_, (uint32_t)(v_))
#define SQ_ST_F(f_, i_, v_) Q_ST(Q_ROW(SQ_BASE, f_), i_, f2u(v_))
#define NQ_LD_U(f_, i_) Q_LD(Q_ROW(NQ_BASE, f_), i_)
#define NQ_ST_U(f_, i_, v_) Q_ST(Q_ROW(NQ_BASE, f_), i_, (uint32_t)(v_))
constexpr uint32_t XQ = SQ_FIELDS * FPOOL;  // X stack = positions XQ.. of the S rows: hits on checker / Perlin materials

// Push finished rays onto S or X: consecutive positions (wave-uniform call; `fin` selects the lanes).
// hmat bit 31 = the hit material reads a non-constant texture (F_TEXTURED of the winning record): such hits are shaded in
// their own passes, so the Perlin / checker code is issued for 64 textured hits at a time instead of in every pass that
// happens to hold one.
template <bool TEX, bool TRACE>
RT_DEV void full_push_finished(const QueueRsrc qr, uint32_t& s_count, uint32_t& x_count, const bool fin, const V3 fo, const V3 fd,
                               const float ftime, const float ft, const uint32_t fpc, const uint32_t fev, const V3 fstrength,
                               const uint32_t fbounces, const uint32_t fsample, const uint32_t fxy, const bool trace,
                               const uint32_t ft0, const uint32_t ft1, const uint32_t ft2) {
  // bit 3 of the winning record's pc = its material reads a non-constant texture (F_TEXTURED): such hits are shaded in passes of
  // their own, so the Perlin / checker code is issued for 64 textured hits at a time instead of in every pass that holds one
  const bool to_x = TEX && fin && fpc != NO_HIT && (fpc & 8u) != 0u;
  const uint64_t m_fin = __builtin_amdgcn_ballot_w64(fin && !to_x), m_x = __builtin_amdgcn_ballot_w64(to_x);
  if (fin) {
    const uint32_t i = to_x ? XQ + x_count + lane_rank(m_x) : s_count + lane_rank(m_fin);
    SQ_ST_U(SQ_HITPC, i, fpc);
    SQ_ST_F(SQ_T, i, ft);
    SQ_ST_F(SQ_O, i, fo.x), SQ_ST_F(SQ_O + 1, i, fo.y), SQ_ST_F(SQ_O + 2, i, fo.z);
    SQ_ST_F(SQ_D, i, fd.x), SQ_ST_F(SQ_D + 1, i, fd.y), SQ_ST_F(SQ_D + 2, i, fd.z);
    SQ_ST_F(SQ_TIME, i, ftime);
    SQ_ST_U(SQ_EVDRAWS, i, fev);
    SQ_ST_F(SQ_STRENGTH, i, fstrength.x), SQ_ST_F(SQ_STRENGTH + 1, i, fstrength.y), SQ_ST_F(SQ_STRENGTH + 2, i, fstrength.z);
    SQ_ST_U(SQ_BOUNCES, i, fbounces), SQ_ST_U(SQ_SAMPLE, i, fsample), SQ_ST_U(SQ_XY, i, fxy);
    if (TRACE && trace) SQ_ST_U(SQ_TRACE, i, ft0), SQ_ST_U(SQ_TRACE + 1, i, ft1), SQ_ST_U(SQ_TRACE + 2, i, ft2);
  }
  s_count += (uint32_t)__builtin_popcountll(m_fin);
  x_count += (uint32_t)__builtin_popcountll(m_x);
}

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
  constexpr uint32_t OP_SLOW_LAST = (uint32_t)OP_SEG;  // records a slow pass executes: SPHERE .. SEG (BEND only occurs in GENB programs)
  constexpr uint32_t STACK_LEVELS = GENB ? 2 * MAX_XFORM_DEPTH : MAX_XFORM_DEPTH;   // a boundary stream nests below the medium's own wrappers
  constexpr bool USE_LDS = PROG != 0;
  const uint32_t win_bytes = RSZ * window;
  if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < window; i += blockDim.x) {
      uint4 h = sc.hi[i];
      if ((h.w & 0xffu) == OP_BOX || (h.w & 0xffu) == OP_SEG) h.z *= RSZ;
      if ((h.w & 0xffu) == OP_MEDIUM) h.x *= RSZ;
      s_mem[2u * i] = sc.lo[i];
      s_mem[2u * i + 1u] = h;
    }
  }
  const char* s_bytes = reinterpret_cast<const char*>(s_mem);
  const uint32_t mat_lds = load_const(&lc->mat_lds);
  if (mat_lds)
    for (uint32_t i = threadIdx.x; i < 2u * sc.n_mat; i += blockDim.x) s_mem[(mat_lds >> 4) + i] = sc.mat[i];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const size_t gwave = (size_t)blockIdx.x * n_waves + wave;
  static_assert(DQ_CAP == 256u, "full_pool_wg_words");
  const size_t wg_words = (size_t)n_waves * (FPOOL * FPOOL_FIELDS) + (size_t)TQ_FIELDS * DQ_CAP;
  // (wave-uniform, but derived from threadIdx: without the readfirstlane the compiler keeps these in VGPRs and loops over the
  // "different" resources of a wave at every access)
  uint32_t* tq = uniform_ptr(g_slots + (size_t)blockIdx.x * wg_words + (size_t)wave * (FPOOL * FPOOL_FIELDS));
  uint32_t* ctl = reinterpret_cast<uint32_t*>(s_mem + 2u * (USE_LDS ? window : 0u));  // hand-over control words behind the program
  if (threadIdx.x < DC_WORDS) ctl[threadIdx.x] = 0u;
  bool hungry = false;  // this wave has run dry and is counted in ctl[DC_HUNGRY] (wave-uniform)
#ifdef RT_DRAIN_WATCHDOG
  uint32_t drain_polls = 0;
#endif
  const QueueRsrc qr = make_queue_rsrc(tq);                  // rows: T, then S (+X), then G (+R) -- see Q_ROW
  float* stack = uniform_ptr(g_stack + gwave * (STACK_LEVELS * 6 * 64));  // [level][component][lane]; level 0 rides in registers
  for (uint32_t j = lane; j < FPOOL; j += 64u) NQ_ST_U(NQ_SAMPLE, j, NQ_NEED_ITEM);  // FPOOL paths-to-be ask for a work item
  __syncthreads();

  const float t_near = load_const(&lc->P.t_near);
  uint32_t t_count = 0, s_count = 0, x_count = 0, n_count = FPOOL, n_dead = 0;
  uint32_t tn_count = 0;  // RT_T_PRIORITY (rt_pool.h): camera rays of new paths, stacked from position FPOOL - 1 of the T rows downwards
  uint32_t w_next = 0, w_end = 0, w_chunk = 0, w_delta = 0;
  bool w_lpt_ready = false;
  bool exhausted = false;
  const unsigned long long t_start = RT_TICK();
  unsigned long long t_exhausted = 0;
  RT_TL_DECL(FPOOL);

  // ---- per-lane traversal state ---------------------------------------------------------------
  bool have_ray = false;
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  float time = 0.f, best = F32_MAX;
  uint32_t pc = 0;
  uint4 cur_lo = make_uint4(0, 0, 0, 0), cur_hi = make_uint4(0, 0, 0, OP_END);
  V3 hp = o, hn = o;                 // (RT_DEFER_HIT: unused -- the hit record is rebuilt by the shade pass)
  uint32_t hmat = NO_HIT;            // the winning record so far: pc | prism face | bit 3 "textured"; NO_HIT = None
  uint32_t depth = 0, tag = 0, nhits = 0, root_hits = 0, ev_draws = 0;
  uint32_t r_xy = 0, r_sample = 0, r_bounces = 0;  // the path's pixel (x | row << 16), sample and bounce count: its RNG stream for media
  V3 r_strength = o;                                // the path's strength rides along (lib.rs:77)
  V3 sv_o = o, sv_d = o;                            // level 0 of the transform stack (the ray outside the outermost wrapper)
  float seg_t = F32_MAX;                             // the hoisted segment's candidate for this ray: t and record pc (| prism face), see OP_SEG
  uint32_t seg_pc = 0;
  uint32_t bmode = 0;                                // GENB: 0 = main walk, 1 / 2 = inside a boundary stream, query 1 / 2
  float t_lo = t_near, b_saved = 0.f, b_t1 = 0.f;    // GENB: lower end of the current range; the main walk's best; query 1's t
  Counts cnt = {0, 0, 0, 0};
  uint32_t hoisted_prim = 0;  // primitive tests of the hoisted segment, made for rays a pass creates
  uint32_t total_draws = 0;
  uint32_t* tr_out = nullptr;  // per-sample trace of the instrumented variant (rt_pool.h)
  uint32_t tr_a0 = 0, tr_p0 = 0, tr_d = 0, tr_a = 0, tr_p = 0;  // the path's running draws / Aabb tests / primitive tests
  if (COUNT) tr_out = reinterpret_cast<uint32_t*>(counters[30]);
  uint32_t n_box_it = 0, n_box_lanes = 0, n_slow_it = 0, n_slow_lanes = 0, n_shade = 0, n_shade_lanes = 0, n_refill = 0;
  uint32_t n_gen = 0, n_gen_lanes = 0;
  unsigned long long t_gen = 0;
#ifdef RT_CENSUS
  unsigned long long census[2] = {0ull, 0ull};  // lane c < 16 holds class c of the box-step census [0] and of the slow-pass census [1]
#endif
  unsigned long long t_shade = 0, t_serv = 0, t_box = 0, t_slow = 0, t_refill = 0, t_fin = 0, n_serv = 0, t_mark = 0, t_mark2 = 0;  // COUNT: s_memtime shares

#define RT_DEFER_HIT 1   // hits are (t, record pc): rt_full_ops.inc
#define RT_HOIST 1       // OP_SEG commits the candidate hoist_eval found when the ray was created
#define RT_REG_STACK0 1  // PUSH / POP of an outermost wrapper touch no memory (book-2: the moving sphere, the sphere cloud)
#ifndef RT_POOL_SAME_KIND_RUN
#define RT_POOL_SAME_KIND_RUN 0
#endif
#define RT_SAME_KIND_RUN RT_POOL_SAME_KIND_RUN  // (consecutive SPHERE / RECT records in one go: the lock-step kernel's; here the box lanes would wait: book2_bvh +1.7 %)
#include "rt_full_ops.inc"
#undef RT_SAME_KIND_RUN
#undef RT_REG_STACK0
#undef RT_HOIST
#undef RT_DEFER_HIT
  // The hoisted segment (flat_scene.h OP_SEG) for a ray that is being created: Sphere::hit / Rect::hit / rect_prism's six
  // Rect::hit for every record of the segment in order against a shrinking t -- exactly what the walk would do with
  // t_range.end = f32::MAX -- keeping the closest candidate.  Called by the shade and camera passes with every lane on the
  // SAME record (the record words are wave-uniform: scalar branches, no divergence but hit / miss).
  auto hoist_eval = [&](const V3 ro, const V3 rd, const float rtime, float& c_t, uint32_t& c_pc, uint32_t& n_tests) {
    c_t = F32_MAX, c_pc = 0u, n_tests = 0u;
    const uint32_t q_end = load_const(&lc->seg_end);
    for (uint32_t q = load_const(&lc->seg_first); q < q_end; q += RSZ) {
      const uint4 q_lo = RT_FETCH_LO(q), q_hi = RT_FETCH_HI(q);
      const uint32_t w = __builtin_amdgcn_readfirstlane(q_hi.w), q_op = w & 0xffu;
      float t;
      if (q_op == OP_SPHERE) {
        V3 lo_o = ro;
        if (w & F_TRANSLATE) lo_o = vsub(ro, mk(u2f(q_lo.x), u2f(q_lo.y), u2f(q_lo.z)));
        if (w & F_MOVE) {
          const uint4 mv = RT_FETCH_LO(q + RSZ);
          lo_o = vsub(lo_o, smul(rtime, mk(u2f(mv.x), u2f(mv.y), u2f(mv.z))));
        }
        if (sphere_hit_t(lo_o, rd, u2f(q_lo.w), t_near, c_t, t)) c_t = t, c_pc = q | ((w >> 16) & 8u);
        n_tests += 1u;
        if (w & F_MOVE) q += RSZ;
      } else if (q_op == OP_RECT) {
        if (rect_hit_t(ro, rd, (w >> F_AXIS_SHIFT) & 3u, u2f(q_lo.x), u2f(q_lo.y), u2f(q_lo.z), u2f(q_lo.w), u2f(q_hi.x), t_near, c_t, t))
          c_t = t, c_pc = q | ((w >> 16) & 8u);
        n_tests += 1u;
      } else if (q_op == OP_PRISM) {
        uint32_t face = 0;
        if (prism_hit_t(q_lo, q_hi, ro, rd, t_near, c_t, t, face)) c_t = t, c_pc = q | face | ((w >> 16) & 8u);
        n_tests += 6u;
      }
    }
  };

  // The hit record (object.rs:61-71) of a finished ray, from the ray, t and the winning record -- what the walk would have built
  // at the hit and carried through the POPs, operation for operation: the wrappers around the record (flattened parent table,
  // outermost first) turn the ray into the one the primitive was tested with (object.rs:275-278, 357-361, 309-313, 505-508), the
  // primitive builds (p, normal) as its `hit` does, the wrappers take the record back out, innermost first (object.rs:279-282,
  // 365-369, 314-318, 249-252).  Same operands, same order: the same bits.
  auto rebuild_hit = [&](const uint32_t bpc, V3 ro, V3 rd, const float rtime, const float t, V3& p, V3& n, uint32_t& mat) {
    constexpr uint32_t NONE = 0xffffffffu;
    const uint32_t rpc = bpc & ~(RSZ - 1u);
    const uint32_t* par = load_const(&lc->parent);
    uint32_t w0 = par[rpc >> 5], w1 = NONE, w2 = NONE, w3 = NONE;  // enclosing PUSH records, innermost first (<= MAX_XFORM_DEPTH)
    if (w0 != NONE) w1 = par[w0];
    if (w1 != NONE) w2 = par[w1];
    if (w2 != NONE) w3 = par[w2];
#pragma unroll 1
    for (int k = 3; k >= 0; k--) {  // in
      const uint32_t w = k == 3 ? w3 : k == 2 ? w2 : k == 1 ? w1 : w0;
      if (w != NONE) {
        const uint4 x_lo = RT_FETCH_LO(w * RSZ), x_hi = RT_FETCH_HI(w * RSZ);
        const uint32_t kind = (x_hi.w >> F_KIND_SHIFT) & 7u;
        const V3 a = mk(u2f(x_lo.x), u2f(x_lo.y), u2f(x_lo.z));
        if (x_hi.w & F_PRE_TRANSLATE) ro = vsub(ro, mk(u2f(x_lo.w), u2f(x_hi.x), u2f(x_hi.y)));
        if (kind == XF_TRANSLATE) ro = vsub(ro, a);
        else if (kind == XF_ROTATE_Y) ro = rot_y(ro, -a.x, a.y), rd = rot_y(rd, -a.x, a.y);
        else if (kind == XF_SCALE) ro = vdiv(ro, a), rd = vdiv(rd, a);
        else if (kind == XF_MOVE) ro = vsub(ro, smul(rtime, a));
      }
    }
    const uint4 r_lo = RT_FETCH_LO(rpc), r_hi = RT_FETCH_HI(rpc);
    const uint32_t r_op = r_hi.w & 0xffu;
    mat = r_hi.z;
    if (r_op == OP_SPHERE) {
      const V3 off = mk(u2f(r_lo.x), u2f(r_lo.y), u2f(r_lo.z));
      V3 lo_o = ro;
      if (r_hi.w & F_TRANSLATE) lo_o = vsub(ro, off);
      if (r_hi.w & F_MOVE) {
        const uint4 mv = RT_FETCH_LO(rpc + RSZ);
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
#pragma unroll 1
    for (int k = 0; k < 4; k++) {  // out
      const uint32_t w = k == 3 ? w3 : k == 2 ? w2 : k == 1 ? w1 : w0;
      if (w != NONE) {
        const uint4 x_lo = RT_FETCH_LO(w * RSZ), x_hi = RT_FETCH_HI(w * RSZ);
        const uint32_t kind = (x_hi.w >> F_KIND_SHIFT) & 7u;
        const V3 a = mk(u2f(x_lo.x), u2f(x_lo.y), u2f(x_lo.z));
        if (kind == XF_TRANSLATE) p = vadd(p, a);
        else if (kind == XF_ROTATE_Y) p = rot_y(p, a.x, a.y), n = rot_y(n, a.x, a.y);
        else if (kind == XF_SCALE) p = vmul(p, a), n = vdiv(n, a);
        else if (kind == XF_FLIP) n = vneg(n);
        if (x_hi.w & F_PRE_TRANSLATE) p = vadd(p, mk(u2f(x_lo.w), u2f(x_hi.x), u2f(x_hi.y)));
      }
    }
  };

  for (;;) {
    uint32_t op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    const uint64_t m_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    const uint64_t m_slow = __builtin_amdgcn_ballot_w64(op >= OP_SPHERE && op <= OP_SLOW_LAST);
    const uint32_t n_busy = (uint32_t)__builtin_popcountll(m_box | m_slow);
    // ============================== SERVICE ======================================================
    // due when enough lanes are idle AND the service can do something for them: rays to hand out (T), or a full shade pass once
    // the lanes that stand at END are pushed.  (With rays waiting in S and X for company, "idle lanes" alone would call it
    // after every traversal step of a starved wave.)
    const uint32_t n_fin = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_END));
    const bool can_serve = (t_count + tn_count) != 0u || s_count + n_fin >= 64u || (TEX && x_count + n_fin >= 64u);
    if ((64u - n_busy >= tune.refill_min && can_serve) || n_busy == 0) {
      __builtin_amdgcn_s_setprio(RT_FULL_SERVICE_PRIO);
      if (COUNT) t_mark = RT_TICK();
      {  // (1) finish (depth is 0 again, so o / d are the ray's own)
        const bool fin = have_ray && op == OP_END;
        full_push_finished<TEX, COUNT>(qr, s_count, x_count, fin, o, d, time, best, hmat, ev_draws, r_strength, r_bounces, r_sample, r_xy,
                                       tr_out != nullptr, tr_d + ev_draws, tr_a + (cnt.aabb - tr_a0), tr_p + (cnt.prim - tr_p0));
        if (fin) have_ray = false;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (COUNT) t_fin += RT_TICK() - t_mark, n_serv++;
      // (2) shade: Material::scatter for 64 finished rays (color() loop body, lib.rs:73-97).  A path that goes on is pushed onto
      // T; one that ends books its sample colour and asks for its successor on N (the next sample of its work item, or a new item)
      auto shade_pass = [&](auto textured_tag, const uint32_t q0, uint32_t& count) {  // q0 = 0 (S) or XQ (X)
        constexpr bool TEXTURED = decltype(textured_tag)::value;
        constexpr uint32_t PASS_FEAT = TEXTURED ? FEAT : (FEAT & ~FEAT_TEXTURE);
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        const uint32_t take = count < 64u ? count : 64u;
        count -= take;
        if (COUNT) n_shade++, n_shade_lanes += take, t_mark2 = RT_TICK();
        uint32_t trd = 0, tra = 0, trp = 0;
        V3 so = mk(0.f, 0.f, 0.f), sd = so, strength = so;
        // no accum field: it is +0 whenever it is read (rt_pool.h PoolField; the host only routes scenes
        // here whose path strength stays finite and non-negative)
        V3 accum = so;
        float stime = 0.f;
        uint32_t bounces = 0, s = 0, x = 0, row = 0;
        bool lpt_on = false, live = false, ended = false, deferred = false;
        uint32_t hm = NO_HIT, hpc = NO_HIT, ev_next = 0;
        V3 p = so, n = so, sd0 = so, ray_o = so;
        float hit_t = 0.f;
        if (lane < take) {
          const uint32_t j = q0 + count + lane;  // pop: the top `take` entries
          hpc = SQ_LD_U(SQ_HITPC, j);
          ray_o = mk(SQ_LD_F(SQ_O, j), SQ_LD_F(SQ_O + 1, j), SQ_LD_F(SQ_O + 2, j));
          sd = mk(SQ_LD_F(SQ_D, j), SQ_LD_F(SQ_D + 1, j), SQ_LD_F(SQ_D + 2, j));
          stime = SQ_LD_F(SQ_TIME, j);
          hit_t = SQ_LD_F(SQ_T, j);
          // The texture value is fetched FIRST, while almost nothing of this pass is live: texture_eval
          // (Perlin turbulence / checker) is an out-of-line call and everything live across it adds to
          // the kernel's register count.
          uint4 mlo = make_uint4(0, 0, 0, 0), mhi = make_uint4(0, 0, 0, 0);
          V3 texval = mk(0.f, 0.f, 0.f);
          if (hpc != NO_HIT) {
            rebuild_hit(hpc, ray_o, sd, stime, hit_t, p, n, hm);
            const uint32_t m_off = load_const(&lc->mat_lds);
            if (m_off) mlo = s_mem[(m_off >> 4) + 2u * hm], mhi = s_mem[(m_off >> 4) + 2u * hm + 1u];
            else mlo = sc.mat[2 * hm], mhi = sc.mat[2 * hm + 1];
            texval = material_texture<PASS_FEAT>(sc, mlo, mhi, p);  // = albedo / emission colour for constant textures
          }
          strength = mk(SQ_LD_F(SQ_STRENGTH, j), SQ_LD_F(SQ_STRENGTH + 1, j), SQ_LD_F(SQ_STRENGTH + 2, j));
          bounces = SQ_LD_U(SQ_BOUNCES, j), s = SQ_LD_U(SQ_SAMPLE, j);
          RT_TL_BOUNCES(bounces);
          if (COUNT && tr_out) trd = SQ_LD_U(SQ_TRACE, j), tra = SQ_LD_U(SQ_TRACE + 1, j), trp = SQ_LD_U(SQ_TRACE + 2, j);
          const uint32_t xy = SQ_LD_U(SQ_XY, j);
          x = xy & 0xffffu, row = xy >> 16;
          SampleRng rng;
          rng.init(seed, (P.ny - 1u - row) * P.nx + x, s);
          rng.set_event(bounces + 1u);
          // continue after the medium draws of this event's traversal -- and after the attempts earlier passes made for this ray
          // (bits 28-31: how many passes tried; rt_pool.h RT_SCATTER_TRIES)
          const uint32_t evw = SQ_LD_U(SQ_EVDRAWS, j), tries = evw >> 28;
          rng.seek(evw & 0x0fffffffu);
          sd0 = sd;
          ended = true;
          V3 result = mk(0.f, 0.f, 0.f);
          if (hpc != NO_HIT) {
            const uint32_t kind = mhi.w & 0xffu;
            const float param = u2f(mlo.w);
            V3 emitted = mk(0.f, 0.f, 0.f);
            if (kind == MAT_DIFFUSE_LIGHT) emitted = smul(param, texval);  // material.rs:120-128
            accum = vadd(accum, vmul(strength, emitted));
            V3 nd = mk(0.f, 0.f, 0.f), att = texval;  // Lambertian / Isotropic: albedo(p)
            bool scattered = true;
            V3 rs = mk(0.f, 0.f, 0.f);
            if (kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC)
              deferred = !in_unit_sphere_tries(rng, (RT_SCATTER_TRIES && tries < 3u) ? (uint32_t)RT_SCATTER_TRIES : 0xffffffffu, rs);
            if (COUNT && !deferred) cnt.shaded++;
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
            if (deferred) {  // no direction yet: the ray goes back on its stack as it came, the stream position noted (rt_pool.h)
              ended = false;
              ev_next = ((evw & 0x0fffffffu) + rng.draws) | ((tries + 1u) << 28);
            } else if (scattered) {
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
          if (COUNT) trd += rng.draws;
          live = !ended && !deferred;
          if (ended) {
            float* sp = cm.scratch + 3ull * ((size_t)s * cm.pix_work + pixel_to_work(P, load_const(&lc->pm), x, row));
            RT_SCRATCH_STORE(sp, result);
            if (COUNT && tr_out) {
              uint32_t* tp = tr_out + 4ull * ((size_t)s * cm.pix_work + pixel_to_work(P, load_const(&lc->pm), x, row));
              tp[0] = bounces, tp[1] = trd, tp[2] = tra, tp[3] = trp;
            }
            s++;
          }
        }
        if (cm.lpt_samples) lpt_count(cm, lpt_on, lpt_on ? pixel_to_work(P, load_const(&lc->pm), x, row) >> 8 : 0u);
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live), m_end = __builtin_amdgcn_ballot_w64(ended);
        RT_TL_SHADE(take, m_live);
        if (live) {  // push onto T
          const uint32_t i = t_count + lane_rank(m_live);
          float c_t;
          uint32_t c_pc, c_n;
          hoist_eval(so, sd, stime, c_t, c_pc, c_n);
          if (COUNT) hoisted_prim += c_n, trp += c_n;  // (not cnt.prim: the lane may hold a ray of its own, whose trace is a difference of cnt)
          TQ_ST_F(TQ_SEG_T, i, c_t), TQ_ST_U(TQ_SEG_PC, i, c_pc);
          TQ_ST_F(TQ_O, i, so.x), TQ_ST_F(TQ_O + 1, i, so.y), TQ_ST_F(TQ_O + 2, i, so.z);
          TQ_ST_F(TQ_D, i, sd.x), TQ_ST_F(TQ_D + 1, i, sd.y), TQ_ST_F(TQ_D + 2, i, sd.z);
          TQ_ST_F(TQ_TIME, i, stime);
          TQ_ST_F(TQ_STRENGTH, i, strength.x), TQ_ST_F(TQ_STRENGTH + 1, i, strength.y), TQ_ST_F(TQ_STRENGTH + 2, i, strength.z);
          TQ_ST_U(TQ_BOUNCES, i, bounces), TQ_ST_U(TQ_SAMPLE, i, s);
          TQ_ST_U(TQ_XY, i, x | (row << 16));
          if (COUNT && tr_out) TQ_ST_U(TQ_TRACE, i, trd), TQ_ST_U(TQ_TRACE + 1, i, tra), TQ_ST_U(TQ_TRACE + 2, i, trp);
          if (COUNT) cnt.rays++;
        }
        if (ended) {  // push onto N: the next sample of this work item, or "a new item, please"
          const uint32_t i = n_count + lane_rank(m_end);
          NQ_ST_U(NQ_SAMPLE, i, (s == P.ns || s % cm.chunk == 0u) ? NQ_NEED_ITEM : s);
          NQ_ST_U(NQ_XY, i, x | (row << 16));
        }
        t_count += (uint32_t)__builtin_popcountll(m_live);
        n_count += (uint32_t)__builtin_popcountll(m_end);
        const uint64_t m_def = __builtin_amdgcn_ballot_w64(deferred);
        if (m_def != 0) {  // back onto the stack they came from, over entries this pass has consumed
          if (deferred) {
            const uint32_t i = q0 + count + lane_rank(m_def);
            SQ_ST_U(SQ_HITPC, i, hpc);
            SQ_ST_F(SQ_T, i, hit_t);
            SQ_ST_F(SQ_O, i, ray_o.x), SQ_ST_F(SQ_O + 1, i, ray_o.y), SQ_ST_F(SQ_O + 2, i, ray_o.z);
            SQ_ST_F(SQ_D, i, sd0.x), SQ_ST_F(SQ_D + 1, i, sd0.y), SQ_ST_F(SQ_D + 2, i, sd0.z);
            SQ_ST_F(SQ_TIME, i, stime);
            SQ_ST_U(SQ_EVDRAWS, i, ev_next);
            SQ_ST_F(SQ_STRENGTH, i, strength.x), SQ_ST_F(SQ_STRENGTH + 1, i, strength.y), SQ_ST_F(SQ_STRENGTH + 2, i, strength.z);
            SQ_ST_U(SQ_BOUNCES, i, bounces), SQ_ST_U(SQ_SAMPLE, i, s), SQ_ST_U(SQ_XY, i, x | (row << 16));
            if (COUNT && tr_out) SQ_ST_U(SQ_TRACE, i, trd), SQ_ST_U(SQ_TRACE + 1, i, tra), SQ_ST_U(SQ_TRACE + 2, i, trp);
          }
          count += (uint32_t)__builtin_popcountll(m_def);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_shade += RT_TICK() - t_mark2;
      };
      // (2b) camera rays for 64 paths that begin (par_cast closure, lib.rs:366-371: event 0): the next sample of the work item,
      // or the next work item of this wave's reservation (rt_pool.h)
      auto gen_pass = [&]() {
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        const uint32_t take = n_count < 64u ? n_count : 64u;
        n_count -= take;
        if (COUNT) n_gen++, n_gen_lanes += take, t_mark2 = RT_TICK();
        uint32_t st = ST_DEAD, s = 0, x = 0, row = 0;
        if (lane < take) {
          const uint32_t j = n_count + lane;
          s = NQ_LD_U(NQ_SAMPLE, j);
          const uint32_t xy = NQ_LD_U(NQ_XY, j);
          x = xy & 0xffffu, row = xy >> 16;
          st = s == NQ_NEED_ITEM ? ST_NEED_PIXEL : ST_GEN;
        }
        for (;;) {  // next work item (see rt_pool.h)
          const uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
          if (need == 0) break;
          if (w_next == w_end && !exhausted) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(queue, cm.work_block);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total_work) {
              exhausted = true;
              if (COUNT) t_exhausted = RT_TICK();
              RT_TL_EXHAUSTED(FPOOL - n_dead - n_count - take, t_count + tn_count, s_count, x_count);
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
              uint32_t w = w_next + r;
              w += w_delta;
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
#if RT_T_PRIORITY
          const uint32_t i = FPOOL - 1u - (tn_count + lane_rank(m_live));
#else
          const uint32_t i = t_count + lane_rank(m_live);
#endif
          float c_t;
          uint32_t c_pc, c_n;
          hoist_eval(so, sd, stime, c_t, c_pc, c_n);
          if (COUNT) hoisted_prim += c_n;
          TQ_ST_F(TQ_SEG_T, i, c_t), TQ_ST_U(TQ_SEG_PC, i, c_pc);
          TQ_ST_F(TQ_O, i, so.x), TQ_ST_F(TQ_O + 1, i, so.y), TQ_ST_F(TQ_O + 2, i, so.z);
          TQ_ST_F(TQ_D, i, sd.x), TQ_ST_F(TQ_D + 1, i, sd.y), TQ_ST_F(TQ_D + 2, i, sd.z);
          TQ_ST_F(TQ_TIME, i, stime);
          TQ_ST_F(TQ_STRENGTH, i, 1.f), TQ_ST_F(TQ_STRENGTH + 1, i, 1.f), TQ_ST_F(TQ_STRENGTH + 2, i, 1.f);
          TQ_ST_U(TQ_BOUNCES, i, 0u), TQ_ST_U(TQ_SAMPLE, i, s);
          TQ_ST_U(TQ_XY, i, x | (row << 16));
          if (COUNT && tr_out) TQ_ST_U(TQ_TRACE, i, rng.draws), TQ_ST_U(TQ_TRACE + 1, i, 0u), TQ_ST_U(TQ_TRACE + 2, i, c_n);
          if (COUNT) cnt.rays++;
        }
#if RT_T_PRIORITY
        tn_count += (uint32_t)__builtin_popcountll(m_live);
#else
        t_count += (uint32_t)__builtin_popcountll(m_live);
#endif
        n_dead += take - (uint32_t)__builtin_popcountll(m_live);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_gen += RT_TICK() - t_mark2;
        RT_TL_ALIVE(FPOOL - n_dead - n_count);
      };
      // full passes first; partial ones only when the lanes would have nothing to traverse (ONE call site per pass kind) -- and then
      // for EVERY stack that holds something: when only the first non-empty stack was served, the paths waiting in S and in X took
      // turns, each population idle while the other traversed (drain of a launch: r04a_tl_book2_*: > 64 live paths, 5 rays per generation)
      const bool drain_all = RT_DRAIN_ALL && n_busy == 0 && t_count + tn_count == 0;
      for (;;) {
        const bool any_part = drain_all || (n_busy == 0 && t_count + tn_count == 0);
        uint32_t which = s_count >= 64u ? 1u : (TEX && x_count >= 64u) ? 2u : n_count >= 64u ? 3u : 0u;
        if (which == 0u && any_part) which = n_count ? 3u : s_count ? 1u : (TEX && x_count) ? 2u : 0u;
        if (which == 0u) break;
        if (which == 1u) shade_pass(std::false_type{}, 0u, s_count);
        else if (which == 3u) gen_pass();
        else if (TEX) shade_pass(std::true_type{}, XQ, x_count);
      }
      if (COUNT) t_mark2 = RT_TICK();
#if RT_DRAIN_SHARE
      if (exhausted && load_const(&lc->cm.drain_share) != 0u) {  // ---- drain-phase work sharing (see DrainCtl) ----
        const uint32_t copy_rows = (COUNT && tr_out) ? (uint32_t)TQ_FIELDS : (uint32_t)TQ_TRACE;
        uint32_t* dq = uniform_ptr(g_slots + (size_t)blockIdx.x * wg_words + (size_t)n_waves * (FPOOL * FPOOL_FIELDS));
        if (n_dead == FPOOL) {  // run dry: register as hungry, adopt what waits on the workgroup's stack
          if (!hungry || __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT)) != 0u) {
            pool_lock(ctl + DC_LOCK, lane);
            const uint32_t n = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT));
            const uint32_t h = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) + (hungry ? 0u : 1u);  // hungry waves, this one included
            uint32_t k = (n + h - 1u) / h;  // an even share of what waits
            k = k < 64u ? k : 64u;
            if (lane < k) {
              const uint32_t src = n - k + lane;
              for (uint32_t f = 0; f < copy_rows; f++) Q_ST(Q_ROW(0u, f), lane, dq[f * DQ_CAP + src]);
            }
            if (lane == 0u) {
              __hip_atomic_store(ctl + DC_COUNT, n - k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(ctl + DC_HUNGRY, k ? h - 1u : h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            pool_unlock(ctl + DC_LOCK, lane);
            hungry = k == 0u;
            t_count = k, n_dead -= k;
          }
        } else if (w_next == w_end && tn_count == 0u && t_count >= 2u && __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) != 0u) {
          pool_lock(ctl + DC_LOCK, lane);  // holds rays while a wave of the workgroup has none: hand half of them over
          const uint32_t n = __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT));
          uint32_t k = t_count / 2u;
          k = k < 64u ? k : 64u;
          k = k < DQ_CAP - n ? k : DQ_CAP - n;
          if (lane < k) {
            const uint32_t src = t_count - k + lane, dst = n + lane;
            for (uint32_t f = 0; f < copy_rows; f++) dq[f * DQ_CAP + dst] = Q_LD(Q_ROW(0u, f), src);
          }
          if (lane == 0u) __hip_atomic_store(ctl + DC_COUNT, n + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          pool_unlock(ctl + DC_LOCK, lane);
          t_count -= k, n_dead += k;
          RT_TL_ALIVE(FPOOL - n_dead - n_count);
        }
      }
#endif
      {  // (3) refill from T
        const uint64_t m_idle = __builtin_amdgcn_ballot_w64(!have_ray);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(m_idle);
        const uint32_t t_all = t_count + tn_count;
        const uint32_t got = n_idle < t_all ? n_idle : t_all;
        if (got) {
          const uint32_t r = lane_rank(m_idle);
          const uint32_t take_c = got < t_count ? got : t_count;  // continuing rays first (rt_pool.h RT_T_PRIORITY)
          if (!have_ray && r < got) {
            const uint32_t i = r < take_c ? t_count - 1u - r : FPOOL - tn_count + (r - take_c);  // pop
            o = mk(TQ_LD_F(TQ_O, i), TQ_LD_F(TQ_O + 1, i), TQ_LD_F(TQ_O + 2, i));
            d = mk(TQ_LD_F(TQ_D, i), TQ_LD_F(TQ_D + 1, i), TQ_LD_F(TQ_D + 2, i));
            time = TQ_LD_F(TQ_TIME, i);
            r_strength = mk(TQ_LD_F(TQ_STRENGTH, i), TQ_LD_F(TQ_STRENGTH + 1, i), TQ_LD_F(TQ_STRENGTH + 2, i));
            r_xy = TQ_LD_U(TQ_XY, i), r_sample = TQ_LD_U(TQ_SAMPLE, i), r_bounces = TQ_LD_U(TQ_BOUNCES, i);
            seg_t = TQ_LD_F(TQ_SEG_T, i), seg_pc = TQ_LD_U(TQ_SEG_PC, i);
            if (COUNT && tr_out) tr_d = TQ_LD_U(TQ_TRACE, i), tr_a = TQ_LD_U(TQ_TRACE + 1, i), tr_p = TQ_LD_U(TQ_TRACE + 2, i);
            pc = 0, best = F32_MAX, hmat = NO_HIT, ev_draws = 0;
            depth = 0, tag = 0, nhits = 0, root_hits = 0;
            bmode = 0, t_lo = t_near;
            if (COUNT) tr_a0 = cnt.aabb, tr_p0 = cnt.prim;
            have_ray = true;
          }
          t_count -= take_c, tn_count -= got - take_c;
          if (COUNT) n_refill++;
          RT_TL_REFILL(got);
        }
      }
      // 1/d and the current record of EVERY lane are (re)derived here -- the same bits for a lane that kept its ray -- so
      // that these 11 registers are dead across the passes above, which need the room (idle lanes: a stale but valid pc)
      inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
      cur_lo = RT_FETCH_LO(pc), cur_hi = RT_FETCH_HI(pc);
      if (COUNT) t_refill += RT_TICK() - t_mark2;
      if (COUNT) t_serv += RT_TICK() - t_mark;
      RT_TL_SERVICE();
      __builtin_amdgcn_s_setprio(RT_FULL_BOX_PRIO);
      if (n_dead == FPOOL) {
#if RT_DRAIN_SHARE
        if (load_const(&lc->cm.drain_share) != 0u) {  // leave when every wave of the workgroup has run dry and nothing waits to be adopted
          if (hungry && __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_HUNGRY)) == n_waves &&
              __builtin_amdgcn_readfirstlane(pool_lds_ld(ctl + DC_COUNT)) == 0u) break;
          __builtin_amdgcn_s_sleep(127);  // (poll every ~3.5 us)
#ifdef RT_DRAIN_WATCHDOG
          if (++drain_polls > (uint32_t)(RT_DRAIN_WATCHDOG)) __builtin_trap();  // a regression of the invariant above shows as an error, not as a hung GPU
#endif
          continue;
        }
#endif
        break;
      }
      if (__builtin_amdgcn_ballot_w64(have_ray) == 0) continue;
      op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    }
    // ============================== TRAVERSE ======================================================
    // box runs and slow passes alternate in this inner loop until a service is due (rt_pool.h)
    for (;;) {
#define RT_R_PIXEL ((load_const(&lc->P.ny) - 1u - (r_xy >> 16)) * load_const(&lc->P.nx) + (r_xy & 0xffffu))
#define RT_R_EVENT (r_bounces + 1u)
#define RT_PHASE_PRIO 1  // (the pool schedule sets wave priorities per phase: see RT_FULL_SLOW_PRIO)
#define RT_CENSUS_HERE 1
#include "rt_full_traverse.inc"
#undef RT_CENSUS_HERE
#undef RT_PHASE_PRIO
#undef RT_R_PIXEL
#undef RT_R_EVENT
    op = have_ray ? (cur_hi.w & 0xffu) : 0xffu;
    const uint32_t busy = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op >= OP_BOX && op <= OP_SLOW_LAST));
    if (64u - busy >= tune.refill_min || busy == 0) break;
    }
  }
  RT_TL_DONE();
  if (COUNT) {
    atomicAdd(&counters[0], (unsigned long long)cnt.aabb);
    atomicAdd(&counters[1], (unsigned long long)cnt.prim + hoisted_prim);
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
#ifdef RT_CENSUS
    }
    if (lane < 16u) atomicAdd(&counters[32u + lane], census[0]), atomicAdd(&counters[48u + lane], census[1]);
    if (lane == 0) {
#endif
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], t_serv - t_shade), atomicAdd(&counters[18], t_box),
          atomicAdd(&counters[19], t_slow);
      // wave timeline (rt_pool.h)
      const unsigned long long dur = RT_TICK() - t_start, exh = t_exhausted - t_start;
      atomicMax(&counters[24], dur), atomicAdd(&counters[25], dur), atomicAdd(&counters[26], 1ull);
      atomicMax(&counters[27], (1ull << 62) - exh), atomicAdd(&counters[28], exh), atomicMax(&counters[29], exh);
    }
  }
}

#undef RT_IN_LDS
#undef RT_FETCH_LO
#undef RT_FETCH_HI
#undef Q_ROW
#undef Q_LD
#undef Q_ST
#undef TQ_LD_U
#undef TQ_LD_F
#undef TQ_ST_U
#undef TQ_ST_F
#undef SQ_LD_U
#undef SQ_LD_F
#undef SQ_ST_U
#undef SQ_ST_F
#undef NQ_LD_U
#undef NQ_ST_U

}  // namespace rtg
