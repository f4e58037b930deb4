// rt_pool.h -- "ray pool" kernel for lean scenes (spheres under Bvhs, constant textures): the tuned
// path for the book-1 north-star workload.  Same arithmetic as rt_trace.h; the SCHEDULE is built for
// CDNA4's 64-wide waves and 160 KB LDS:
//
//  * every wave owns a private pool of POOL = 156 path slots.  A slot holds a whole path state: the ray
//    (o, d) in LDS when the program leaves room (RAY_LDS), best hit, strength, bounce / sample counters
//    and pixel as SoA rows in a per-wave region of global memory sized to stay in the L2s.
//  * the wave's 64 lanes only TRAVERSE: a lane holds (o, d, 1/d, best, pc, current record) of one
//    slot's ray in registers and walks the flat program (staged in LDS).  When its ray reaches END the
//    lane writes (best, best_pc) back to the slot, pushes the slot on the wave's S- or E-list and refills
//    from the T-list -- lanes do not wait for each other's paths (wave-ballot compaction of the
//    active-ray stream: ballot + mbcnt prefix ranks, no atomics because the lists are wave-private).
//  * when 64 slots wait on the S-list the wave runs ONE full-width SCATTER pass over them (hit record,
//    Material::scatter), when 64 wait on the E-list one END pass (book the sample, next work item,
//    camera ray); both write the new rays into the slots and push them on the T-list.  Shading
//    therefore always runs 64 lanes wide instead of "whoever happened to finish".
//  * lanes that reach a SPHERE record park until enough of them wait, so the sqrt/divide sequence
//    never runs for a handful of lanes; box steps run in a tight loop between schedule decisions.
//  * workgroups are persistent (one 16-wave workgroup per CU); work items -- one sample of one pixel --
//    come from one global counter, 256 at a time, in an order that follows measured cost (LptQueue);
//    a second kernel folds each pixel's samples in order.
//
// Scheduling never changes results: every (pixel, sample, event) has its own RNG stream and a pixel's
// samples are folded in sample order.
#pragma once
#include <type_traits>

#include "rt_trace.h"

namespace rtg {

constexpr uint32_t ST_NEED_PIXEL = 0, ST_GEN = 1, ST_TRAV = 2, ST_SHADE = 3, ST_DEAD = 4;
constexpr uint32_t NO_HIT = 0xffffffffu;

RT_DEV uint32_t lane_rank(uint64_t mask) {  // number of set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

RT_DEV uint32_t fdiv(uint32_t n, const FastDiv f) {
#ifdef RT_NO_FASTDIV
  return n / f.d;  // (measurement switch: the plain divisions of rounds 1-2)
#endif
  if (f.d == 1u) return n;  // (wave-uniform)
  const uint32_t q = __umulhi(n, f.m);
  return (((n - q) >> 1) + q) >> f.s;
}

// work item -> pixel.  Work items enumerate this rank's tiles (tile % nranks == rank) in order, each
// tile as 8x8 blocks, so the 64 items a wave grabs at start form one coherent 8x8 block.
RT_DEV bool work_to_pixel(const DevParams& P, const PixMap& M, uint32_t w, uint32_t& x, uint32_t& row) {
  const uint32_t px_per_tile = P.tile_w * P.tile_h;
  uint32_t k = fdiv(w, M.d_px_per_tile), r = w - k * px_per_tile;
  uint32_t tile = P.rank + k * P.nranks;
  uint32_t ty = fdiv(tile, M.d_tiles_x), tx = tile - ty * M.tiles_x;
  uint32_t blocks_x = P.tile_w >> 3;
  uint32_t b = r >> 6, l = r & 63u;
  uint32_t by = fdiv(b, M.d_blocks_x), bx = b - by * blocks_x;
  x = tx * P.tile_w + bx * 8u + (l & 7u);
  row = ty * P.tile_h + by * 8u + (l >> 3);
  return x < P.nx && row < P.ny;
}

// A wave-uniform pointer derived from threadIdx (the wave's own memory): readfirstlane tells the compiler it is uniform
template <typename T>
RT_DEV T* uniform_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

#define RT_TICK() (COUNT ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)

// Drain timeline (diagnostic builds only, -DRT_TIMELINE; tools/timeline.py): every wave of a pool kernel leaves one 16-dword
// record in the buffer counters[31] points to -- s_memrealtime (the chip-wide 100 MHz clock) at its start [0], when it sees the
// work queue empty [1], when its live paths fall to <= 64 [3] / <= 8 [10] / <= 1 [11], when it is done [4]; its live paths [2] and
// list sizes [12..14] at exhaustion; after exhaustion: shade passes [5], services [6], shade passes that held a ray at bounce
// >= 16 [7], rays handed to lanes [9], lanes of the shade passes [8].  Production builds compile none of it.
#ifdef RT_TIMELINE
#define RT_TL_NOW() ((uint32_t)__builtin_amdgcn_s_memrealtime())
#define RT_TL_DECL(pool_) \
  uint32_t* tl = nullptr; \
  uint32_t tl_passes = 0, tl_serv = 0, tl_deep = 0, tl_rays = 0, tl_lanes = 0, tl_stage = 0; \
  bool tl_deep_now = false; \
  { const unsigned long long p_ = counters[31]; \
    if (p_) tl = reinterpret_cast<uint32_t*>(p_) + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16u; } \
  if (tl && (threadIdx.x & 63u) == 0u) tl[0] = RT_TL_NOW()
#define RT_TL_EXHAUSTED(alive_, t_, s_, e_) \
  if (tl && (threadIdx.x & 63u) == 0u) tl[1] = RT_TL_NOW(), tl[2] = (alive_), tl[12] = (t_), tl[13] = (s_), tl[14] = (e_)
#define RT_TL_BOUNCES(b_) tl_deep_now = (b_) >= 16u
#define RT_TL_SHADE(take_, m_live_) \
  if (exhausted) { \
    tl_passes++, tl_lanes += (take_); \
    if (__builtin_amdgcn_ballot_w64(tl_deep_now && (threadIdx.x & 63u) < (take_)) != 0) tl_deep++; \
  } \
  tl_deep_now = false
#define RT_TL_ALIVE(alive_) \
  if (tl && exhausted) { \
    const uint32_t a_ = (alive_); \
    if (tl_stage < 1u && a_ <= 64u) { tl_stage = 1u; if ((threadIdx.x & 63u) == 0u) tl[3] = RT_TL_NOW(); } \
    if (tl_stage < 2u && a_ <= 8u) { tl_stage = 2u; if ((threadIdx.x & 63u) == 0u) tl[10] = RT_TL_NOW(); } \
    if (tl_stage < 3u && a_ <= 1u) { tl_stage = 3u; if ((threadIdx.x & 63u) == 0u) tl[11] = RT_TL_NOW(); } \
  }
#define RT_TL_REFILL(got_) if (exhausted) tl_rays += (got_)
#define RT_TL_SERVICE() if (exhausted) tl_serv++
#define RT_TL_DONE() \
  if (tl && (threadIdx.x & 63u) == 0u) tl[4] = RT_TL_NOW(), tl[5] = tl_passes, tl[6] = tl_serv, tl[7] = tl_deep, tl[8] = tl_lanes, tl[9] = tl_rays
#else
#define RT_TL_DECL(pool_)
#define RT_TL_EXHAUSTED(alive_, t_, s_, e_)
#define RT_TL_BOUNCES(b_)
#define RT_TL_SHADE(take_, m_live_)
#define RT_TL_ALIVE(alive_)
#define RT_TL_REFILL(got_)
#define RT_TL_SERVICE()
#define RT_TL_DONE()
#endif


#ifndef RT_UNIFORM_SLOT_PTR
#define RT_UNIFORM_SLOT_PTR 1
#endif
#ifndef RT_GATED_SERVICE
#define RT_GATED_SERVICE 1
#endif
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 156  // what book-1's image leaves room for in LDS (36 B per slot and wave: 163 584 of 163 840 B); C2: 136 7.95 ms, 144 7.81, 148 7.77, 152 7.75, 156 7.72
#endif
constexpr uint32_t POOL = RT_POOL_SLOTS;           // path slots per wave: 64 in lanes + 92 waiting (a wait list reaches 64 while the lanes drain)
constexpr uint32_t POOL_FIELDS = 17;     // dwords per slot (SoA: field f of slot j at [f * POOL + j]); 14 in chunk mode
constexpr uint32_t WORK_BLOCK = 256;     // work items a wave reserves per global atomic (2048 cost 20 % on C2: ~6 blocks per wave = a coarse tail)
constexpr uint32_t SLOT_NEED_PIXEL = 0xfffffffeu;  // best_pc marker: slot holds no ray yet
constexpr uint32_t SLOT_ENDED = 0xfffffffdu;       // best_pc marker: path ended in a SCATTER pass (its colour is accum = +0)

enum PoolField : uint32_t {
  PF_O = 0, PF_D = 3, PF_BEST = 6, PF_BEST_PC = 7, PF_STRENGTH = 8, PF_BOUNCES = 11, PF_SAMPLE = 12, PF_XY = 13, PF_COL = 14,
};
// There is no accum field.  color()'s `accum = accum + strength * emitted` (lib.rs:76) adds
// strength * 0 at every hit on a scattering material (material.rs:126), and the only emitter, a
// DiffuseLight, ends the path.  With every albedo component in [0, 4] (Perlin turbulence is <= 2) the
// strength stays finite and non-negative over 51 bounces, strength * 0 = +0 and accum is +0 whenever
// it is read; the flattener routes anything else to the baseline kernel (FEAT_WIDE_ALBEDO).

// Sample-chunk mode.  One lane per pixel cannot fill the chip when a rank owns few pixels (8-GPU
// shards, small images).  A work item is then (pixel, chunk of `chunk` consecutive samples); every
// sample colour goes to an HBM scratch laid out [sample][pixel-work-index] and fold_samples_kernel
// adds them per pixel IN SAMPLE ORDER afterwards -- the same left fold as lib.rs:365-374, bit for bit.
constexpr uint32_t LPT_BLOCK = 256;   // pixels per cost block = WORK_BLOCK
constexpr uint32_t LPT_CLASSES = 64;  // one per lane
constexpr uint32_t LPT_CTL = 80;      // count[64], [64] = blocks filed so far
struct LptQueue {       // lives in device memory: the kernels touch it once per 256-item reservation
  uint32_t* cost;       // [n_blocks] scatter events per block during phase 1
  uint32_t* ctl;        // [LPT_CTL]
  uint32_t* list;       // [LPT_CLASSES][n_blocks] blocks of each class, in filing order
  uint32_t n_blocks;
  uint32_t phase1;      // chunks in natural order
  uint32_t phase2_base; // first phase-2 work item = phase1 * pix_work
  uint32_t span;        // phase-2 items per block = LPT_BLOCK * (n_chunks - phase1)
  uint32_t mode;        // 1: block-major (all chunks of a block back to back); 2: per class, chunk-major
  uint32_t shift;       // classes are merged in groups of 1 << shift
};

struct ChunkMode {
  float* scratch;      // null = off: a slot folds its pixel's samples itself
  uint32_t chunk;      // samples per work item
  uint32_t n_chunks;   // work items per pixel
  uint32_t pix_work;   // pixel work items of this rank (tiles x tile area, incl. out-of-image padding)
  // Cost-ordered queue ("longest processing time first"), lpt != null.  The end of a frame is a chain
  // problem, not a throughput problem: a path that runs into the bounce cap (glass: ~0.2 % of book-1's
  // samples) needs 51 dependent traverse + scatter generations, and the ones that START late finish
  // ~1.2 ms after the queue is empty (measured: fixed cost per C2 launch 1.69 ms at cap 50, 0.46 ms at
  // cap 4).  So the first `phase1` chunks run in natural order while every scatter event is counted per
  // 256-pixel block; during the last of them each reservation files its block under one of 64
  // logarithmic cost classes; the remaining chunks run block-major, classes in descending cost order:
  // long-path-prone blocks first, sky last.  Order never changes results (per-event RNG streams,
  // ordered fold).
  const LptQueue* lpt;   // descriptor in device memory (a by-value copy inside the kernel arguments cost 19 more SGPR spills and
                         // 8 % of a C2 launch); written on the LAUNCH STREAM by write_lpt_descriptor, so launches stay ordered
  uint32_t lpt_samples;  // samples [0, lpt_samples) of every pixel belong to phase 1 (0 = off)
  uint32_t lpt_deep;     // a scatter event counts towards its block's cost from this bounce on
  // Sample passes (bounded scratch): a launch renders the samples [s_begin, P.ns) of every pixel -- P.ns of the LAUNCH is
  // the end of its pass, not the frame's sample count -- and `scratch` is biased so that sample s of pixel work index w
  // still sits at scratch[3 * (s * pix_work + w)].  One pass (s_begin = 0, P.ns = ns) unless the frame's per-sample
  // colours exceed the scratch budget (rtg_launch.inc samples_per_pass); the fold kernel carries the running per-pixel sum
  // from pass to pass in the framebuffer itself, so the fold stays the reference's left fold (lib.rs:365-374).
  uint32_t s_begin;
  // Work items a wave reserves per global atomic: WORK_BLOCK (256) for frames that fill the chip several times over (and always
  // with the cost-ordered queue, whose blocks are 256 pixels); 64 = one item per lane for small frames, where a 256-item
  // reservation would leave most waves without work and make the others run four generations one after another
  // (rtg_launch.inc pool_geometry).  pix_work is a multiple of 256, so a reservation never straddles a chunk.
  uint32_t work_block;
  uint32_t drain_share;  // full-feature pool kernel: drain-phase work sharing (rt_pool_full.h RT_DRAIN_SHARE), 0 = off
};
RT_DEV uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// phase 1: lanes with `on` add one to their block's cost -- one atomic per distinct block of the wave
// (a pass mostly holds samples of one or two blocks; 64 same-address atomics cost Cornell 25 %)
RT_DEV void lpt_count(const ChunkMode& cm, bool on, uint32_t blk) {
  uint64_t todo = __builtin_amdgcn_ballot_w64(on);
  while (todo) {
    const uint32_t b0 = __builtin_amdgcn_readlane(blk, (uint32_t)__builtin_ctzll(todo));
    const uint64_t same = __builtin_amdgcn_ballot_w64(on && blk == b0);
    if (on && blk == b0 && lane_rank(same) == 0u) atomicAdd(&cm.lpt->cost[b0], (uint32_t)__builtin_popcountll(same));
    todo &= ~same;
  }
}

// A wave reserved work items [base, base + WORK_BLOCK): they all belong to one chunk and one 256-pixel
// block (pix_work is a multiple of 256).  Returns the chunk; item w is pixel work index w + delta.
RT_DEV uint32_t lpt_reservation(const ChunkMode& cm, uint32_t base, uint32_t lane, uint32_t& delta, bool& ready) {
  const LptQueue* q = cm.lpt;
  if (q == nullptr || base < q->phase2_base) {
    const uint32_t c = base / cm.pix_work;
    delta = 0u - c * cm.pix_work;
    if (q != nullptr && c + 1u == q->phase1) {  // last natural-order chunk: file this block under its cost class
      const uint32_t blk = (base + delta) >> 8;
      if (lane == 0u) {
        const uint32_t v = ld_agent(q->cost + blk) + 1u;            // class = 4 per octave of (cost + 1), 0 = most expensive
        const uint32_t e = 31u - (uint32_t)__builtin_clz(v);
        const uint32_t k = 4u * e + (e >= 2u ? ((v >> (e - 2u)) & 3u) : ((v << (2u - e)) & 3u));
        const uint32_t cls = (LPT_CLASSES - 1u - (k > LPT_CLASSES - 1u ? LPT_CLASSES - 1u : k)) >> q->shift;
        const uint32_t pos = atomicAdd(&q->ctl[cls], 1u);
        __hip_atomic_store(q->list + (size_t)cls * q->n_blocks + pos, blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&q->ctl[LPT_CLASSES], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return c;
  }
  const uint32_t n = q->n_blocks;
  if (!ready) {  // once per wave: an agent-scope acquire drops the CU's L1, which the path slots live in
    while (__hip_atomic_load(&q->ctl[LPT_CLASSES], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n) __builtin_amdgcn_s_sleep(8);
    ready = true;
  }
  const uint32_t span = q->span, w2 = base - q->phase2_base;
  const uint32_t r = w2 / span, rem = w2 - r * span;  // r-th block of the class order, chunk phase1 + rem / 256
  const uint32_t cnt = q->ctl[lane];  // ctl[] and list[] no longer change: ordinary cached loads
  uint32_t incl = cnt;
  for (int k = 1; k < 64; k <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, k, 64);
    if ((int)lane >= k) incl += t;
  }
  const uint32_t cls = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(incl > r));
  const uint32_t before = __builtin_amdgcn_readlane(incl - cnt, cls);
  uint32_t idx = r - before, c = rem >> 8;
  if (q->mode == 2u) {  // inside the class: chunk-major, blocks in filing order
    const uint32_t per_chunk = __builtin_amdgcn_readlane(cnt, cls) * LPT_BLOCK, rem_c = w2 - before * span;
    c = rem_c / per_chunk;
    idx = (rem_c - c * per_chunk) >> 8;
  }
  const uint32_t blk = q->list[(size_t)cls * n + idx];
  delta = blk * LPT_BLOCK - base;  // reservations are block-aligned
  return q->phase1 + c;
}

// The per-sample colours are written once and read once by the fold kernel, 0.58 GB per C2 frame: stream them
// past the L2, which the path slots want to themselves.
#ifndef RT_NT_SCRATCH
#define RT_NT_SCRATCH 1
#endif
#if RT_NT_SCRATCH
#ifndef RT_SCRATCH_X3
#define RT_SCRATCH_X3 1
#endif
#if RT_SCRATCH_X3
typedef float f32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
#define RT_SCRATCH_STORE(p_, v_) __builtin_nontemporal_store(f32x3_a4{(v_).x, (v_).y, (v_).z}, reinterpret_cast<f32x3_a4*>(p_))  // one global_store_dwordx3
#else
#define RT_SCRATCH_STORE(p_, v_) (__builtin_nontemporal_store((v_).x, (p_)), __builtin_nontemporal_store((v_).y, (p_) + 1), __builtin_nontemporal_store((v_).z, (p_) + 2))
#endif
#define RT_SCRATCH_LOAD(p_) __builtin_nontemporal_load(p_)
#else
#define RT_SCRATCH_STORE(p_, v_) ((p_)[0] = (v_).x, (p_)[1] = (v_).y, (p_)[2] = (v_).z)
#define RT_SCRATCH_LOAD(p_) (*(p_))
#endif

// inverse of work_to_pixel
RT_DEV uint32_t pixel_to_work(const DevParams& P, const PixMap& M, uint32_t x, uint32_t row) {
  uint32_t tx = fdiv(x, M.d_tile_w), ty = fdiv(row, M.d_tile_h);
  uint32_t k = fdiv(ty * M.tiles_x + tx, M.d_nranks);  // this rank's k-th tile
  uint32_t lx = x - tx * P.tile_w, ly = row - ty * P.tile_h;
  uint32_t b = (ly >> 3) * (P.tile_w >> 3) + (lx >> 3);
  return k * (P.tile_w * P.tile_h) + b * 64u + (ly & 7u) * 8u + (lx & 7u);
}

// Launch constants live in device memory, not in the kernel arguments: camera, frame parameters and the chunk / queue
// description are only read inside the SCATTER and END passes, and by-value copies would sit in (spilled) SGPRs across the
// traversal loops -- SGPR spill code is VALU code (v_readlane / v_writelane), and 19 more spilled SGPRs once cost 8 % of a
// C2 launch.  load_const reads through a laundered constant-address-space pointer: the loads stay scalar (s_load) and INSIDE
// the pass (the compiler can neither hoist them out of the main loop nor treat them as loop-invariant).
struct LaunchConsts {
  DevCamera cam;
  DevParams P;
  ChunkMode cm;
  PixMap pm;
  const uint32_t* parent;       // full-feature pool kernel: per program record, the index of the PUSH record of the innermost wrapper around it
                                // (0xffffffff: none) -- what rebuild_hit replays (rt_pool_full.h)
  uint32_t mat_lds;             // full-feature kernels: byte offset of the material records' copy in LDS (behind the program and, in the pool kernel, its
                                // control words), 0 = they do not fit: fetched from global memory.  A shade pass's material fetch is a dependent load
                                // behind the hit's material index: from LDS it costs ~130 cycles instead of a trip to L2 (~1 500)
  uint32_t seg_first, seg_end;  // full-feature pool kernel: the records of the hoisted segment as program counters (flat_scene.h OP_SEG);
                                // seg_end = 0: no segment, or hoisting switched off -- OP_SEG is then stepped over
  uint32_t p2_lists;            // pool-2 kernel (rt_pool2.h): byte offset of the waves' id lists in LDS
};
typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;
template <typename T>
RT_DEV T load_const(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword-sized");
  const_u32_ptr w = (const_u32_ptr)(uintptr_t)p;
  asm volatile("" : "+s"(w));
  T c;
  uint32_t* dst = reinterpret_cast<uint32_t*>(&c);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4u; i++) dst[i] = w[i];
  return c;
}
// (also resets the work-queue head of the launch: one stream operation less per sample pass than a hipMemsetAsync)
__global__ void write_launch_consts(LaunchConsts* dst, LaunchConsts v, unsigned long long* queue) { *dst = v, *queue = 0ull; }

// stream-ordered update of the descriptor (a kernel argument by value: no host buffer has to outlive the call)
__global__ void write_lpt_descriptor(LptQueue* dst, LptQueue v) { *dst = v; }

// second pass of the chunk mode: ordered fold of the per-sample colours (vec3.rs:195-203, lib.rs:374).  Samples
// [cm.s_begin, P.ns) of this pass are added IN ORDER to the running sum of the earlier passes -- (0, 0, 0) for the first
// pass (vec3.rs:197), else what the previous pass left in the framebuffer, the same f32 bits the sequential fold would hold
// at that point -- and the last pass divides by the frame's sample count `ns_frame` (lib.rs:374).
__global__ void fold_samples_kernel(DevParams P, ChunkMode cm, PixMap pm, float* __restrict__ out, uint32_t ns_frame) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= cm.pix_work) return;
  uint32_t x, row;
  if (!work_to_pixel(P, pm, w, x, row)) return;
  float* o = out + 3ull * ((size_t)row * P.nx + x);
  V3 col = mk(0.f, 0.f, 0.f);
  if (cm.s_begin != 0u) col = mk(o[0], o[1], o[2]);
  for (uint32_t s = cm.s_begin; s < P.ns; s++) {
    const float* c = cm.scratch + 3ull * ((size_t)s * cm.pix_work + w);
    col = vadd(col, mk(RT_SCRATCH_LOAD(c), RT_SCRATCH_LOAD(c + 1), RT_SCRATCH_LOAD(c + 2)));
  }
  if (P.ns == ns_frame) col = sdiv(col, (float)ns_frame);
  o[0] = col.x, o[1] = col.y, o[2] = col.z;
}

struct PoolTuning {
  uint32_t refill_min;   // idle lanes before the wave services (finish / shade / refill)
  uint32_t sphere_min;   // parked lanes before a sphere pass
  uint32_t box_leave;    // lanes leaving the BOX state before a box run re-evaluates the schedule
  uint32_t run_ahead;    // full-feature kernel: records a slow pass may execute per lane ...
  uint32_t run_ahead_min;  // ... while at least this many lanes sit on slow records
  uint32_t gather_min;   // full-feature kernel: lanes waiting at an F_GATHER record before they are released together
};

// LDS image of the program (staged variant): variable-size records, pc = the record's absolute LDS address (a step forms
// no base + offset sum); every record starts with the dword pair a traversal step needs first:
//   BOX     56 B  dw0 skip pc (absolute)   dw1 op/flags | LDS_BOX_BIT
//                 dw2-7  (min.x, max.x) (min.y, max.y) (min.z, max.z)     the pairs a ray with 1/d >= 0 wants
//                 dw8-13 (max.x, min.x) (max.y, min.y) (max.z, min.z)     ... and one with 1/d < 0
//   SPHERE  24 B  dw0 material index       dw1 op/flags (F_TRANSLATE, F_FLIP, material kind)    dw2-5 offset.xyz, radius
//   END      8 B  dw0 -                    dw1 OP_END      (+ 56 B of padding: a step's plane loads stay inside the image)
// A BOX that passes continues at pc + 56 (its left child, or its leaf's SPHERE), a SPHERE at pc + 24; "is this a BOX" is
// the sign bit of the flag word (one compare, no mask).  book-1: 969 x 56 + 485 x 24 + 64 = 66.0 KB (81.5 KB with
// uniform 56-byte records): the 15.5 KB pay for the slots' (best, best_pc) in LDS.
// Aabb::hit swaps (t0, t1) when 1/d < 0 (aabb.rs:20-23).  A lane reads ITS (near plane, far plane) pair
// per axis with one ALIGNED 8-byte load (ds_read_b64: 2 LDS cycles per wave, against 4 for the
// ds_read2_b32 an unaligned pair needs); the offset is fixed per ray, so the swap costs no instruction per step.
constexpr uint32_t LDS_BOX_BYTES = 56, LDS_SPHERE_BYTES = 24, LDS_END_BYTES = 8 + 56;
constexpr uint32_t LDS_BOX_BIT = 0x80000000u;
// 4-wide image (template parameter WIDE, option `bvh4`; built by rtg_launch.inc build_wide_image; offsets relative to the image):
//   NODE   208 B  dw0 parent node (WIDE_NO_PARENT: the root)   dw1 LDS_BOX_BIT | OP_BOX | level << 8 | children << 12 | leaf mask << 16
//                 dw2-3 four child offsets / 8 (u16 each)       dw4-51 four boxes, each as the BOX record above: the three
//                 (min, max) pairs, then the three (max, min) pairs (a lane reads ITS pair per axis with one ds_read_b64)
//   SPHERE  32 B  dw0-5 as above                                dw6 parent node
// A step tests a node's four boxes at once (the children's and grandchildren's boxes of the reference's binary node) with the
// ray's CURRENT best and keeps the four results in 4 bits of a per-lane register (one nibble per level, <= 8 levels); the
// children are then visited left to right, as the reference visits them.  A box tested earlier is tested against a best that
// is no smaller than the reference's at that point, so every leaf the reference tests is tested here too, in the same order:
// the same closest hit, the same image -- other counter values (more sphere tests, fewer dependent steps).
constexpr uint32_t WIDE_NODE_BYTES = 208, WIDE_SPHERE_BYTES = 32, WIDE_NO_PARENT = 0xffffffffu;
typedef __attribute__((address_space(3))) const char* lds_cptr;

// byte offset of every record inside the image (host side, at scene creation); returns the image size (multiple of 16).
// Only lean programs (BOX / SPHERE / END) have an image.
inline uint32_t lds_image_offsets(const uint32_t* op_words, size_t n, uint32_t* off) {
  uint32_t at = 0;
  for (size_t i = 0; i < n; i++) {
    off[i] = at;
    const uint32_t op = op_words[i] & 0xffu;
    at += op == OP_BOX ? LDS_BOX_BYTES : (op == OP_SPHERE ? LDS_SPHERE_BYTES : LDS_END_BYTES);
  }
  return (at + 15u) & ~15u;
}

// dynamic LDS bytes for a workgroup of `waves` waves: [program image | materials] [T-, S-, E-list] [slot rays] [slot best]
// [slot best_pc].  The HOT slot fields -- the ray (o, d): written by the SCATTER and END passes, read by the refill and the
// SCATTER pass; (best, best_pc): written when a ray finishes, read by the pass that shades it -- live in LDS (`hot_lds`,
// when the program leaves room); the cold ones (strength, bounces, sample, pixel) in a per-wave SoA region of global memory
// sized to stay in the L2s.
inline size_t pool_lds_bytes(uint32_t image_bytes, uint32_t n_mat, uint32_t waves, bool stage_program, bool hot_lds) {
  size_t b = stage_program ? ((size_t)image_bytes + (size_t)n_mat * 16) : 0;  // records + (albedo | emission, param) per material
  b += (size_t)waves * POOL * 3 * 2;  // T-, S- and E-list (u16 slot ids)
  b = (b + 15) & ~(size_t)15;
  if (hot_lds) b += (size_t)waves * POOL * (6 * sizeof(float) + sizeof(float) + (stage_program ? 2 : 4));
  return (b + 15) & ~(size_t)15;
}

// spin lock + relaxed load on LDS control words of a workgroup (rt_pool_full.h: drain-phase work sharing)
RT_DEV uint32_t pool_lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void pool_lock(uint32_t* lock_word, uint32_t lane) {
  if (lane == 0u) {
    for (;;) {
      uint32_t expect = 0u;
      if (__hip_atomic_compare_exchange_strong(lock_word, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
RT_DEV void pool_unlock(uint32_t* lock_word, uint32_t lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (waits for the record loads and stores of every lane)
  if (lane == 0u) __hip_atomic_store(lock_word, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define RT_AS3(type_, addr_) (*(const __attribute__((address_space(3))) type_*)(uintptr_t)(addr_))
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
RT_DEV uint4 lds_u4(uint32_t a) {  // 16 bytes at an 8-byte aligned absolute LDS address (ds_read2_b64)
  const u32x4 v = RT_AS3(u32x4_a8, a);
  return make_uint4(v.x, v.y, v.z, v.w);
}

RT_DEV bool hi_is_root(uint4 hi) { return (hi.w & F_BVH_ROOT) != 0u; }

// Full-feature kernels: a program counter is RSZ x the record index = the record's byte offset in the LDS copy of the program,
// where (lo, hi) of a record sit side by side (ONE address per fetch: ds_read_b128 v, pc / ds_read_b128 v, pc offset:16).
constexpr uint32_t RSZ = 32u;
RT_DEV uint4 fetch_hi_global(const DevScene& sc, uint32_t idx) {  // pc-scaled skip pointers, global-memory variants
  uint4 h = sc.hi[idx];
  if ((h.w & 0xffu) == OP_BOX || (h.w & 0xffu) == OP_SEG) h.z *= RSZ;
  if ((h.w & 0xffu) == OP_MEDIUM) h.x *= RSZ;  // end of the boundary's stream
  return h;
}

// Continuing rays first.  The T-list is TWO stacks in one array: rays written by a SCATTER pass (paths that go on) grow from
// position 0 upwards, camera rays of new paths (END pass) from the top downwards; a refill takes continuing rays first and
// new ones only for the lanes that are left.  A path in flight never waits behind paths that have not begun: with ONE stack
// the camera rays of every END pass buried the continuing rays below them until the list ran dry -- at the end of the frame
// (profiles/r04_experiments/r04a_tl_*.txt: the last waves of a launch each finish one ~50-bounce chain alone, 1.6 lanes per
// pass; on book-2 the bottom of the stack held piles of old deep paths that then ran as 64-wide batches one after another).
// Measured (profiles/r04_experiments/r04b_*): book-1 unchanged (the chain that ends a launch began in one of the last blocks),
// book-2 fixed cost 9.6 -> 8.9 ms but slope +3 % (camera rays of one 8x8 block no longer traverse together): off.
#ifndef RT_T_PRIORITY
#define RT_T_PRIORITY 0
#endif
#ifndef RT_SCATTER_TRIES
#define RT_SCATTER_TRIES 4  // in_unit_sphere attempts per SCATTER pass and slot (0 = as many as the unluckiest lane needs)
#endif
#ifndef RT_POOL_MAX_THREADS
#define RT_POOL_MAX_THREADS 1024
#endif
#ifndef RT_POOL_WAVES_PER_EU
#define RT_POOL_WAVES_PER_EU 1
#endif
#ifndef RT_BOX_UNROLL
#define RT_BOX_UNROLL 2
#endif
#ifndef RT_PK_MATH
#define RT_PK_MATH 0  // packed f32 math is an anti-lever on gfx950: 6 v_pk_* per box step cost more than the 12 plain
                      // instructions they replace (C2 8.56 -> 8.11 ms) and tie up 6 more VGPRs for the duplicated operands
#endif
// Two wait lists, two pass types.  A finished ray is classified by what its path does next:
//   E ("end")     the path ends here: a miss, a DiffuseLight hit (the sky dome ends 38 % of book-1's
//                 rays), a path ended by a SCATTER pass, or a slot without a ray.  The END pass books the
//                 sample colour, pulls the next work item and generates its camera ray (event 0).
//   S ("scatter") Lambertian / Metal / Dielectric / Isotropic hit: the SCATTER pass rebuilds the hit
//                 record and runs Material::scatter.
// Each pass type runs 64 lanes wide over slots of ITS class, so a wave no longer issues the union of
// the scatter code and the camera code for every batch of finished rays.  The class of a hit is read
// from the winning SPHERE record (the flattener copies the material kind into its flag word).
// HOT_LDS: the hot slot fields -- the ray (o, d) and (best, best_pc), see pool_lds_bytes -- live in LDS when the
// program leaves room; the cold fields stay in the global SoA region.
template <bool USE_LDS, bool COUNT, bool HOT_LDS, bool WIDE = false>
__global__ __launch_bounds__(RT_POOL_MAX_THREADS, RT_POOL_WAVES_PER_EU) void render_lean_pool(
    DevScene sc, const LaunchConsts* __restrict__ lc, float* __restrict__ out, uint32_t total_work, uint32_t* __restrict__ queue,
    unsigned long long* counters, PoolTuning tune, uint32_t* __restrict__ g_slots) {
  extern __shared__ uint4 s_mem[];
  const uint32_t n_prog = sc.n_prog;
  // Program counters are BYTE offsets: REC r.  Staged: into the LDS image above (a step needs no shift,
  // BOX skip pointers are stored pre-multiplied).  Not staged (program larger than LDS): REC = 16 and
  // (lo, hi) come from global memory.
  constexpr uint32_t REC_BOX = USE_LDS ? LDS_BOX_BYTES : 16u, REC_SPHERE = USE_LDS ? LDS_SPHERE_BYTES : 16u;
  const uint32_t image = USE_LDS ? sc.lds_image_bytes : 0u;
  const uint32_t staged = USE_LDS ? image + 16u * sc.n_mat : 0u;  // bytes
  uint32_t* s_words = reinterpret_cast<uint32_t*>(s_mem);
  const uint32_t pc0 = USE_LDS ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)s_mem : 0u;  // pc of record 0
  if (WIDE) {
    for (uint32_t i = threadIdx.x; i < image / 4u; i += blockDim.x) s_words[i] = sc.lds_off[i];  // (the 4-wide image itself)
  } else if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < n_prog; i += blockDim.x) {
      const uint4 l = sc.lo[i], h = sc.hi[i];
      uint32_t* r = s_words + (sc.lds_off[i] >> 2);
      const uint32_t op = h.w & 0xffu;
      if (op == OP_BOX) {  // lo = (min.x, max.x, min.y, max.y), hi = (min.z, max.z, skip, flags)
        r[0] = pc0 + sc.lds_off[h.z], r[1] = h.w | LDS_BOX_BIT;
        r[2] = l.x, r[3] = l.y, r[4] = l.z, r[5] = l.w, r[6] = h.x, r[7] = h.y;
        r[8] = l.y, r[9] = l.x, r[10] = l.w, r[11] = l.z, r[12] = h.y, r[13] = h.x;
      } else if (op == OP_SPHERE) {  // lo = (offset.xyz, radius), hi = (-, -, material, flags)
        r[0] = h.z, r[1] = h.w, r[2] = l.x, r[3] = l.y, r[4] = l.z, r[5] = l.w;
      } else {  // END + padding
        r[0] = 0u, r[1] = h.w;
        for (uint32_t k = 2; k < LDS_END_BYTES / 4u; k++) r[k] = 0u;
      }
    }
  }
  if (USE_LDS) {
    for (uint32_t i = threadIdx.x; i < sc.n_mat; i += blockDim.x) {
      const uint4 m = sc.mat[2u * i];
      uint32_t* r = s_words + (image >> 2) + 4u * i;
      r[0] = m.x, r[1] = m.y, r[2] = m.z, r[3] = m.w;
    }
  }
  // a SPHERE record as (offset.xyz, radius), its flag word and its material index
#define RT_SPHERE_GEOM(pc_) (USE_LDS ? lds_u4((pc_) + 8u) : sc.lo[(pc_) >> 4])
#define RT_SPHERE_FLAGS(pc_) (USE_LDS ? RT_AS3(uint32_t, (pc_) + 4u) : sc.hi[(pc_) >> 4].w)
#define RT_SPHERE_MAT(pc_) (USE_LDS ? RT_AS3(uint32_t, (pc_)) : sc.hi[(pc_) >> 4].z)
#define RT_FETCH_MATLO(i_) (USE_LDS ? lds_u4(pc0 + image + 16u * (i_)) : sc.mat[2u * (i_)])
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
#if RT_UNIFORM_SLOT_PTR
  uint32_t* slot = uniform_ptr(g_slots + ((size_t)blockIdx.x * n_waves + wave) * (POOL * POOL_FIELDS));
#else
  uint32_t* slot = g_slots + ((size_t)blockIdx.x * n_waves + wave) * (POOL * POOL_FIELDS);
#endif
  // (the cold rows as GLOBAL accesses: through the laundered generic pointer they were flat_load / flat_store, which count
  // on lgkmcnt too -- every wait for an LDS read in a pass then also waited for the rows' trip to L2)
  typedef __attribute__((address_space(1))) uint32_t* g_u32_ptr;
  typedef __attribute__((address_space(1))) float* g_f32_ptr;
  const g_u32_ptr slot_g = (g_u32_ptr)(uintptr_t)slot;
  const g_f32_ptr slotf_g = (g_f32_ptr)(uintptr_t)slot;
  float* slotf = reinterpret_cast<float*>(slot);
  uint16_t* tlist = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(s_mem) + staged) + wave * (3u * POOL);
  uint16_t* slist = tlist + POOL;
  uint16_t* elist = slist + POOL;
  char* hot_base = reinterpret_cast<char*>(s_mem) + (((size_t)staged + n_waves * POOL * 6u + 15u) & ~(size_t)15u);
  float* lds_ray = reinterpret_cast<float*>(hot_base) + (size_t)wave * (POOL * 6u);
  float* lds_best = reinterpret_cast<float*>(hot_base) + (size_t)n_waves * (POOL * 6u) + (size_t)wave * POOL;
  // best_pc in LDS: staged programs fit 16 bits ((pc - pc0) / 8; the three markers keep their low 16 bits), others 32
  typedef typename std::conditional<USE_LDS, uint16_t, uint32_t>::type bpc_t;
  bpc_t* lds_bpc = reinterpret_cast<bpc_t*>(reinterpret_cast<float*>(hot_base) + (size_t)n_waves * (POOL * 7u)) + (size_t)wave * POOL;
#define SLOT_U(f_, j_) slot_g[(f_)*POOL + (j_)]
#define SLOT_F(f_, j_) slotf_g[(f_)*POOL + (j_)]
#define RAY_F(f_, j_) (*(HOT_LDS ? &lds_ray[(f_)*POOL + (j_)] : &slotf[(f_)*POOL + (j_)]))
#ifndef RT_HOT_BEST
#define RT_HOT_BEST 1  // 0: (best, best_pc) stay in the global slot rows even when the rays are in LDS
#endif
  constexpr bool BEST_LDS = HOT_LDS && RT_HOT_BEST;
#define BEST_F(j_) (*(BEST_LDS ? &lds_best[(j_)] : &slotf[PF_BEST * POOL + (j_)]))
  // (the markers NO_HIT / SLOT_NEED_PIXEL / SLOT_ENDED are 0xffffffff / ...fe / ...fd: distinct in 16 bits, above any record)
  auto put_bpc = [&](uint32_t j, uint32_t v) {
    if (!BEST_LDS) slot[PF_BEST_PC * POOL + j] = v;
    else if (USE_LDS) lds_bpc[j] = (bpc_t)(v >= SLOT_ENDED ? v : (v - pc0) >> 3);
    else lds_bpc[j] = (bpc_t)v;
  };
  auto get_bpc = [&](uint32_t j) -> uint32_t {
    if (!BEST_LDS) return slot[PF_BEST_PC * POOL + j];
    const uint32_t v = lds_bpc[j];
    if (!USE_LDS) return v;
    return v >= (SLOT_ENDED & 0xffffu) ? (v | 0xffff0000u) : pc0 + (v << 3);
  };
  // S-list entries (hits on scattering materials: always a record, never a marker) carry in the two top bits of best_pc how
  // many SCATTER passes have already tried to draw their in_unit_sphere (RT_SCATTER_TRIES attempts each, see the pass)
  constexpr uint32_t BPC_TRY_SHIFT = (BEST_LDS && USE_LDS) ? 14u : 30u;  // (16-bit form: image offsets / 8 < 2^14, i.e. images < 128 KB: LDS holds no larger one)
  auto get_bpc_tries = [&](uint32_t j, uint32_t& tries) -> uint32_t {
    uint32_t v = BEST_LDS ? (uint32_t)lds_bpc[j] : slot[PF_BEST_PC * POOL + j];
    tries = v >> BPC_TRY_SHIFT;
    v &= (1u << BPC_TRY_SHIFT) - 1u;
    return (BEST_LDS && USE_LDS) ? pc0 + (v << 3) : v;
  };
  auto bump_bpc_tries = [&](uint32_t j) {
    if (BEST_LDS) lds_bpc[j] = (bpc_t)(lds_bpc[j] + (bpc_t)(1u << BPC_TRY_SHIFT));
    else slot[PF_BEST_PC * POOL + j] += 1u << BPC_TRY_SHIFT;
  };
  // all slots start as "need a work item", all on the E-list
  for (uint32_t j = lane; j < POOL; j += 64u) {
    put_bpc(j, SLOT_NEED_PIXEL);
    elist[j] = (uint16_t)j;
  }
  __syncthreads();  // program staged, pools initialised (the only workgroup barrier)

  const float t_near = load_const(&lc->P.t_near);
  uint32_t t_count = 0, s_count = 0, e_count = POOL, n_dead = 0;  // wave-uniform list sizes / retired slots
  uint32_t tn_count = 0;  // RT_T_PRIORITY: camera rays of new paths, stacked from the top of tlist downwards (t_count: continuing rays, from 0 upwards)
  uint32_t w_next = 0, w_end = 0;                                  // this wave's reserved range of work items
  uint32_t w_chunk = 0, w_delta = 0;                               // ... their chunk, and pixel work index - item index
  bool w_lpt_ready = false;
  bool exhausted = false;                                          // the global counter ran past total_work
  const unsigned long long t_start = RT_TICK();
  unsigned long long t_exhausted = 0;
  RT_TL_DECL(POOL);

  // ---- per-lane traversal state ---------------------------------------------------------------
  uint32_t my_slot = 0;
  bool have_ray = false;  // lane holds a ray (traversing or parked at a SPHERE record)
  V3 o = mk(0.f, 0.f, 0.f), d = o, inv = o;
  uint32_t pc = 0, best_pc = NO_HIT, best_flags = 0;
  float best = F32_MAX;
  // the record at pc as the box step wants it: per axis (near plane, far plane), skip pc, op/flags
  f32x2 cx = {0.f, 0.f}, cy = cx, cz = cx;
  uint32_t c_skip = 0, c_flags = 0xffu;
  uint32_t sgn_x = 0, sgn_y = 0, sgn_z = 0;  // staged: byte offset of this ray's plane pair inside each triple
  uint32_t mstack = 0;                       // WIDE: per level 4 bits "this child's box was hit and the child is still to visit"
  Counts cnt = {0, 0, 0, 0};
  uint32_t total_draws = 0;
  // Per-sample trace (instrumented variant only, rtg_debug_samples on THIS kernel): counters[30] = base of the
  // [sample][pixel work index] x {bounces, draws, Aabb tests, primitive tests} table, counters[31] = per-slot accumulators
  // (draws | aabb | prim rows per wave); 0 = tracing off.  A ray's test counts are the lane's counters at finish minus refill.
  uint32_t* tr_out = nullptr;
  uint32_t* tr_slot = nullptr;
  uint32_t tr_a0 = 0, tr_p0 = 0;
  if (COUNT) {
    tr_out = reinterpret_cast<uint32_t*>(counters[30]);
    if (tr_out) tr_slot = reinterpret_cast<uint32_t*>(counters[31]) + ((size_t)blockIdx.x * n_waves + wave) * (POOL * 3u);
  }
  uint32_t n_box_it = 0, n_box_lanes = 0, n_sph_it = 0, n_sph_lanes = 0, n_shade = 0, n_shade_lanes = 0, n_refill = 0;
  uint32_t n_end = 0, n_end_lanes = 0, n_refill_lanes = 0;
  unsigned long long t_shade = 0, t_serv = 0, t_box = 0, t_sph = 0, t_mark = 0, t_mark2 = 0;  // COUNT: s_memtime shares

  // load the record at pc into (cx, cy, cz, c_skip, c_flags)
#define RT_LOAD_REC() \
        if (USE_LDS) { \
          const u32x2 sf_ = RT_AS3(u32x2, pc); \
          cx = RT_AS3(f32x2, pc + sgn_x); \
          cy = RT_AS3(f32x2, pc + sgn_y); \
          cz = RT_AS3(f32x2, pc + sgn_z); \
          c_skip = sf_.x, c_flags = sf_.y; \
        } else { \
          const uint4 l_ = sc.lo[pc >> 4], h_ = sc.hi[pc >> 4]; \
          cx = inv.x < 0.f ? f32x2{u2f(l_.y), u2f(l_.x)} : f32x2{u2f(l_.x), u2f(l_.y)}; \
          cy = inv.y < 0.f ? f32x2{u2f(l_.w), u2f(l_.z)} : f32x2{u2f(l_.z), u2f(l_.w)}; \
          cz = inv.z < 0.f ? f32x2{u2f(h_.y), u2f(h_.x)} : f32x2{u2f(h_.x), u2f(h_.y)}; \
          c_skip = h_.z * 16u, c_flags = h_.w; \
        }
  // Aabb::hit, aabb.rs:16-27, packed: per axis (t_near_plane, t_far_plane) = ((planes) - o) * inv -- the
  // same two products as the reference's (t0, t1), already in swapped order
  // RT_PK_MATH = 1: v_pk_add_f32 / v_pk_mul_f32 (6 per step); 0: twelve plain f32 instructions (the build then
  // needs -fno-slp-vectorize, or clang re-packs them)
#if RT_PK_MATH
#define RT_BOX_T(t_, c_, o_, i_) const f32x2 t_ = ((c_) - f32x2{o_, o_}) * f32x2{i_, i_}
#else
#define RT_BOX_T(t_, c_, o_, i_) f32x2 t_; t_.x = ((c_).x - (o_)) * (i_); t_.y = ((c_).y - (o_)) * (i_)
#endif
#define RT_IS_BOX() (USE_LDS ? (int32_t)c_flags < 0 : (c_flags & 0xffu) == OP_BOX)
#define RT_BOX_STEP() \
        if (RT_IS_BOX()) { \
          if (COUNT) cnt.aabb++; \
          RT_BOX_T(tx, cx, o.x, inv.x); \
          RT_BOX_T(ty, cy, o.y, inv.y); \
          RT_BOX_T(tz, cz, o.z, inv.z); \
          float start = rs_max(t_near, rs_max(rs_max(tx.x, ty.x), tz.x)); \
          float end = rs_min(best, rs_min(rs_min(tx.y, ty.y), tz.y)); \
          pc = (end > start) ? pc + REC_BOX : c_skip; \
          RT_LOAD_REC(); \
        }
  // WIDE: continue at node `n_pc`: its next pending child (left to right), else up to its parent, else the program ends
  auto wide_next = [&](uint32_t n_pc) {
    for (;;) {
      const uint4 h = lds_u4(n_pc);  // parent, flags, child offsets
      const uint32_t lvl4 = ((h.y >> 8) & 0xfu) * 4u;
      const uint32_t m = (mstack >> lvl4) & 0xfu;
      if (m != 0u) {
        const uint32_t c = (uint32_t)__builtin_ctz(m);
        mstack &= ~(1u << (lvl4 + c));
        const uint32_t pair = c < 2u ? h.z : h.w;
        pc = pc0 + (((c & 1u) ? pair >> 16 : pair & 0xffffu) << 3);
        c_flags = ((h.y >> (16u + c)) & 1u) ? RT_AS3(uint32_t, pc + 4u) : (LDS_BOX_BIT | (uint32_t)OP_BOX);
        return;
      }
      if (h.x == WIDE_NO_PARENT) {
        c_flags = OP_END;
        return;
      }
      n_pc = pc0 + h.x;
    }
  };
  // WIDE: one step at a node: Aabb::hit (aabb.rs:16-27, the arithmetic of rt_full_traverse.inc) for its <= 4 boxes at once
#define RT_WIDE_STEP() \
        if (RT_IS_BOX()) { \
          const uint4 h_ = lds_u4(pc); \
          const uint32_t nch_ = (h_.y >> 12) & 7u; \
          uint32_t m_ = 0; \
          _Pragma("unroll") for (uint32_t k_ = 0; k_ < 4u; k_++) { \
            const f32x2 px_ = RT_AS3(f32x2, pc + 8u + 48u * k_ + sgn_x); \
            const f32x2 py_ = RT_AS3(f32x2, pc + 8u + 48u * k_ + sgn_y); \
            const f32x2 pz_ = RT_AS3(f32x2, pc + 8u + 48u * k_ + sgn_z); \
            const float ax = (px_.x - o.x) * inv.x, bx = (px_.y - o.x) * inv.x; \
            const float ay = (py_.x - o.y) * inv.y, by = (py_.y - o.y) * inv.y; \
            const float az = (pz_.x - o.z) * inv.z, bz = (pz_.y - o.z) * inv.z; \
            const float start = rs_max(t_near, rs_max(rs_max(ax, ay), az)); \
            const float end = rs_min(best, rs_min(rs_min(bx, by), bz)); \
            m_ |= (end > start ? 1u : 0u) << k_; \
          } \
          m_ &= (1u << nch_) - 1u; \
          if (COUNT) cnt.aabb += nch_; \
          const uint32_t lvl4_ = ((h_.y >> 8) & 0xfu) * 4u; \
          mstack = (mstack & ~(0xfu << lvl4_)) | (m_ << lvl4_); \
          wide_next(pc); \
        }
  for (;;) {
    uint32_t op = have_ray ? (c_flags & 0xffu) : 0xffu;
    const uint64_t m_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    const uint64_t m_sph = __builtin_amdgcn_ballot_w64(op == OP_SPHERE);
    const uint32_t n_busy = (uint32_t)__builtin_popcountll(m_box | m_sph);
    // ============================== SERVICE ======================================================
    // due when enough lanes are idle AND it can do something for them: rays to hand out, or a full pass once the lanes at
    // END are pushed (rays wait in S and E for company: "idle lanes" alone would call it after every step of a starved wave)
#if RT_GATED_SERVICE
    const uint32_t n_fin = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(op == OP_END));
    const bool can_serve = (t_count + tn_count) != 0u || s_count + n_fin >= 64u || e_count + n_fin >= 64u;
#else
    const bool can_serve = true;
#endif
    if ((64u - n_busy >= tune.refill_min && can_serve) || n_busy == 0) {
      if (COUNT) t_mark = RT_TICK();
      // (1) finish: rays that reached END hand (best, best_pc) to their slot and join the E- or S-list
      {
        const bool fin = have_ray && op == OP_END;
        const bool to_e = fin && (best_pc == NO_HIT || ((best_flags >> F_MATKIND_SHIFT) & 7u) == MAT_DIFFUSE_LIGHT);
        const bool to_s = fin && !to_e;
        const uint64_t m_e = __builtin_amdgcn_ballot_w64(to_e), m_s = __builtin_amdgcn_ballot_w64(to_s);
        if (fin) {
          BEST_F(my_slot) = best;
          put_bpc(my_slot, best_pc);
          if (COUNT && tr_slot) tr_slot[POOL + my_slot] += cnt.aabb - tr_a0, tr_slot[2u * POOL + my_slot] += cnt.prim - tr_p0;
          if (to_e) elist[e_count + lane_rank(m_e)] = (uint16_t)my_slot;
          else slist[s_count + lane_rank(m_s)] = (uint16_t)my_slot;
          have_ray = false;
          c_flags = 0xffu;  // no record: the box loop tests c_flags alone
        }
        e_count += (uint32_t)__builtin_popcountll(m_e);
        s_count += (uint32_t)__builtin_popcountll(m_s);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const bool starving = (t_count + tn_count == 0 && n_busy == 0);  // nothing to traverse: run partial passes too
      // (2a) SCATTER pass: Material::scatter for 64 hits on scattering materials
      while (s_count >= 64u || (s_count > 0 && starving)) {
        const uint32_t take = s_count < 64u ? s_count : 64u;
        s_count -= take;
        if (COUNT) n_shade++, n_shade_lanes += take, t_mark2 = RT_TICK();
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        bool live = false, ended = false, lpt_on = false, deferred = false;
        uint32_t j = 0, lpt_blk = 0;
        if (lane < take) {
          j = slist[s_count + lane];
          uint32_t tries;
          const uint32_t bpc = get_bpc_tries(j, tries);
          V3 so = mk(RAY_F(PF_O, j), RAY_F(PF_O + 1, j), RAY_F(PF_O + 2, j));
          V3 sd = mk(RAY_F(PF_D, j), RAY_F(PF_D + 1, j), RAY_F(PF_D + 2, j));
          uint32_t bounces = SLOT_U(PF_BOUNCES, j);
          // the strength row is only valid from the first scatter on: a fresh path carries (1, 1, 1) (lib.rs:63) implicitly
          V3 strength = mk(SLOT_F(PF_STRENGTH, j), SLOT_F(PF_STRENGTH + 1, j), SLOT_F(PF_STRENGTH + 2, j));
          if (bounces == 0u) strength = splat(1.f);
          RT_TL_BOUNCES(bounces);
          const uint32_t s = SLOT_U(PF_SAMPLE, j), xy = SLOT_U(PF_XY, j);
          const float hb = BEST_F(j);
          // ---------------- color() loop body, lib.rs:73-97, for a hit on a scattering material ----------
          // The hit record first: it needs LDS data only (ray, best, the winning record, its material), so this arithmetic runs
          // while the cold rows above (bounces, strength, sample, pixel: one trip to L2) are still on their way; the RNG, which is
          // keyed by them, comes after.
          const uint4 plo = RT_SPHERE_GEOM(bpc);
          const uint32_t pflags = RT_SPHERE_FLAGS(bpc);
          V3 off = mk(u2f(plo.x), u2f(plo.y), u2f(plo.z));
          V3 lo_o = so;
          if (pflags & F_TRANSLATE) lo_o = vsub(so, off);      // object.rs:275-278
          V3 hp = vadd(lo_o, smul(hb, sd));                    // ray.rs:15
          V3 hn = sdiv(hp, u2f(plo.w));                        // object.rs:104
          if (pflags & F_TRANSLATE) hp = vadd(hp, off);        // object.rs:279-282
          if (pflags & F_FLIP) hn = vneg(hn);                  // object.rs:249-252
          const uint4 mlo = RT_FETCH_MATLO(RT_SPHERE_MAT(bpc));
          const uint32_t kind = (pflags >> F_MATKIND_SHIFT) & 7u;  // the flattener's copy of the material kind
          const float param = u2f(mlo.w);
          const V3 mcol = mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z));
          // |d| and unit(d) once for the Metal and the Dielectric lanes of the pass (vec3.rs:59,66): the two
          // branches below are exclusive, the wave usually runs both, and a sqrt + three divides is what they share
          float sd_len = 0.f;
          V3 sd_unit = sd;
          if (kind == MAT_METAL || kind == MAT_DIELECTRIC) sd_len = vlen(sd), sd_unit = sdiv(sd, sd_len);
          __builtin_amdgcn_sched_barrier(0);  // (keeps the block above in front of the first use of the cold rows)
          SampleRng rng;
          rng.init(seed, (P.ny - 1u - (xy >> 16)) * P.nx + (xy & 0xffffu), s);
          rng.set_event(bounces + 1u);
          rng.seek(3u * RT_SCATTER_TRIES * tries);  // the attempts earlier passes made (whole Philox blocks: no block is generated here)
          // lib.rs:76 with emitted = 0 (material.rs:126): accum stays +0, see PoolField
          V3 nd = mk(0.f, 0.f, 0.f), att = mcol;
          bool scattered = true;
          // Lambertian, Metal and Isotropic each draw exactly one in_unit_sphere before any other draw of
          // this event (reflect() consumes no randomness): ONE rejection loop serves all three.
          // The rejection loop runs in lock-step: the wave pays for the UNLUCKIEST of its 64 lanes (6.9 attempts expected,
          // against 1.9 for one lane; an attempt is three draws, a Philox block four).  So a pass makes at most
          // RT_SCATTER_TRIES = 4 attempts (12 draws = 3 whole blocks: every lane crosses the block boundaries at the same
          // draws); the 5 % of the lanes that are still without a direction leave their slot as it is, note the try in
          // best_pc and go back on the S-list -- a later pass continues their stream where this one stopped (same draws, same
          // order: the stream is counter-based).  The fourth pass of a slot loops to the end.
          V3 rs = mk(0.f, 0.f, 0.f);
          if (kind != MAT_DIELECTRIC) deferred = !in_unit_sphere_tries(rng, (RT_SCATTER_TRIES && tries < 3u) ? (uint32_t)RT_SCATTER_TRIES : 0xffffffffu, rs);
          if (COUNT && !deferred) cnt.shaded++;
          if (kind == MAT_LAMBERTIAN) {  // material.rs:57-65
            V3 target = vadd(vadd(hp, hn), rs);
            nd = vsub(target, hp);
          } else if (kind == MAT_METAL) {  // material.rs:66-80
            V3 refl = reflect(sd_unit, hn);
            nd = vadd(refl, smul(param, rs));
            scattered = vdot(nd, hn) > 0.f;
          } else if (kind == MAT_DIELECTRIC) {  // material.rs:81-107
            V3 outward;
            float ni_over_nt, cosine;
            float dn = vdot(sd, hn);
            if (dn > 0.f) {
              outward = vneg(hn);
              ni_over_nt = param;
              cosine = param * dn / sd_len;
            } else {
              outward = hn;
              ni_over_nt = 1.0f / param;
              cosine = -dn / sd_len;
            }
            V3 uv = sd_unit;  // refract, vec3.rs:321-330
            float dt = vdot(uv, outward);
            float disc = 1.0f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
            bool refracted = disc > 0.f;
            if (refracted) {
              nd = vsub(smul(ni_over_nt, vsub(uv, smul(dt, outward))), smul(__builtin_sqrtf(disc), outward));
              refracted = rng.gen_f32() >= schlick<true>(cosine, param);  // material.rs:97: draw only if Some
            }
            if (!refracted) nd = reflect(sd, hn);
            att = splat(1.f);
          } else {  // Isotropic, material.rs:109-116
            nd = rs;
          }
          if (COUNT) total_draws += rng.draws;
          if (COUNT && tr_slot) tr_slot[j] += rng.draws;
          lpt_on = !deferred && s < cm.lpt_samples && bounces == cm.lpt_deep;  // phase 1 of the cost-ordered queue: a deep scatter event
          if (lpt_on) lpt_blk = pixel_to_work(P, load_const(&lc->pm), xy & 0xffffu, xy >> 16) >> 8;
          if (deferred) {
            bump_bpc_tries(j);
          } else if (scattered) {
            strength = vmul(strength, att);  // lib.rs:87
            if (bounces != P.max_bounces) {  // lib.rs:93-97
              bounces += 1;
              live = true;
            }
          }
          if (deferred) {
          } else if (live) {
            RAY_F(PF_O, j) = hp.x, RAY_F(PF_O + 1, j) = hp.y, RAY_F(PF_O + 2, j) = hp.z;
            RAY_F(PF_D, j) = nd.x, RAY_F(PF_D + 1, j) = nd.y, RAY_F(PF_D + 2, j) = nd.z;
            SLOT_F(PF_STRENGTH, j) = strength.x, SLOT_F(PF_STRENGTH + 1, j) = strength.y, SLOT_F(PF_STRENGTH + 2, j) = strength.z;
            SLOT_U(PF_BOUNCES, j) = bounces;
            if (COUNT) cnt.rays++;
          } else {  // both early returns of color() yield accum (lib.rs:90,94): the END pass books it
            ended = true;
            put_bpc(j, SLOT_ENDED);
          }
        }
        if (cm.lpt_samples) lpt_count(cm, lpt_on, lpt_blk);
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live), m_ended = __builtin_amdgcn_ballot_w64(ended);
        const uint64_t m_def = __builtin_amdgcn_ballot_w64(deferred);
        RT_TL_SHADE(take, m_live);
        if (live) tlist[t_count + lane_rank(m_live)] = (uint16_t)j;
        if (ended) elist[e_count + lane_rank(m_ended)] = (uint16_t)j;
        if (deferred) slist[s_count + lane_rank(m_def)] = (uint16_t)j;  // (over entries this pass has already read)
        t_count += (uint32_t)__builtin_popcountll(m_live);
        e_count += (uint32_t)__builtin_popcountll(m_ended);
        s_count += (uint32_t)__builtin_popcountll(m_def);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_shade += RT_TICK() - t_mark2;
      }
      // (2b) END pass: book the sample colour, next work item, camera ray
      while (e_count >= 64u || (e_count > 0 && starving && t_count + tn_count == 0)) {
        const uint32_t take = e_count < 64u ? e_count : 64u;
        e_count -= take;
        if (COUNT) n_end++, n_end_lanes += take, t_mark2 = RT_TICK();
        const DevParams P = load_const(&lc->P);
        const ChunkMode cm = load_const(&lc->cm);
        const uint64_t seed = ((uint64_t)P.seed_hi << 32) | P.seed_lo;
        uint32_t st = ST_DEAD, j = 0, s = 0, x = 0, row = 0;
        V3 col = mk(0.f, 0.f, 0.f);
        if (lane < take) {
          j = elist[e_count + lane];
          const uint32_t bpc = get_bpc(j);
          if (bpc == SLOT_NEED_PIXEL) {
            st = ST_NEED_PIXEL;
          } else {
            s = SLOT_U(PF_SAMPLE, j);
            const uint32_t xy = SLOT_U(PF_XY, j);
            x = xy & 0xffffu, row = xy >> 16;
            V3 result = mk(0.f, 0.f, 0.f);  // lib.rs:100: a miss is black, accum is discarded
            if (bpc != NO_HIT) {
              const V3 accum = mk(0.f, 0.f, 0.f);
              result = accum;  // SLOT_ENDED: color() already returned accum
              if (bpc != SLOT_ENDED) {  // DiffuseLight hit: lib.rs:76 then scatter() == None (material.rs:108)
                if (COUNT) cnt.shaded++;
                V3 strength = mk(SLOT_F(PF_STRENGTH, j), SLOT_F(PF_STRENGTH + 1, j), SLOT_F(PF_STRENGTH + 2, j));
                if (SLOT_U(PF_BOUNCES, j) == 0u) strength = splat(1.f);  // the camera ray hit the light: lib.rs:63
                const uint4 mlo = RT_FETCH_MATLO(RT_SPHERE_MAT(bpc));
                const V3 emitted = smul(u2f(mlo.w), mk(u2f(mlo.x), u2f(mlo.y), u2f(mlo.z)));  // material.rs:120-128
                result = vadd(accum, vmul(strength, emitted));
              }
            }
            if (cm.scratch) {  // chunk mode: park the sample colour, folded in order afterwards
              float* sp = cm.scratch + 3ull * ((size_t)s * cm.pix_work + pixel_to_work(P, load_const(&lc->pm), x, row));
              RT_SCRATCH_STORE(sp, result);
              if (COUNT && tr_out) {
                uint32_t* tp = tr_out + 4ull * ((size_t)s * cm.pix_work + pixel_to_work(P, load_const(&lc->pm), x, row));
                tp[0] = SLOT_U(PF_BOUNCES, j), tp[1] = tr_slot[j], tp[2] = tr_slot[POOL + j], tp[3] = tr_slot[2u * POOL + j];
              }
            } else {
              col = vadd(mk(SLOT_F(PF_COL, j), SLOT_F(PF_COL + 1, j), SLOT_F(PF_COL + 2, j)), result);  // vec3.rs:195-203
            }
            s++;
            if (s == P.ns || (cm.scratch && s % cm.chunk == 0u)) {
              if (!cm.scratch) {
                V3 px = sdiv(col, (float)P.ns);  // lib.rs:374
                float* op_ = out + 3ull * ((size_t)row * P.nx + x);
                op_[0] = px.x, op_[1] = px.y, op_[2] = px.z;
              }
              st = ST_NEED_PIXEL;
            } else {
              st = ST_GEN;
            }
          }
        }
        // next work item.  The wave reserves WORK_BLOCK items at a time from the global counter (one
        // returning atomic per ~WORK_BLOCK samples: a single counter word saturates near 88 dequeues/us
        // on this chip) and hands them out to its lanes locally.
        for (;;) {
          const uint64_t need = __builtin_amdgcn_ballot_w64(st == ST_NEED_PIXEL);
          if (need == 0) break;
          if (w_next == w_end && !exhausted) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(queue, cm.work_block);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total_work) {
              exhausted = true;
              if (COUNT) t_exhausted = RT_TICK();
              RT_TL_EXHAUSTED(POOL - n_dead, t_count + tn_count, s_count, e_count);
            } else {
              if (cm.scratch) w_chunk = lpt_reservation(cm, base, lane, w_delta, w_lpt_ready);
              w_next = base;
              w_end = (total_work - base < cm.work_block) ? total_work : base + cm.work_block;
            }
          }
          const uint32_t avail = w_end - w_next;
          if (st == ST_NEED_PIXEL) {
            const uint32_t r = lane_rank(need);
            if (r < avail) {
              uint32_t w = w_next + r, first = 0;
              if (cm.scratch) {  // work item = (chunk, pixel), pixel-minor so a wave's grab stays coherent
                w += w_delta;
                first = cm.s_begin + w_chunk * cm.chunk;
              }
              if (work_to_pixel(P, load_const(&lc->pm), w, x, row) && first < P.ns) {
                s = first;
                col = mk(0.f, 0.f, 0.f);
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
        if (live) {  // par_cast closure, lib.rs:366-371 (event 0) and color()'s initial state, lib.rs:62-67
          const uint32_t y = P.ny - 1u - row;
          SampleRng rng;
          rng.init(seed, y * P.nx + x, s);
          float u = ((float)x + rng.gen_f32()) / (float)P.nx;
          float v = ((float)y + rng.gen_f32()) / (float)P.ny;
          V3 so, sd;
          float time;
          const DevCamera cam = load_const(&lc->cam);
          get_ray(cam, u, v, rng, so, sd, time);
          if (COUNT) total_draws += rng.draws, cnt.rays++;
          if (COUNT && tr_slot) tr_slot[j] = rng.draws, tr_slot[POOL + j] = 0u, tr_slot[2u * POOL + j] = 0u;
          RAY_F(PF_O, j) = so.x, RAY_F(PF_O + 1, j) = so.y, RAY_F(PF_O + 2, j) = so.z;
          RAY_F(PF_D, j) = sd.x, RAY_F(PF_D + 1, j) = sd.y, RAY_F(PF_D + 2, j) = sd.z;
          if (!cm.scratch) SLOT_F(PF_COL, j) = col.x, SLOT_F(PF_COL + 1, j) = col.y, SLOT_F(PF_COL + 2, j) = col.z;
          SLOT_U(PF_BOUNCES, j) = 0u, SLOT_U(PF_SAMPLE, j) = s;
          SLOT_U(PF_XY, j) = x | (row << 16);
        }
        const uint64_t m_live = __builtin_amdgcn_ballot_w64(live);
#if RT_T_PRIORITY
        if (live) tlist[POOL - 1u - (tn_count + lane_rank(m_live))] = (uint16_t)j;
        tn_count += (uint32_t)__builtin_popcountll(m_live);
#else
        if (live) tlist[t_count + lane_rank(m_live)] = (uint16_t)j;
        t_count += (uint32_t)__builtin_popcountll(m_live);
#endif
        n_dead += take - (uint32_t)__builtin_popcountll(m_live);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (COUNT) t_shade += RT_TICK() - t_mark2;
        RT_TL_ALIVE(POOL - n_dead);
      }
      // (3) refill idle lanes from the T-list
      {
        const uint64_t m_idle = __builtin_amdgcn_ballot_w64(!have_ray);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(m_idle);
        const uint32_t t_all = t_count + tn_count;
        const uint32_t got = n_idle < t_all ? n_idle : t_all;
        if (got) {
          const uint32_t r = lane_rank(m_idle);
          const uint32_t take_c = got < t_count ? got : t_count;  // continuing rays first
          if (!have_ray && r < got) {
            my_slot = r < take_c ? tlist[t_count - 1u - r] : tlist[POOL - tn_count + (r - take_c)];
            o = mk(RAY_F(PF_O, my_slot), RAY_F(PF_O + 1, my_slot), RAY_F(PF_O + 2, my_slot));
            d = mk(RAY_F(PF_D, my_slot), RAY_F(PF_D + 1, my_slot), RAY_F(PF_D + 2, my_slot));
            inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);  // aabb.rs:17
            sgn_x = inv.x < 0.f ? 32u : 8u, sgn_y = inv.y < 0.f ? 40u : 16u, sgn_z = inv.z < 0.f ? 48u : 24u;  // aabb.rs:20-23
            pc = pc0, best = F32_MAX, best_pc = NO_HIT, best_flags = 0;
            if (COUNT) tr_a0 = cnt.aabb, tr_p0 = cnt.prim;
            if (WIDE) {
              c_flags = LDS_BOX_BIT | (uint32_t)OP_BOX, mstack = 0u;  // the root node (offset 0 of the image)
            } else {
              RT_LOAD_REC();
            }
            have_ray = true;
          }
          t_count -= take_c, tn_count -= got - take_c;
          if (COUNT) n_refill++, n_refill_lanes += got;
          RT_TL_REFILL(got);
        }
      }
      if (COUNT) t_serv += RT_TICK() - t_mark;
      RT_TL_SERVICE();
      if (n_dead == POOL) break;  // every slot retired: this wave is done
      if (__builtin_amdgcn_ballot_w64(have_ray) == 0) continue;  // nothing to traverse yet: service again
      op = have_ray ? (c_flags & 0xffu) : 0xffu;
    }
    // ============================== TRAVERSE ======================================================
    // box runs and sphere passes alternate in this inner loop until a service is due (the same test as at
    // the top): the big SERVICE block stays out of the cycle the wave spends its time in
    // (at least one box run or sphere pass per visit: after a service that found nothing to refill the test
    // still says "service" although there is nothing to service)
    uint64_t b_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    uint64_t b_sph = __builtin_amdgcn_ballot_w64(op == OP_SPHERE);
    for (;;) {
    if (b_box != 0 && (uint32_t)__builtin_popcountll(b_sph) < tune.sphere_min) {
      // box run: tight loop, schedule re-evaluated once `box_leave` lanes have left the BOX state
      if (COUNT) t_mark = RT_TICK();
      const uint32_t n0 = (uint32_t)__builtin_popcountll(b_box);
      const uint32_t floor_lanes = n0 > tune.box_leave ? n0 - tune.box_leave : 0u;
      uint32_t n_now;
      do {
        if (COUNT) n_box_it++;
        if (WIDE) {
          RT_WIDE_STEP();
        } else {
        RT_BOX_STEP();
#if RT_BOX_UNROLL >= 2
        RT_BOX_STEP();  // lanes that left the BOX state sit this one out; the schedule check runs every other step
#endif
#if RT_BOX_UNROLL >= 3
        RT_BOX_STEP();
#endif
#if RT_BOX_UNROLL >= 4
        RT_BOX_STEP();
#endif
        }
        n_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(RT_IS_BOX()));
        if (COUNT) n_box_lanes += n_now;
      } while (n_now > floor_lanes);
      if (COUNT) t_box += RT_TICK() - t_mark;
    } else if (b_sph != 0) {
      if (COUNT) n_sph_it++, n_sph_lanes += (uint32_t)__builtin_popcountll(b_sph), t_mark = RT_TICK();
      if (op == OP_SPHERE) {  // Sphere::hit, object.rs:84-111 (+ Translate :275)
        if (COUNT) cnt.prim++;
        const uint4 slo = RT_SPHERE_GEOM(pc);  // (offset.xyz, radius)
        V3 lo_o = o;
        if (c_flags & F_TRANSLATE) lo_o = vsub(o, mk(u2f(slo.x), u2f(slo.y), u2f(slo.z)));
        float t;
        if (sphere_hit_t(lo_o, d, u2f(slo.w), t_near, best, t)) {
          best = t;
          best_pc = pc;
          best_flags = c_flags;
        }
        if (WIDE) {
          wide_next(pc0 + RT_AS3(uint32_t, pc + 24u));  // back to the parent node
        } else {
          pc += REC_SPHERE;
          RT_LOAD_REC();
        }
      }
      if (COUNT) t_sph += RT_TICK() - t_mark;
    }
    op = have_ray ? (c_flags & 0xffu) : 0xffu;
    b_box = __builtin_amdgcn_ballot_w64(op == OP_BOX);
    b_sph = __builtin_amdgcn_ballot_w64(op == OP_SPHERE);
    const uint32_t busy = (uint32_t)__builtin_popcountll(b_box | b_sph);
    if (64u - busy >= tune.refill_min || busy == 0) break;
    }
  }
  RT_TL_DONE();
  if (COUNT) {
    atomicAdd(&counters[0], (unsigned long long)cnt.aabb);
    atomicAdd(&counters[1], (unsigned long long)cnt.prim);
    atomicAdd(&counters[2], (unsigned long long)cnt.shaded);
    atomicAdd(&counters[3], (unsigned long long)cnt.rays);
    atomicAdd(&counters[4], (unsigned long long)total_draws);
    if (lane == 0) {
      unsigned long long* sched = counters + 8;
      atomicAdd(&sched[0], (unsigned long long)n_box_it), atomicAdd(&sched[1], (unsigned long long)n_box_lanes);
      atomicAdd(&sched[2], (unsigned long long)n_sph_it), atomicAdd(&sched[3], (unsigned long long)n_sph_lanes);
      atomicAdd(&sched[4], (unsigned long long)n_shade), atomicAdd(&sched[5], (unsigned long long)n_shade_lanes);
      atomicAdd(&sched[6], (unsigned long long)n_refill), atomicAdd(&sched[7], (unsigned long long)n_refill_lanes);
      atomicAdd(&counters[16], t_shade), atomicAdd(&counters[17], t_serv - t_shade), atomicAdd(&counters[18], t_box),
          atomicAdd(&counters[19], t_sph);
      atomicAdd(&counters[20], (unsigned long long)n_end), atomicAdd(&counters[21], (unsigned long long)n_end_lanes);
      // wave timeline (s_memtime is not synchronised across XCDs: only per-wave differences are meaningful)
      const unsigned long long dur = RT_TICK() - t_start, exh = t_exhausted - t_start;
      atomicMax(&counters[24], dur), atomicAdd(&counters[25], dur), atomicAdd(&counters[26], 1ull);
      atomicMax(&counters[27], (1ull << 62) - exh), atomicAdd(&counters[28], exh), atomicMax(&counters[29], exh);
    }
  }
#undef RT_BOX_STEP
#undef RT_BOX_T
#undef RT_LOAD_REC
#undef RT_IS_BOX
#undef RT_SPHERE_GEOM
#undef RT_SPHERE_FLAGS
#undef RT_SPHERE_MAT
#undef RT_FETCH_MATLO
#undef BEST_F
#undef SLOT_U
#undef SLOT_F
#undef RAY_F
}

}  // namespace rtg
