/*
 * rtiow_gpu.h -- C ABI of the MI355X-native path-tracing hot path (librtiow_gpu.so).
 *
 * Drop-in boundary for cbiffle/rtiow-rust's `par_cast(nx, ny, ns, &camera, world)` seam
 * (reference src/lib.rs:363) and everything below it: World::hit_top (lib.rs:23-55), color()
 * (lib.rs:60-101), Bvh/Aabb traversal (bvh.rs:84-120, aabb.rs:16-27), Object::hit for
 * Sphere/Rect/FlipNormals/Translate/Scale/RotateY/And/LinearMove/ConstantMedium (object.rs),
 * Material::{scatter,emitted} (material.rs:55-128), Texture (texture.rs), Perlin (perlin.rs),
 * Camera::get_ray (camera.rs:52-63).
 *
 * The reference has no FFI (`#![forbid(unsafe_code)]`, lib.rs:1).  A Rust `-sys` binding would walk
 * its own object graph and mirror each constructor through the builder calls below (one call per
 * reference constructor -- see INTEGRATION.md), then call rtg_par_cast where it called par_cast.
 *
 * Conventions: plain pointers and sizes only; return 0 = ok, negative = error (never throws or
 * aborts across the boundary); rtg_last_error() gives the message for the calling thread.  Handles
 * (rtg_id) are indices local to one builder.  A scene is bound to one device and may be used from
 * one thread at a time; different scenes are independent.
 */
#ifndef RTIOW_GPU_H
#define RTIOW_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTG_OK 0
#define RTG_ERR_INVALID (-1)     /* bad argument / handle (reference: type error at compile time)   */
#define RTG_ERR_EMPTY_BVH (-2)   /* bvh.rs:60 panic "Can't create a BVH from zero objects."          */
#define RTG_ERR_NAN (-3)         /* bvh.rs:45,56 partial_cmp().unwrap() panic on NaN extents          */
#define RTG_ERR_RANGE (-4)       /* camera.rs:55 gen_range(lo,hi) asserts lo < hi                     */
#define RTG_ERR_UNSUPPORTED (-5) /* nesting beyond the general walk's stacks (32 wrappers, 3 media levels) */
#define RTG_ERR_DEVICE (-6)      /* HIP runtime error / no GPU                                         */

typedef uint32_t rtg_id;
#define RTG_INVALID_ID 0xffffffffu

typedef struct rtg_builder rtg_builder; /* a scene under construction (host only)              */
typedef struct rtg_scene rtg_scene;     /* a flattened scene resident in one GPU's HBM         */

/* camera.rs:6-15 `struct Camera` -- 21 floats, plain data. */
typedef struct rtg_camera {
  float origin[3];
  float lower_left_corner[3];
  float horizontal[3];
  float vertical[3];
  float u[3];
  float v[3];
  float lens_radius;
  float exposure_start, exposure_end;
} rtg_camera;

/* Arguments of par_cast (lib.rs:363) plus the constants the reference bakes in. */
typedef struct rtg_params {
  uint32_t struct_size; /* = sizeof(rtg_params)                                               */
  uint32_t nx, ny, ns;  /* lib.rs:363                                                         */
  uint32_t max_bounces; /* literal 50 at lib.rs:93                                            */
  float t_near;         /* NEAR = 0.001 at lib.rs:35,53                                       */
  uint64_t seed;        /* key of the per-(pixel,sample) counter RNG (DESIGN.md determinism)  */
  /* Pixel sharding for multi-GPU (one process per GPU): the image is cut into tile_w x tile_h
   * tiles numbered row-major from the top-left; this call renders tiles with
   * tile_index % nranks == rank and leaves every other pixel of `out` untouched. */
  uint32_t tile_w, tile_h; /* multiples of 8; 0 -> 16                                         */
  uint32_t rank, nranks;   /* nranks 0 -> 1                                                   */
  uint32_t flags;          /* RTG_FLAG_*                                                      */
  uint32_t reserved;
} rtg_params;

#define RTG_FLAG_COUNTERS 1u /* fill rtg_stats counters (instrumented kernel variant, slower) */
#define RTG_FLAG_TRACE_KERNEL 2u /* rtg_debug_samples: trace the production ray-pool kernel instead of the one-lane probe */

typedef struct rtg_stats {
  uint32_t struct_size; /* = sizeof(rtg_stats)                                                */
  float kernel_ms;      /* HIP-event time of the render kernel on its stream                  */
  uint64_t samples;     /* pixels rendered by this call x ns                                   */
  uint64_t aabb_tests;  /* Aabb::hit calls        (aabb.rs:16)                                 */
  uint64_t prim_tests;  /* Sphere/Rect::hit calls (object.rs:84,185)                           */
  uint64_t shaded_hits; /* hit_top() == Some      (lib.rs:73)                                  */
  uint64_t rays;        /* hit_top() calls                                                     */
  uint64_t draws;       /* RNG u32 draws                                                       */
} rtg_stats;

/* ---- library ---------------------------------------------------------------------------- */
const char* rtg_version(void);
const char* rtg_last_error(void);
int rtg_device_count(int* n);

/* ---- builder: one call per reference constructor ----------------------------------------- */
int rtg_builder_create(rtg_builder** out);
void rtg_builder_destroy(rtg_builder* b);

/* texture.rs:8 constant, :12 checker, :23 perlin.  Return RTG_INVALID_ID on error. */
rtg_id rtg_texture_constant(rtg_builder* b, const float rgb[3]);
rtg_id rtg_texture_checker(rtg_builder* b, rtg_id t0, rtg_id t1);
rtg_id rtg_texture_perlin(rtg_builder* b, float scale);
/* perlin.rs:24-29 VECS / PERM_X / PERM_Y / PERM_Z (thread_rng-seeded globals in the reference):
 * 256 xyz gradient vectors and three 256-entry permutations, supplied by the caller. */
int rtg_builder_set_perlin_tables(rtg_builder* b, const float vecs[768], const uint8_t perm_x[256],
                                  const uint8_t perm_y[256], const uint8_t perm_z[256]);

/* material.rs:10-39 enum Material */
rtg_id rtg_material_lambertian(rtg_builder* b, rtg_id albedo_texture);
rtg_id rtg_material_metal(rtg_builder* b, const float albedo[3], float fuzz);
rtg_id rtg_material_dielectric(rtg_builder* b, float ref_idx);
rtg_id rtg_material_diffuse_light(rtg_builder* b, rtg_id emission_texture, float brightness);
rtg_id rtg_material_isotropic(rtg_builder* b, rtg_id albedo_texture);

/* object.rs: Sphere :75, Rect<A> :131 (axis 0/1/2 = StaticX/Y/Z), FlipNormals :239, Translate :262,
 * Scale :296, rotate_y :477, And :394, rect_prism :420, LinearMove :489, ConstantMedium :533;
 * bvh.rs:128 from_scene (a Bvh is itself an Object, bvh.rs:84). */
rtg_id rtg_object_sphere(rtg_builder* b, float radius, rtg_id material);
rtg_id rtg_object_rect(rtg_builder* b, int orthogonal_to, float range0_start, float range0_end,
                       float range1_start, float range1_end, float k, rtg_id material);
rtg_id rtg_object_flip_normals(rtg_builder* b, rtg_id object);
rtg_id rtg_object_translate(rtg_builder* b, const float offset[3], rtg_id object);
rtg_id rtg_object_scale(rtg_builder* b, const float factor[3], rtg_id object);
rtg_id rtg_object_rotate_y(rtg_builder* b, float degrees, rtg_id object);
rtg_id rtg_object_and(rtg_builder* b, rtg_id object0, rtg_id object1);
rtg_id rtg_object_rect_prism(rtg_builder* b, const float p0[3], const float p1[3], rtg_id material);
rtg_id rtg_object_linear_move(rtg_builder* b, rtg_id object, const float motion[3]);
rtg_id rtg_object_constant_medium(rtg_builder* b, rtg_id boundary, float density, rtg_id material);
rtg_id rtg_object_bvh(rtg_builder* b, const rtg_id* objects, size_t n, float exposure_start,
                      float exposure_end);

/* NOT in the reference (non-parity option): the same Bvh object built with a surface-area
 * heuristic instead of Bvh::new's widest-axis median split.  Closest-hit results are tree-invariant except
 * at exact-t ties; fewer Aabb tests per ray (book-1: 26.0 instead of 40.8). */
rtg_id rtg_object_bvh_sah(rtg_builder* b, const rtg_id* objects, size_t n, float exposure_start,
                          float exposure_end);

/* camera.rs:18 Camera::look */
int rtg_camera_look(const float look_from[3], const float look_at[3], const float up[3], float fov,
                    float aspect, float aperture, float focus_dist, float exposure_start,
                    float exposure_end, rtg_camera* out);

/* ---- scene: flatten the world once into HBM ---------------------------------------------- */
/* `world` is the `[Box<dyn Object>]` list world of lib.rs:33; a `Bvh` world (lib.rs:51) is the
 * one-element list holding the rtg_object_bvh handle (identical arithmetic). */
int rtg_scene_create(rtg_builder* b, const rtg_id* world, size_t n, int device, rtg_scene** out);
void rtg_scene_destroy(rtg_scene* s);
/* Scheduling / measurement switches of one scene handle -- kernel generation, cost-ordered work queue, pool
 * thresholds, workgroup size (names: DESIGN.md section 4 "Knobs").  None of them changes a bit of the result.  The
 * library reads no environment variable for these.  "bvh4" = 1 (scenes that are ONE Bvh of spheres; RTG_ERR_INVALID
 * otherwise) traverses the reference's tree (bvh.rs:22-120) as 4-wide nodes: the same framebuffer, other
 * rtg_stats.aabb_tests / prim_tests than the reference's walk. */
int rtg_scene_set_option(rtg_scene* s, const char* name, int value);
/* size of the flattened program (for DESIGN.md's byte accounting / tests) */
int rtg_scene_info(const rtg_scene* s, uint32_t* n_instructions, uint32_t* n_materials,
                   uint32_t* n_textures, uint64_t* hbm_bytes);

/* ---- the hot path ------------------------------------------------------------------------- */
/* par_cast (lib.rs:363): out_rgb is caller-owned HOST memory, nx*ny*3 floats, row 0 = top
 * (y = ny-1, lib.rs:328), linear radiance (no gamma).  Synchronous. */
int rtg_par_cast(rtg_scene* s, const rtg_camera* camera, const rtg_params* params, float* out_rgb,
                 rtg_stats* stats_or_null);
/* Same, but out_rgb is DEVICE memory on the scene's device and the kernel is enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream).  Asynchronous unless stats are requested
 * (stats need the kernel to finish).  Frames in flight: a handle owns a ring of launch contexts (work-queue counter,
 * launch constants, cost-ordered queue, scratch, path slots; scene option "frames_in_flight" = 1..4, default 1).  A call
 * takes the next context and first WAITS (on the host) for the frame that used it last, so back-to-back asynchronous calls
 * on one handle are always safe -- with one context they serialise, with n they overlap up to n frames (on different
 * streams).  Each context keeps its own scratch (12 B per pixel and sample up to the budget). */
int rtg_par_cast_device(rtg_scene* s, const rtg_camera* camera, const rtg_params* params,
                        float* d_out_rgb, void* hip_stream, rtg_stats* stats_or_null);

/* Single-process multi-GPU par_cast (SURVEY.md 8b `rtg_render_multi`; reference seam lib.rs:363-376): `scenes[i]` is
 * the SAME world flattened onto device i's HBM (rtg_scene_create with that device index; the scene is small and
 * read-only, so it is replicated).  Scene i renders the pixel tiles (params->tile_w x tile_h; 0 = 16x16, 8x8 from 8 scenes on) with tile_index % n_scenes == i into a
 * zero-filled full frame on its device -- pixels, not samples, are sharded, so every pixel keeps the reference's
 * ordered sample fold -- then ONE collective, ncclReduce(sum) of the float3 framebuffer to the first device over
 * RCCL / xGMI (librccl is dlopen()ed on first use with > 1 distinct device), assembles the frame: x + 0 is exact, the
 * result is bit-identical to rtg_par_cast on one GPU.  Scenes that share a device are summed on that device first.
 * params->rank / nranks must be 0 / 0-or-1 (the call shards by itself).  out_rgb: caller-owned HOST memory.
 * stats: kernel_ms = the slowest shard, counters summed over the shards.  Synchronous.
 * Scene option "multi_gather" = 1 (on any handle) selects the PACKED collective instead: every scene packs the pixels it owns
 * (1 / n_scenes of the frame), grouped ncclSend / ncclRecv bring the packed tiles of the other devices to the first one, which
 * scatters them into its frame -- copies only, bit-identical by construction, 1 / n_scenes of the bytes per device. */
int rtg_par_cast_multi(rtg_scene* const* scenes, int n_scenes, const rtg_camera* camera,
                       const rtg_params* params, float* out_rgb, rtg_stats* stats_or_null);

/* Forget the multi-GPU state of the library: destroy the cached RCCL communicators (ncclCommDestroy), unload librccl,
 * and choose the library the NEXT rtg_par_cast_multi loads: a path / soname, or NULL for the default search (an already
 * loaded librccl first, then librccl.so.1 / librccl.so / /opt/rocm/lib/librccl.so.1).  A library that cannot be loaded makes
 * rtg_par_cast_multi return RTG_ERR_DEVICE with dlopen's reason.  *n_reduces (may be NULL) receives the number of
 * ncclReduce calls issued since the last reset.  Scene option "force_rccl" = 1 sends rtg_par_cast_multi through the RCCL
 * collective even when all handles sit on ONE device (a clique of one), so that one-GPU hosts exercise the same code. */
int rtg_multi_reset(const char* rccl_library_or_null, uint64_t* n_reduces_or_null);

/* ---- output stage -------------------------------------------------------------------------- */
/* print_ppm's per-channel quantisation (lib.rs:348-356): sqrt gamma, `(255.99 * x) as i32` (saturating,
 * NaN -> 0), clamped to 0..=255.  n floats in, n bytes out; both HOST pointers, computed on `device`.
 * The ASCII P3 writer itself stays on the host (rtiow-rust_amd/ppm.py, host/rtiow.hpp). */
int rtg_tonemap(int device, size_t n, const float* rgb, uint8_t* out_u8);
/* Same on DEVICE pointers, enqueued on `hip_stream`. */
int rtg_tonemap_device(int device, size_t n, const float* d_rgb, uint8_t* d_out_u8, void* hip_stream);

/* ---- probes used by the parity tests (not part of the reference surface) ------------------ */
/* One hit_top() (lib.rs:33-49) per ray. rays: n x 7 floats (origin, direction, time).
 * out: n x 8 floats (hit?1:0, t, p.xyz, normal.xyz); out_material: n material handles.
 * Media inside the scene draw from the counter RNG keyed (seed, pixel=i, sample=0). */
int rtg_debug_hit_top(rtg_scene* s, size_t n, const float* rays, uint64_t seed, float t_near,
                      float* out, uint32_t* out_material);
/* One sample of par_cast's closure (lib.rs:366-372) per (x, y, sample) triple, y counted from the
 * bottom as in the reference. out_rgb: n x 3; out_info: n x 4 (bounces, draws, aabb_tests, prim_tests).
 * Default: a one-lane-per-key probe kernel built from the same device functions as the baseline kernel.  With
 * params->flags & RTG_FLAG_TRACE_KERNEL the WHOLE frame (nx, ny, ns, rank / nranks of `params`) is rendered by the
 * instrumented variant of the production kernel par_cast uses for this scene, with a per-sample trace table switched
 * on, and the keys are read out of it -- so a broken schedule can be localised to a (pixel, sample).
 * RTG_ERR_UNSUPPORTED when that kernel is the baseline one (nothing pooled to trace). */
int rtg_debug_samples(rtg_scene* s, const rtg_camera* camera, const rtg_params* params, size_t n,
                      const uint32_t* xs, const uint32_t* ys, const uint32_t* samples,
                      float* out_rgb, uint32_t* out_info);
/* Evaluate the shared libm restatements on the GPU: op 0 = rt_logf, 1 = rt_pow5f, 2 = rt_sinf,
 * 3 = sqrtf, 4 = 1/x, 5 = x/y with y = in2[i] (in2 may be NULL for unary ops). */
int rtg_debug_math(int device, int op, size_t n, const float* in, const float* in2, float* out);

/* Host-only: flatten `world` and copy the flat program out (8 words per instruction: the lo packet
 * then the hi packet, see csrc/flat_scene.h).  Works without a GPU; used by the CPU-side tests. */
int rtg_debug_flatten(rtg_builder* b, const rtg_id* world, size_t n, uint32_t* n_instructions,
                      uint32_t* features, uint32_t* words_out, size_t capacity_instructions);

/* Host-only: the SECOND flat program of `world` -- the one the pool-2 kernel walks (csrc/flat_scene.h "the list level,
 * hoisted": Bvh streams and OP_LIST records, then the records of the list-level items) -- and its item table: 4 words
 * (kind, a, b, c) per item, P2_MAX_ITEMS = 5 items, then (n_items, n_media, n_wrapped, 0): 24 words in `table_out`.
 * *n_instructions = 0 when the world has another shape (such worlds render on the first program only). */
int rtg_debug_flatten_pool2(rtg_builder* b, const rtg_id* world, size_t n, uint32_t* n_instructions, uint32_t* table_out,
                            uint32_t* words_out, size_t capacity_instructions);

#ifdef __cplusplus
}
#endif
#endif /* RTIOW_GPU_H */
