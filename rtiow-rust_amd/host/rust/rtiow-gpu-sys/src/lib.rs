//! Raw bindings to `librtiow_gpu.so` -- every entry point `include/rtiow_gpu.h` declares, in the header's order.
//!
//! UNTESTED: written without a Rust toolchain (none exists in the build image or on the GPU box).  What IS checked:
//! `tests/test_abi.py::test_rust_sys_crate_declares_every_header_symbol` parses this file and the header and compares
//! the symbol sets and the argument counts.  The same entry points are exercised through ctypes and C++ by the test-suite.
//!
//! The boundary replaces `par_cast(nx, ny, ns, &camera, world)` (cbiffle/rtiow-rust `src/lib.rs:363`) and everything
//! below it; the reference itself has no FFI (`#![forbid(unsafe_code)]`, `src/lib.rs:1`).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_int, c_void};

pub const RTG_OK: c_int = 0;
pub const RTG_ERR_INVALID: c_int = -1;
pub const RTG_ERR_EMPTY_BVH: c_int = -2; // bvh.rs:60
pub const RTG_ERR_NAN: c_int = -3; // bvh.rs:45,56
pub const RTG_ERR_RANGE: c_int = -4; // camera.rs:55
pub const RTG_ERR_UNSUPPORTED: c_int = -5;
pub const RTG_ERR_DEVICE: c_int = -6;

pub type rtg_id = u32;
pub const RTG_INVALID_ID: rtg_id = 0xffff_ffff;
pub const RTG_FLAG_COUNTERS: u32 = 1;

#[repr(C)]
pub struct rtg_builder {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rtg_scene {
    _private: [u8; 0],
}

/// camera.rs:6-15 `struct Camera` -- 21 floats.
#[repr(C)]
#[derive(Copy, Clone, Debug, Default, PartialEq)]
pub struct rtg_camera {
    pub origin: [c_float; 3],
    pub lower_left_corner: [c_float; 3],
    pub horizontal: [c_float; 3],
    pub vertical: [c_float; 3],
    pub u: [c_float; 3],
    pub v: [c_float; 3],
    pub lens_radius: c_float,
    pub exposure_start: c_float,
    pub exposure_end: c_float,
}

/// Arguments of par_cast (lib.rs:363) plus the constants the reference bakes in (56 bytes).
#[repr(C)]
#[derive(Copy, Clone, Debug, Default)]
pub struct rtg_params {
    pub struct_size: u32,
    pub nx: u32,
    pub ny: u32,
    pub ns: u32,
    pub max_bounces: u32, // literal 50 at lib.rs:93
    pub t_near: c_float,  // 0.001 at lib.rs:35,53
    pub seed: u64,
    pub tile_w: u32,
    pub tile_h: u32,
    pub rank: u32,
    pub nranks: u32,
    pub flags: u32,
    pub reserved: u32,
}

/// 56 bytes.
#[repr(C)]
#[derive(Copy, Clone, Debug, Default)]
pub struct rtg_stats {
    pub struct_size: u32,
    pub kernel_ms: c_float,
    pub samples: u64,
    pub aabb_tests: u64,
    pub prim_tests: u64,
    pub shaded_hits: u64,
    pub rays: u64,
    pub draws: u64,
}

extern "C" {
    // ---- library
    pub fn rtg_version() -> *const c_char;
    pub fn rtg_last_error() -> *const c_char;
    pub fn rtg_device_count(n: *mut c_int) -> c_int;

    // ---- builder: one call per reference constructor
    pub fn rtg_builder_create(out: *mut *mut rtg_builder) -> c_int;
    pub fn rtg_builder_destroy(b: *mut rtg_builder);
    pub fn rtg_texture_constant(b: *mut rtg_builder, rgb: *const c_float) -> rtg_id; // texture.rs:8
    pub fn rtg_texture_checker(b: *mut rtg_builder, t0: rtg_id, t1: rtg_id) -> rtg_id; // texture.rs:12
    pub fn rtg_texture_perlin(b: *mut rtg_builder, scale: c_float) -> rtg_id; // texture.rs:23
    pub fn rtg_builder_set_perlin_tables(
        b: *mut rtg_builder,
        vecs: *const c_float, // [768]
        perm_x: *const u8,    // [256]
        perm_y: *const u8,
        perm_z: *const u8,
    ) -> c_int; // perlin.rs:24-29
    pub fn rtg_material_lambertian(b: *mut rtg_builder, albedo_texture: rtg_id) -> rtg_id; // material.rs:10-39
    pub fn rtg_material_metal(b: *mut rtg_builder, albedo: *const c_float, fuzz: c_float) -> rtg_id;
    pub fn rtg_material_dielectric(b: *mut rtg_builder, ref_idx: c_float) -> rtg_id;
    pub fn rtg_material_diffuse_light(b: *mut rtg_builder, emission_texture: rtg_id, brightness: c_float) -> rtg_id;
    pub fn rtg_material_isotropic(b: *mut rtg_builder, albedo_texture: rtg_id) -> rtg_id;
    pub fn rtg_object_sphere(b: *mut rtg_builder, radius: c_float, material: rtg_id) -> rtg_id; // object.rs:75
    pub fn rtg_object_rect(
        b: *mut rtg_builder,
        orthogonal_to: c_int, // 0/1/2 = StaticX/Y/Z (object.rs:147-181)
        range0_start: c_float,
        range0_end: c_float,
        range1_start: c_float,
        range1_end: c_float,
        k: c_float,
        material: rtg_id,
    ) -> rtg_id; // object.rs:131
    pub fn rtg_object_flip_normals(b: *mut rtg_builder, object: rtg_id) -> rtg_id; // object.rs:239
    pub fn rtg_object_translate(b: *mut rtg_builder, offset: *const c_float, object: rtg_id) -> rtg_id; // :262
    pub fn rtg_object_scale(b: *mut rtg_builder, factor: *const c_float, object: rtg_id) -> rtg_id; // :296
    pub fn rtg_object_rotate_y(b: *mut rtg_builder, degrees: c_float, object: rtg_id) -> rtg_id; // :477
    pub fn rtg_object_and(b: *mut rtg_builder, object0: rtg_id, object1: rtg_id) -> rtg_id; // :394
    pub fn rtg_object_rect_prism(b: *mut rtg_builder, p0: *const c_float, p1: *const c_float, material: rtg_id) -> rtg_id; // :420
    pub fn rtg_object_linear_move(b: *mut rtg_builder, object: rtg_id, motion: *const c_float) -> rtg_id; // :489
    pub fn rtg_object_constant_medium(b: *mut rtg_builder, boundary: rtg_id, density: c_float, material: rtg_id) -> rtg_id; // :533
    pub fn rtg_object_bvh(b: *mut rtg_builder, objects: *const rtg_id, n: usize, exposure_start: c_float, exposure_end: c_float) -> rtg_id; // bvh.rs:128
    /// NOT in the reference: SAH-built Bvh (same image up to exact-t ties, fewer Aabb tests).
    pub fn rtg_object_bvh_sah(b: *mut rtg_builder, objects: *const rtg_id, n: usize, exposure_start: c_float, exposure_end: c_float) -> rtg_id;
    pub fn rtg_camera_look(
        look_from: *const c_float,
        look_at: *const c_float,
        up: *const c_float,
        fov: c_float,
        aspect: c_float,
        aperture: c_float,
        focus_dist: c_float,
        exposure_start: c_float,
        exposure_end: c_float,
        out: *mut rtg_camera,
    ) -> c_int; // camera.rs:18

    // ---- scene
    pub fn rtg_scene_create(b: *mut rtg_builder, world: *const rtg_id, n: usize, device: c_int, out: *mut *mut rtg_scene) -> c_int;
    pub fn rtg_scene_destroy(s: *mut rtg_scene);
    pub fn rtg_scene_set_option(s: *mut rtg_scene, name: *const c_char, value: c_int) -> c_int;
    pub fn rtg_scene_info(s: *const rtg_scene, n_instructions: *mut u32, n_materials: *mut u32, n_textures: *mut u32, hbm_bytes: *mut u64) -> c_int;

    // ---- the hot path
    pub fn rtg_par_cast(s: *mut rtg_scene, camera: *const rtg_camera, params: *const rtg_params, out_rgb: *mut c_float, stats_or_null: *mut rtg_stats) -> c_int;
    pub fn rtg_par_cast_device(
        s: *mut rtg_scene,
        camera: *const rtg_camera,
        params: *const rtg_params,
        d_out_rgb: *mut c_float,
        hip_stream: *mut c_void,
        stats_or_null: *mut rtg_stats,
    ) -> c_int;
    pub fn rtg_par_cast_multi(
        scenes: *const *mut rtg_scene,
        n_scenes: c_int,
        camera: *const rtg_camera,
        params: *const rtg_params,
        out_rgb: *mut c_float,
        stats_or_null: *mut rtg_stats,
    ) -> c_int;

    /// Drop the cached RCCL communicators, unload librccl, choose the library the next `rtg_par_cast_multi` loads.
    pub fn rtg_multi_reset(rccl_library_or_null: *const c_char, n_reduces_or_null: *mut u64) -> c_int;

    // ---- output stage (print_ppm's quantisation, lib.rs:348-356)
    pub fn rtg_tonemap(device: c_int, n: usize, rgb: *const c_float, out_u8: *mut u8) -> c_int;
    pub fn rtg_tonemap_device(device: c_int, n: usize, d_rgb: *const c_float, d_out_u8: *mut u8, hip_stream: *mut c_void) -> c_int;

    // ---- probes used by the parity tests
    pub fn rtg_debug_hit_top(s: *mut rtg_scene, n: usize, rays: *const c_float, seed: u64, t_near: c_float, out: *mut c_float, out_material: *mut u32) -> c_int;
    pub fn rtg_debug_samples(
        s: *mut rtg_scene,
        camera: *const rtg_camera,
        params: *const rtg_params,
        n: usize,
        xs: *const u32,
        ys: *const u32,
        samples: *const u32,
        out_rgb: *mut c_float,
        out_info: *mut u32,
    ) -> c_int;
    pub fn rtg_debug_math(device: c_int, op: c_int, n: usize, input: *const c_float, input2: *const c_float, out: *mut c_float) -> c_int;
    pub fn rtg_debug_flatten(
        b: *mut rtg_builder,
        world: *const rtg_id,
        n: usize,
        n_instructions: *mut u32,
        features: *mut u32,
        words_out: *mut u32,
        capacity_instructions: usize,
    ) -> c_int;
    pub fn rtg_debug_flatten_pool2(
        b: *mut rtg_builder,
        world: *const rtg_id,
        n: usize,
        n_instructions: *mut u32,
        table_out: *mut u32,
        words_out: *mut u32,
        capacity_instructions: usize,
    ) -> c_int;
}
