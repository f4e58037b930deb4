//! Raw bindings to `librtiow_gpu.so` (C ABI declared in `include/rtiow_gpu.h`).
//! UNTESTED: written without a Rust toolchain; mirrors the header entry for entry for the path
//! `par_cast -> color -> hit_top` of cbiffle/rtiow-rust (src/lib.rs:363).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_int, c_void};

pub type rtg_id = u32;
pub const RTG_INVALID_ID: rtg_id = 0xffff_ffff;
#[repr(C)] pub struct rtg_builder { _p: [u8; 0] }
#[repr(C)] pub struct rtg_scene   { _p: [u8; 0] }

#[repr(C)] #[derive(Copy, Clone, Default)]
pub struct rtg_camera {                     // camera.rs:6-15
    pub origin: [c_float; 3], pub lower_left_corner: [c_float; 3],
    pub horizontal: [c_float; 3], pub vertical: [c_float; 3],
    pub u: [c_float; 3], pub v: [c_float; 3],
    pub lens_radius: c_float, pub exposure_start: c_float, pub exposure_end: c_float,
}
#[repr(C)] #[derive(Copy, Clone, Default)]
pub struct rtg_params {
    pub struct_size: u32, pub nx: u32, pub ny: u32, pub ns: u32,
    pub max_bounces: u32, pub t_near: c_float, pub seed: u64,
    pub tile_w: u32, pub tile_h: u32, pub rank: u32, pub nranks: u32, pub flags: u32, pub reserved: u32,
}

extern "C" {
    pub fn rtg_last_error() -> *const c_char;
    pub fn rtg_builder_create(out: *mut *mut rtg_builder) -> c_int;
    pub fn rtg_builder_destroy(b: *mut rtg_builder);
    pub fn rtg_texture_constant(b: *mut rtg_builder, rgb: *const c_float) -> rtg_id;     // texture.rs:8
    pub fn rtg_texture_checker(b: *mut rtg_builder, t0: rtg_id, t1: rtg_id) -> rtg_id;   // texture.rs:12
    pub fn rtg_texture_perlin(b: *mut rtg_builder, scale: c_float) -> rtg_id;            // texture.rs:23
    pub fn rtg_builder_set_perlin_tables(b: *mut rtg_builder, vecs: *const c_float,
        px: *const u8, py: *const u8, pz: *const u8) -> c_int;                           // perlin.rs:24-29
    pub fn rtg_material_lambertian(b: *mut rtg_builder, albedo: rtg_id) -> rtg_id;       // material.rs:15
    pub fn rtg_material_metal(b: *mut rtg_builder, albedo: *const c_float, fuzz: c_float) -> rtg_id;
    pub fn rtg_material_dielectric(b: *mut rtg_builder, ref_idx: c_float) -> rtg_id;
    pub fn rtg_material_diffuse_light(b: *mut rtg_builder, emission: rtg_id, brightness: c_float) -> rtg_id;
    pub fn rtg_material_isotropic(b: *mut rtg_builder, albedo: rtg_id) -> rtg_id;
    pub fn rtg_object_sphere(b: *mut rtg_builder, radius: c_float, material: rtg_id) -> rtg_id;
    pub fn rtg_object_rect(b: *mut rtg_builder, axis: c_int, r0s: c_float, r0e: c_float,
        r1s: c_float, r1e: c_float, k: c_float, material: rtg_id) -> rtg_id;
    pub fn rtg_object_flip_normals(b: *mut rtg_builder, o: rtg_id) -> rtg_id;
    pub fn rtg_object_translate(b: *mut rtg_builder, offset: *const c_float, o: rtg_id) -> rtg_id;
    pub fn rtg_object_scale(b: *mut rtg_builder, factor: *const c_float, o: rtg_id) -> rtg_id;
    pub fn rtg_object_rotate_y(b: *mut rtg_builder, degrees: c_float, o: rtg_id) -> rtg_id;
    pub fn rtg_object_and(b: *mut rtg_builder, o0: rtg_id, o1: rtg_id) -> rtg_id;
    pub fn rtg_object_linear_move(b: *mut rtg_builder, o: rtg_id, motion: *const c_float) -> rtg_id;
    pub fn rtg_object_constant_medium(b: *mut rtg_builder, boundary: rtg_id, density: c_float, m: rtg_id) -> rtg_id;
    pub fn rtg_object_bvh(b: *mut rtg_builder, objs: *const rtg_id, n: usize, e0: c_float, e1: c_float) -> rtg_id;
    pub fn rtg_scene_create(b: *mut rtg_builder, world: *const rtg_id, n: usize, device: c_int,
        out: *mut *mut rtg_scene) -> c_int;
    pub fn rtg_scene_destroy(s: *mut rtg_scene);
    pub fn rtg_par_cast(s: *mut rtg_scene, cam: *const rtg_camera, p: *const rtg_params,
        out_rgb: *mut c_float, stats: *mut c_void) -> c_int;
}
