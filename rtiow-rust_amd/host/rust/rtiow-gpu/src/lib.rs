//! Safe Rust side of the boundary: `gpu_cast(nx, ny, ns, &camera, world)` with the shape of the reference's
//! `par_cast(nx, ny, ns, &camera, world)` (cbiffle/rtiow-rust `src/lib.rs:363`).
//!
//! UNTESTED -- no Rust toolchain exists in the build image; this is the source a maintainer would start from.
//! The identical call sequence is exercised by the test-suite through C++ (`host/rtiow.hpp`, whose `par_cast` this file
//! transliterates) and ctypes (`capi.py`).
//!
//! How the reference plugs in: its objects are opaque `Box<dyn Object>` values, so `trait Object` (object.rs:15)
//! gains ONE method, `fn flatten(&self, b: &mut GpuBuilder) -> Result<ObjectId>`, with one-line impls that call the
//! builder method of the same name; `World` types implement [`FlattenWorld`].
use rtiow_gpu_sys as sys;
use std::ffi::CStr;
use std::ops::Range;
use std::ptr;

#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rtiow-gpu error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}
pub type Result<T> = std::result::Result<T, Error>;

fn last_error(code: i32) -> Error {
    // rtg_last_error() is thread-local and never null
    let message = unsafe { CStr::from_ptr(sys::rtg_last_error()) }.to_string_lossy().into_owned();
    Error { code, message }
}
fn check(rc: i32) -> Result<()> {
    if rc == sys::RTG_OK {
        Ok(())
    } else {
        Err(last_error(rc))
    }
}

#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub struct TextureId(sys::rtg_id);
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub struct MaterialId(sys::rtg_id);
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub struct ObjectId(sys::rtg_id);

/// object.rs:147-181 `StaticX / StaticY / StaticZ`
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub enum Axis {
    X = 0,
    Y = 1,
    Z = 2,
}

pub type Vec3 = [f32; 3];

/// A scene under construction: one method per reference constructor.
pub struct GpuBuilder {
    raw: *mut sys::rtg_builder,
}
impl GpuBuilder {
    pub fn new() -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::rtg_builder_create(&mut raw) })?;
        Ok(GpuBuilder { raw })
    }
    fn id(&self, id: sys::rtg_id) -> Result<sys::rtg_id> {
        if id == sys::RTG_INVALID_ID {
            Err(last_error(sys::RTG_ERR_INVALID))
        } else {
            Ok(id)
        }
    }
    // texture.rs
    pub fn constant(&mut self, color: Vec3) -> Result<TextureId> {
        self.id(unsafe { sys::rtg_texture_constant(self.raw, color.as_ptr()) }).map(TextureId)
    }
    pub fn checker(&mut self, t0: TextureId, t1: TextureId) -> Result<TextureId> {
        self.id(unsafe { sys::rtg_texture_checker(self.raw, t0.0, t1.0) }).map(TextureId)
    }
    pub fn perlin(&mut self, scale: f32) -> Result<TextureId> {
        self.id(unsafe { sys::rtg_texture_perlin(self.raw, scale) }).map(TextureId)
    }
    /// perlin.rs:24-29 (the reference seeds these from thread_rng; any fixed tables are admissible)
    pub fn set_perlin_tables(&mut self, vecs: &[Vec3; 256], px: &[u8; 256], py: &[u8; 256], pz: &[u8; 256]) -> Result<()> {
        check(unsafe { sys::rtg_builder_set_perlin_tables(self.raw, vecs.as_ptr() as *const f32, px.as_ptr(), py.as_ptr(), pz.as_ptr()) })
    }
    // material.rs:10-39
    pub fn lambertian(&mut self, albedo: TextureId) -> Result<MaterialId> {
        self.id(unsafe { sys::rtg_material_lambertian(self.raw, albedo.0) }).map(MaterialId)
    }
    pub fn metal(&mut self, albedo: Vec3, fuzz: f32) -> Result<MaterialId> {
        self.id(unsafe { sys::rtg_material_metal(self.raw, albedo.as_ptr(), fuzz) }).map(MaterialId)
    }
    pub fn dielectric(&mut self, ref_idx: f32) -> Result<MaterialId> {
        self.id(unsafe { sys::rtg_material_dielectric(self.raw, ref_idx) }).map(MaterialId)
    }
    pub fn diffuse_light(&mut self, emission: TextureId, brightness: f32) -> Result<MaterialId> {
        self.id(unsafe { sys::rtg_material_diffuse_light(self.raw, emission.0, brightness) }).map(MaterialId)
    }
    pub fn isotropic(&mut self, albedo: TextureId) -> Result<MaterialId> {
        self.id(unsafe { sys::rtg_material_isotropic(self.raw, albedo.0) }).map(MaterialId)
    }
    // object.rs / bvh.rs
    pub fn sphere(&mut self, radius: f32, material: MaterialId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_sphere(self.raw, radius, material.0) }).map(ObjectId)
    }
    pub fn rect(&mut self, orthogonal_to: Axis, range0: Range<f32>, range1: Range<f32>, k: f32, material: MaterialId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_rect(self.raw, orthogonal_to as i32, range0.start, range0.end, range1.start, range1.end, k, material.0) })
            .map(ObjectId)
    }
    pub fn flip_normals(&mut self, object: ObjectId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_flip_normals(self.raw, object.0) }).map(ObjectId)
    }
    pub fn translate(&mut self, offset: Vec3, object: ObjectId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_translate(self.raw, offset.as_ptr(), object.0) }).map(ObjectId)
    }
    pub fn scale(&mut self, factor: Vec3, object: ObjectId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_scale(self.raw, factor.as_ptr(), object.0) }).map(ObjectId)
    }
    pub fn rotate_y(&mut self, degrees: f32, object: ObjectId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_rotate_y(self.raw, degrees, object.0) }).map(ObjectId)
    }
    pub fn and(&mut self, a: ObjectId, b: ObjectId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_and(self.raw, a.0, b.0) }).map(ObjectId)
    }
    pub fn rect_prism(&mut self, p0: Vec3, p1: Vec3, material: MaterialId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_rect_prism(self.raw, p0.as_ptr(), p1.as_ptr(), material.0) }).map(ObjectId)
    }
    pub fn linear_move(&mut self, object: ObjectId, motion: Vec3) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_linear_move(self.raw, object.0, motion.as_ptr()) }).map(ObjectId)
    }
    pub fn constant_medium(&mut self, boundary: ObjectId, density: f32, material: MaterialId) -> Result<ObjectId> {
        self.id(unsafe { sys::rtg_object_constant_medium(self.raw, boundary.0, density, material.0) }).map(ObjectId)
    }
    /// bvh::from_scene (bvh.rs:128): the library runs the same widest-axis median split over the leaves, in order.
    pub fn bvh(&mut self, objects: &[ObjectId], exposure: Range<f32>) -> Result<ObjectId> {
        let ids: Vec<sys::rtg_id> = objects.iter().map(|o| o.0).collect();
        self.id(unsafe { sys::rtg_object_bvh(self.raw, ids.as_ptr(), ids.len(), exposure.start, exposure.end) }).map(ObjectId)
    }
    /// Not in the reference: SAH split (same image up to exact-t ties, fewer Aabb tests).
    pub fn bvh_sah(&mut self, objects: &[ObjectId], exposure: Range<f32>) -> Result<ObjectId> {
        let ids: Vec<sys::rtg_id> = objects.iter().map(|o| o.0).collect();
        self.id(unsafe { sys::rtg_object_bvh_sah(self.raw, ids.as_ptr(), ids.len(), exposure.start, exposure.end) }).map(ObjectId)
    }
    /// Flatten `world` (the `[Box<dyn Object>]` of lib.rs:33, or a one-element list holding a Bvh) once into `device`'s HBM.
    pub fn scene(&self, world: &[ObjectId], device: i32) -> Result<GpuScene> {
        let ids: Vec<sys::rtg_id> = world.iter().map(|o| o.0).collect();
        let mut raw = ptr::null_mut();
        check(unsafe { sys::rtg_scene_create(self.raw, ids.as_ptr(), ids.len(), device, &mut raw) })?;
        Ok(GpuScene { raw })
    }
}
impl Drop for GpuBuilder {
    fn drop(&mut self) {
        unsafe { sys::rtg_builder_destroy(self.raw) }
    }
}

/// camera.rs:6-15; built by the library so that `tan` / rounding order match the GPU path's checker exactly.
#[derive(Copy, Clone, Debug)]
pub struct Camera(pub sys::rtg_camera);
impl Camera {
    /// Camera::look, camera.rs:18-50
    #[allow(clippy::too_many_arguments)]
    pub fn look(look_from: Vec3, look_at: Vec3, up: Vec3, fov: f32, aspect: f32, aperture: f32, focus_dist: f32, exposure: Range<f32>) -> Result<Camera> {
        let mut c = sys::rtg_camera::default();
        check(unsafe {
            sys::rtg_camera_look(look_from.as_ptr(), look_at.as_ptr(), up.as_ptr(), fov, aspect, aperture, focus_dist, exposure.start, exposure.end, &mut c)
        })?;
        Ok(Camera(c))
    }
}

/// A flattened scene resident in one GPU's HBM.  One frame in flight per scene; different scenes are independent.
pub struct GpuScene {
    raw: *mut sys::rtg_scene,
}
unsafe impl Send for GpuScene {}
impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { sys::rtg_scene_destroy(self.raw) }
    }
}

/// lib.rs:321 `Image(Vec<Vec<Vec3>>)`: row 0 = top scanline (lib.rs:328), linear radiance.
pub struct Image {
    pub nx: usize,
    pub ny: usize,
    pub rgb: Vec<f32>,
}
impl Image {
    pub fn pixel(&self, x: usize, row: usize) -> Vec3 {
        let i = 3 * (row * self.nx + x);
        [self.rgb[i], self.rgb[i + 1], self.rgb[i + 2]]
    }
    /// print_ppm's per-channel quantisation (lib.rs:348-356), computed on `device`.
    pub fn to_u8(&self, device: i32) -> Result<Vec<u8>> {
        let mut out = vec![0u8; self.rgb.len()];
        check(unsafe { sys::rtg_tonemap(device, self.rgb.len(), self.rgb.as_ptr(), out.as_mut_ptr()) })?;
        Ok(out)
    }
}

/// What par_cast bakes in (bounce cap 50 at lib.rs:93, NEAR = 0.001 at lib.rs:35) plus the counter-RNG seed.
#[derive(Copy, Clone, Debug)]
pub struct CastOptions {
    pub seed: u64,
    pub max_bounces: u32,
    pub t_near: f32,
}
impl Default for CastOptions {
    fn default() -> Self {
        CastOptions { seed: 0xDEAD_BEEF, max_bounces: 50, t_near: 0.001 }
    }
}
fn params(nx: usize, ny: usize, ns: usize, o: &CastOptions) -> sys::rtg_params {
    sys::rtg_params {
        struct_size: std::mem::size_of::<sys::rtg_params>() as u32,
        nx: nx as u32,
        ny: ny as u32,
        ns: ns as u32,
        max_bounces: o.max_bounces,
        t_near: o.t_near,
        seed: o.seed,
        nranks: 1,
        ..Default::default()
    }
}

impl GpuScene {
    /// Scheduling / resource switches of this handle (`rtg_scene_set_option`; none changes a bit of the result): e.g.
    /// `("scratch_mb", 4096)` bounds the per-sample colour scratch (larger frames render in sample passes),
    /// `("frames_in_flight", 2)` lets two asynchronous frames of this handle overlap, `("force_rccl", 1)` sends
    /// `gpu_cast_multi` through the RCCL collective even on one device.
    pub fn set_option(&mut self, name: &str, value: i32) -> Result<()> {
        let c = std::ffi::CString::new(name).map_err(|_| last_error(sys::RTG_ERR_INVALID))?;
        check(unsafe { sys::rtg_scene_set_option(self.raw, c.as_ptr(), value) })
    }

    /// par_cast (lib.rs:363) on this scene's GPU.
    pub fn par_cast(&mut self, nx: usize, ny: usize, ns: usize, camera: &Camera, options: &CastOptions) -> Result<Image> {
        let mut rgb = vec![0f32; nx * ny * 3];
        let p = params(nx, ny, ns, options);
        check(unsafe { sys::rtg_par_cast(self.raw, &camera.0, &p, rgb.as_mut_ptr(), ptr::null_mut()) })?;
        Ok(Image { nx, ny, rgb })
    }
}

/// Implemented by the reference's `World` types (lib.rs:23-55): `[Box<dyn Object>]` flattens each object in order,
/// `Bvh` flattens its leaves and calls `GpuBuilder::bvh`.
pub trait FlattenWorld {
    fn flatten_world(&self, b: &mut GpuBuilder) -> Result<Vec<ObjectId>>;
}

/// Drop-in for `par_cast(nx, ny, ns, &camera, world)` (lib.rs:363): flatten once, render on GPU 0.
pub fn gpu_cast<W: FlattenWorld + ?Sized>(nx: usize, ny: usize, ns: usize, camera: &Camera, world: &W) -> Result<Image> {
    let mut b = GpuBuilder::new()?;
    let ids = world.flatten_world(&mut b)?;
    b.scene(&ids, 0)?.par_cast(nx, ny, ns, camera, &CastOptions::default())
}

/// The same on every GPU of the node: the world is flattened onto each device, pixel tiles are sharded over them and
/// ONE RCCL reduce(sum) of the float3 framebuffer (inside the library) assembles the frame -- bit-identical to `gpu_cast`.
pub fn gpu_cast_multi<W: FlattenWorld + ?Sized>(nx: usize, ny: usize, ns: usize, camera: &Camera, world: &W, n_devices: i32) -> Result<Image> {
    let mut b = GpuBuilder::new()?;
    let ids = world.flatten_world(&mut b)?;
    let scenes: Vec<GpuScene> = (0..n_devices).map(|d| b.scene(&ids, d)).collect::<Result<_>>()?;
    let raws: Vec<*mut sys::rtg_scene> = scenes.iter().map(|s| s.raw).collect();
    let mut rgb = vec![0f32; nx * ny * 3];
    let p = params(nx, ny, ns, &CastOptions::default());
    check(unsafe { sys::rtg_par_cast_multi(raws.as_ptr(), raws.len() as i32, &camera.0, &p, rgb.as_mut_ptr(), ptr::null_mut()) })?;
    Ok(Image { nx, ny, rgb })
}

/// Forget the library's multi-GPU state: destroy the cached RCCL communicators, unload librccl (the next `gpu_cast_multi`
/// loads it again).  Returns the number of `ncclReduce` calls issued since the last reset.
pub fn multi_reset() -> Result<u64> {
    let mut n = 0u64;
    check(unsafe { sys::rtg_multi_reset(ptr::null(), &mut n) })?;
    Ok(n)
}

pub fn device_count() -> Result<i32> {
    let mut n = 0;
    check(unsafe { sys::rtg_device_count(&mut n) })?;
    Ok(n)
}
