"""ctypes binding of the C ABI declared in include/rtiow_gpu.h (librtiow_gpu.so, prefix ``rtg_``).

Method names mirror the reference crate's constructors (object.rs / material.rs / texture.rs /
camera.rs / lib.rs) so scene code reads like the reference's `src/main.rs`.  The symbol prefix is a
parameter so that a test harness can drive another library exporting the same entry points with the
same calls.
"""
import ctypes as C
import os

import numpy as np

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)
c_u8p = C.POINTER(C.c_uint8)

INVALID_ID = 0xFFFFFFFF
FLAG_COUNTERS = 1
FLAG_TRACE_KERNEL = 2


class Camera(C.Structure):
    """camera.rs:6-15"""
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lens_radius", C.c_float), ("exposure_start", C.c_float), ("exposure_end", C.c_float)]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("ns", C.c_uint32),
                ("max_bounces", C.c_uint32), ("t_near", C.c_float), ("seed", C.c_uint64),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("rank", C.c_uint32),
                ("nranks", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kernel_ms", C.c_float), ("samples", C.c_uint64),
                ("aabb_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("rays", C.c_uint64), ("draws", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "struct_size"}


def make_params(nx, ny, ns, seed=0xDEADBEEF, max_bounces=50, t_near=0.001, tile_w=0, tile_h=0, rank=0,
                nranks=1, flags=0):
    p = Params()
    p.struct_size = C.sizeof(Params)
    p.nx, p.ny, p.ns = nx, ny, ns
    p.max_bounces = max_bounces
    p.t_near = t_near
    p.seed = seed
    p.tile_w, p.tile_h, p.rank, p.nranks, p.flags = tile_w, tile_h, rank, nranks, flags
    return p


# error codes of include/rtiow_gpu.h
ERR_INVALID, ERR_EMPTY_BVH, ERR_NAN, ERR_RANGE, ERR_UNSUPPORTED, ERR_DEVICE = -1, -2, -3, -4, -5, -6


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt error %d: %s" % (code, msg))
        self.code = code


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


# every symbol include/rtiow_gpu.h declares (suffix after the prefix); test_abi checks all of them
ABI_SYMBOLS = [
    "version", "last_error", "device_count", "builder_create", "builder_destroy",
    "texture_constant", "texture_checker", "texture_perlin", "builder_set_perlin_tables",
    "material_lambertian", "material_metal", "material_dielectric", "material_diffuse_light",
    "material_isotropic", "object_sphere", "object_rect", "object_flip_normals", "object_translate",
    "object_scale", "object_rotate_y", "object_and", "object_rect_prism", "object_linear_move",
    "object_constant_medium", "object_bvh", "object_bvh_sah", "camera_look", "scene_create", "scene_destroy",
    "scene_set_option", "scene_info", "par_cast", "par_cast_device", "par_cast_multi", "multi_reset", "debug_hit_top", "debug_samples", "debug_math", "debug_flatten", "debug_flatten_pool2", "tonemap", "tonemap_device",
]


class Backend:
    """A loaded library + symbol prefix."""

    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found -- build it first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        f = self._fn
        f("last_error", C.c_char_p, [])
        f("version", C.c_char_p, [])
        f("builder_create", C.c_int, [C.POINTER(C.c_void_p)])
        f("builder_destroy", None, [C.c_void_p])
        f("texture_constant", C.c_uint32, [C.c_void_p, c_f32p])
        f("texture_checker", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32])
        f("texture_perlin", C.c_uint32, [C.c_void_p, C.c_float])
        f("builder_set_perlin_tables", C.c_int, [C.c_void_p, c_f32p, c_u8p, c_u8p, c_u8p])
        f("material_lambertian", C.c_uint32, [C.c_void_p, C.c_uint32])
        f("material_metal", C.c_uint32, [C.c_void_p, c_f32p, C.c_float])
        f("material_dielectric", C.c_uint32, [C.c_void_p, C.c_float])
        f("material_diffuse_light", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_float])
        f("material_isotropic", C.c_uint32, [C.c_void_p, C.c_uint32])
        f("object_sphere", C.c_uint32, [C.c_void_p, C.c_float, C.c_uint32])
        f("object_rect", C.c_uint32, [C.c_void_p, C.c_int] + [C.c_float] * 5 + [C.c_uint32])
        f("object_flip_normals", C.c_uint32, [C.c_void_p, C.c_uint32])
        f("object_translate", C.c_uint32, [C.c_void_p, c_f32p, C.c_uint32])
        f("object_scale", C.c_uint32, [C.c_void_p, c_f32p, C.c_uint32])
        f("object_rotate_y", C.c_uint32, [C.c_void_p, C.c_float, C.c_uint32])
        f("object_and", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32])
        f("object_rect_prism", C.c_uint32, [C.c_void_p, c_f32p, c_f32p, C.c_uint32])
        f("object_linear_move", C.c_uint32, [C.c_void_p, C.c_uint32, c_f32p])
        f("object_constant_medium", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint32])
        f("object_bvh", C.c_uint32, [C.c_void_p, c_u32p, C.c_size_t, C.c_float, C.c_float])
        f("object_bvh_sah", C.c_uint32, [C.c_void_p, c_u32p, C.c_size_t, C.c_float, C.c_float])
        f("camera_look", C.c_int, [c_f32p, c_f32p, c_f32p] + [C.c_float] * 6 + [C.POINTER(Camera)])
        f("scene_create", C.c_int, [C.c_void_p, c_u32p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)])
        f("scene_destroy", None, [C.c_void_p])
        self._declare_render()
        f("debug_hit_top", C.c_int, [C.c_void_p, C.c_size_t, c_f32p, C.c_uint64, C.c_float, c_f32p, c_u32p])
        f("debug_samples", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_size_t,
                                     c_u32p, c_u32p, c_u32p, c_f32p, c_u32p])
        f("debug_math", C.c_int, [C.c_int, C.c_int, C.c_size_t, c_f32p, c_f32p, c_f32p])
        f("tonemap", C.c_int, [C.c_int, C.c_size_t, c_f32p, c_u8p])

    def _declare_render(self):
        f = self._fn
        f("device_count", C.c_int, [C.POINTER(C.c_int)])
        f("scene_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int])
        f("scene_info", C.c_int, [C.c_void_p, c_u32p, c_u32p, c_u32p, C.POINTER(C.c_uint64)])
        f("par_cast", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), c_f32p, C.POINTER(Stats)])
        f("par_cast_device", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p,
                                       C.c_void_p, C.POINTER(Stats)])
        f("par_cast_multi", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Camera), C.POINTER(Params), c_f32p,
                                      C.POINTER(Stats)])
        f("multi_reset", C.c_int, [C.c_char_p, C.POINTER(C.c_uint64)])
        f("debug_flatten", C.c_int, [C.c_void_p, c_u32p, C.c_size_t, c_u32p, c_u32p, c_u32p, C.c_size_t])
        f("debug_flatten_pool2", C.c_int, [C.c_void_p, c_u32p, C.c_size_t, c_u32p, c_u32p, c_u32p, C.c_size_t])
        f("tonemap_device", C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p])

    def _fn(self, name, restype, argtypes):
        try:
            fn = getattr(self.lib, self.prefix + name)
        except AttributeError:   # an OLDER build of the library (RTIOW_GPU_LIB A/B runs): the call fails when it is made
            def fn(*_a, _n=self.prefix + name):
                raise RtError(ERR_INVALID, "%s: symbol %s is not exported by this build" % (self.path, _n))
            setattr(self, "_" + name, fn)
            return fn
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, "_" + name, fn)
        return fn

    def last_error(self):
        return (self._last_error() or b"").decode()

    def check(self, code):
        if code != 0:
            raise RtError(code, self.last_error())

    def check_id(self, i):
        if i == INVALID_ID:
            raise RtError(-1, self.last_error())
        return i

    def builder(self):
        return Builder(self)

    def device_count(self):
        n = C.c_int(0)
        self.check(self._device_count(C.byref(n)))
        return n.value

    def camera_look(self, look_from, look_at, up, fov, aspect, aperture, focus_dist, exposure=(0.0, 1.0)):
        """Camera::look, camera.rs:18-50"""
        cam = Camera()
        self.check(self._camera_look(_f3(look_from), _f3(look_at), _f3(up), fov, aspect, aperture,
                                     focus_dist, exposure[0], exposure[1], C.byref(cam)))
        return cam

    def par_cast_multi(self, scenes, camera, nx, ny, ns, seed=0xDEADBEEF, stats=False, **kw):
        """rtg_par_cast_multi: one scene handle per device (the same world flattened on each), tiles sharded over
        them, ONE RCCL reduce(sum) of the float3 framebuffer inside the library.  Returns the assembled frame."""
        p = make_params(nx, ny, ns, seed=seed, flags=FLAG_COUNTERS if stats else 0, **kw)
        out = np.zeros((ny, nx, 3), dtype=np.float32)
        st = Stats()
        st.struct_size = C.sizeof(Stats)
        arr = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
        self.check(self._par_cast_multi(arr, len(scenes), C.byref(camera), C.byref(p), out.ctypes.data_as(c_f32p),
                                        C.byref(st)))
        return (out, st.as_dict()) if stats else out

    def multi_reset(self, rccl_library=None):
        """rtg_multi_reset: drop the cached RCCL communicators, unload librccl, choose the library to load next (None =
        default search).  Returns the number of ncclReduce calls issued since the last reset."""
        n = C.c_uint64(0)
        self.check(self._multi_reset(rccl_library.encode() if rccl_library else None, C.byref(n)))
        return n.value

    def tonemap(self, img, device=0):
        """print_ppm's sqrt-gamma + `(255.99 * x) as i32` clamp (lib.rs:348-356) -> uint8 array of img's shape."""
        x = np.ascontiguousarray(img, dtype=np.float32)
        out = np.empty(x.shape, dtype=np.uint8)
        self.check(self._tonemap(device, x.size, x.ctypes.data_as(c_f32p), out.ctypes.data_as(c_u8p)))
        return out

    def debug_math(self, op, x, y=None, device=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32)
            yp = y.ctypes.data_as(c_f32p)
        self.check(self._debug_math(device, op, x.size, x.ctypes.data_as(c_f32p), yp,
                                    out.ctypes.data_as(c_f32p)))
        return out


class Builder:
    """Scene under construction.  One method per reference constructor."""

    def __init__(self, backend):
        self.be = backend
        h = C.c_void_p()
        backend.check(backend._builder_create(C.byref(h)))
        self.h = h
        self._scenes = []

    def close(self):
        if self.h:
            self.be._builder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # texture.rs
    def constant(self, color):
        return self.be.check_id(self.be._texture_constant(self.h, _f3(color)))

    def checker(self, t0, t1):
        return self.be.check_id(self.be._texture_checker(self.h, t0, t1))

    def perlin(self, scale):
        return self.be.check_id(self.be._texture_perlin(self.h, scale))

    def set_perlin_tables(self, vecs, perm_x, perm_y, perm_z):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32).reshape(768)
        px, py, pz = (np.ascontiguousarray(a, dtype=np.uint8) for a in (perm_x, perm_y, perm_z))
        self.be.check(self.be._builder_set_perlin_tables(
            self.h, vecs.ctypes.data_as(c_f32p), px.ctypes.data_as(c_u8p), py.ctypes.data_as(c_u8p),
            pz.ctypes.data_as(c_u8p)))

    # material.rs
    def lambertian(self, albedo):
        return self.be.check_id(self.be._material_lambertian(self.h, albedo))

    def metal(self, albedo, fuzz):
        return self.be.check_id(self.be._material_metal(self.h, _f3(albedo), fuzz))

    def dielectric(self, ref_idx):
        return self.be.check_id(self.be._material_dielectric(self.h, ref_idx))

    def diffuse_light(self, emission, brightness):
        return self.be.check_id(self.be._material_diffuse_light(self.h, emission, brightness))

    def isotropic(self, albedo):
        return self.be.check_id(self.be._material_isotropic(self.h, albedo))

    # object.rs / bvh.rs
    def sphere(self, radius, material):
        return self.be.check_id(self.be._object_sphere(self.h, radius, material))

    def rect(self, orthogonal_to, range0, range1, k, material):
        return self.be.check_id(self.be._object_rect(self.h, orthogonal_to, range0[0], range0[1],
                                                     range1[0], range1[1], k, material))

    def flip_normals(self, obj):
        return self.be.check_id(self.be._object_flip_normals(self.h, obj))

    def translate(self, offset, obj):
        return self.be.check_id(self.be._object_translate(self.h, _f3(offset), obj))

    def scale(self, factor, obj):
        return self.be.check_id(self.be._object_scale(self.h, _f3(factor), obj))

    def rotate_y(self, degrees, obj):
        return self.be.check_id(self.be._object_rotate_y(self.h, degrees, obj))

    def and_(self, a, b):
        return self.be.check_id(self.be._object_and(self.h, a, b))

    def rect_prism(self, p0, p1, material):
        return self.be.check_id(self.be._object_rect_prism(self.h, _f3(p0), _f3(p1), material))

    def linear_move(self, obj, motion):
        return self.be.check_id(self.be._object_linear_move(self.h, obj, _f3(motion)))

    def constant_medium(self, boundary, density, material):
        return self.be.check_id(self.be._object_constant_medium(self.h, boundary, density, material))

    def bvh(self, objs, exposure=(0.0, 1.0)):
        """bvh::from_scene, bvh.rs:128"""
        arr = (C.c_uint32 * max(1, len(objs)))(*objs)
        return self.be.check_id(self.be._object_bvh(self.h, arr, len(objs), exposure[0], exposure[1]))

    def flatten(self, world):
        """Host-only (product library): the flat program as uint32 [n, 8] plus the feature mask."""
        arr = (C.c_uint32 * max(1, len(world)))(*world)
        n, feat = C.c_uint32(), C.c_uint32()
        self.be.check(self.be._debug_flatten(self.h, arr, len(world), C.byref(n), C.byref(feat), None, 0))
        words = np.zeros((n.value, 8), dtype=np.uint32)
        self.be.check(self.be._debug_flatten(self.h, arr, len(world), C.byref(n), C.byref(feat),
                                             words.ctypes.data_as(c_u32p), n.value))
        return words, feat.value

    def flatten_pool2(self, world):
        """The second flat program (pool-2 kernel) and its item table: (words [n, 8], items [(kind, a, b, c)], n_media, n_wrapped);
        words has 0 rows when the world has another shape."""
        arr = (C.c_uint32 * max(1, len(world)))(*world)
        n = C.c_uint32()
        table = np.zeros(24, dtype=np.uint32)
        self.be.check(self.be._debug_flatten_pool2(self.h, arr, len(world), C.byref(n), table.ctypes.data_as(c_u32p), None, 0))
        words = np.zeros((n.value, 8), dtype=np.uint32)
        self.be.check(self.be._debug_flatten_pool2(self.h, arr, len(world), C.byref(n), table.ctypes.data_as(c_u32p),
                                                   words.ctypes.data_as(c_u32p), n.value))
        items = [tuple(int(v) for v in table[4 * k:4 * k + 4]) for k in range(int(table[20]))]
        return words, items, int(table[21]), int(table[22])

    def bvh_sah(self, objs, exposure=(0.0, 1.0)):
        """Not in the reference: SAH-built Bvh; same results up to exact-t ties, fewer box tests."""
        arr = (C.c_uint32 * max(1, len(objs)))(*objs)
        return self.be.check_id(self.be._object_bvh_sah(self.h, arr, len(objs), exposure[0], exposure[1]))

    def scene(self, world, device=0):
        """Flatten `world` (the `[Box<dyn Object>]` of lib.rs:33) once into device memory."""
        arr = (C.c_uint32 * max(1, len(world)))(*world)
        h = C.c_void_p()
        self.be.check(self.be._scene_create(self.h, arr, len(world), device, C.byref(h)))
        sc = self.be.scene_class(self.be, h, self)
        sc.apply_env_options()
        return sc


class Scene:
    def __init__(self, backend, handle, builder):
        self.be = backend
        self.h = handle
        self._builder = builder  # keep alive

    def close(self):
        if self.h:
            self.be._scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # measurement / test hook: RTG_<OPTION>=<int> in the environment of the PYTHON process becomes
    # rtg_scene_set_option(scene, "<option>", <int>) -- the library itself reads no environment variable
    ENV_OPTIONS = ("kernel", "chunks", "lpt", "lpt_phase1", "lpt_deep", "lpt_shift", "ray_lds", "sync", "block", "wg_per_cu", "window",
                   "box_leave", "refill_min", "gather_min", "run_ahead", "run_ahead_min", "sphere_min", "verbose", "bvh4", "force_rccl", "multi_gather", "scratch_mb", "frames_in_flight", "small_frames", "drain_share", "hoist", "deep_sized", "mat_lds", "pool2", "p2_refill", "p2_box_leave", "p2_park", "p2_sphere", "p2_prism", "p2_list", "p2_push")

    def set_option(self, name, value):
        self.be.check(self.be._scene_set_option(self.h, name.encode(), int(value)))

    def apply_env_options(self):
        if not hasattr(self.be, "_scene_set_option"):
            return
        for name in self.ENV_OPTIONS:
            v = os.environ.get("RTG_" + name.upper())
            if v is not None:
                self.set_option(name, int(v))

    def info(self):
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.be.check(self.be._scene_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"instructions": a.value, "materials": b.value, "textures": c.value, "hbm_bytes": d.value}

    def _par_cast_args(self, args, threads):
        return args

    def par_cast(self, camera, nx, ny, ns, seed=0xDEADBEEF, stats=False, out=None, threads=0, **kw):
        """par_cast, lib.rs:363.  Returns float32 [ny, nx, 3], row 0 = top, linear radiance."""
        p = make_params(nx, ny, ns, seed=seed, flags=FLAG_COUNTERS if stats else 0, **kw)
        if out is None:
            out = np.zeros((ny, nx, 3), dtype=np.float32)
        st = Stats()
        st.struct_size = C.sizeof(Stats)
        args = self._par_cast_args([self.h, C.byref(camera), C.byref(p), out.ctypes.data_as(c_f32p), C.byref(st)],
                                   threads)
        self.be.check(self.be._par_cast(*args))
        return (out, st.as_dict()) if stats else out

    def par_cast_device(self, camera, params, d_out_ptr, stream=None, want_stats=False):
        st = Stats()
        st.struct_size = C.sizeof(Stats)
        self.be.check(self.be._par_cast_device(self.h, C.byref(camera), C.byref(params), d_out_ptr, stream,
                                               C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def debug_hit_top(self, rays, seed=1, t_near=0.001):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        out = np.zeros((n, 8), dtype=np.float32)
        mat = np.zeros(n, dtype=np.uint32)
        self.be.check(self.be._debug_hit_top(self.h, n, rays.ctypes.data_as(c_f32p), seed, t_near,
                                             out.ctypes.data_as(c_f32p), mat.ctypes.data_as(c_u32p)))
        return out, mat

    def debug_samples(self, camera, nx, ny, ns, xs, ys, samples, seed=0xDEADBEEF, trace_kernel=False, **kw):
        """trace_kernel=True: read the keys out of the production ray-pool kernel's per-sample trace (whole frame rendered)."""
        p = make_params(nx, ny, ns, seed=seed, flags=FLAG_TRACE_KERNEL if trace_kernel else 0, **kw)
        xs, ys, samples = (np.ascontiguousarray(a, dtype=np.uint32) for a in (xs, ys, samples))
        n = xs.size
        rgb = np.zeros((n, 3), dtype=np.float32)
        info = np.zeros((n, 4), dtype=np.uint32)
        self.be.check(self.be._debug_samples(self.h, C.byref(camera), C.byref(p), n,
                                             xs.ctypes.data_as(c_u32p), ys.ctypes.data_as(c_u32p),
                                             samples.ctypes.data_as(c_u32p), rgb.ctypes.data_as(c_f32p),
                                             info.ctypes.data_as(c_u32p)))
        return rgb, info


Backend.scene_class = Scene
