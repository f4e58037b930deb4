"""Scene builders of the reference, transliterated against the builder API (capi.Builder).

Each function takes a builder `b` (HIP product or, in tests, the oracle -- same calls) and returns
`(world, camera, exposure)` exactly like the reference's scene functions:

  cornell_box / cornell_box_with_boxes      src/lib.rs:103-193
  cornell_box_scene / motion_test / volume_test / simple_light_scene / book_final_scene
                                            src/main.rs:11-319
  bench_scene                               benches/scene.rs:8-36
  random_scene (book 1)                     src/lib.rs:238-319 (dead code in the reference, written
                                            against an older enum API) re-expressed in the live API
                                            as SURVEY.md 8(d) specifies.

All scalar arithmetic is done in numpy float32 so the scene data equals what the f32 Rust code builds.
"""
import numpy as np

from .small_rng import SmallRng, perlin_tables

f32 = np.float32
X, Y, Z = 0, 1, 2  # object::StaticX / StaticY / StaticZ


def v(x, y, z):
    return np.array([x, y, z], dtype=f32)


def vfrom(x):
    return np.array([x, x, x], dtype=f32)


# ---------------------------------------------------------------------------------------------
# src/lib.rs
# ---------------------------------------------------------------------------------------------
def cornell_box(b):
    """lib.rs:103-166"""
    def diffuse_color(c):
        return b.lambertian(b.constant(c))

    red = diffuse_color(v(0.65, 0.05, 0.05))
    white = diffuse_color(vfrom(0.73))
    green = diffuse_color(v(0.12, 0.45, 0.15))
    light = b.diffuse_light(b.constant(vfrom(1.0)), 15.0)
    return [
        b.rect(Y, (213.0, 343.0), (227.0, 332.0), 554.0, light),
        b.rect(Y, (0.0, 555.0), (0.0, 555.0), 0.0, white),                      # floor
        b.flip_normals(b.rect(Z, (0.0, 555.0), (0.0, 555.0), 555.0, white)),    # rear wall
        b.flip_normals(b.rect(Y, (0.0, 555.0), (0.0, 555.0), 555.0, white)),    # ceiling
        b.rect(X, (0.0, 555.0), (0.0, 555.0), 0.0, red),                        # right wall
        b.flip_normals(b.rect(X, (0.0, 555.0), (0.0, 555.0), 555.0, green)),    # left wall
    ]


def cornell_box_with_boxes(b):
    """lib.rs:168-193"""
    scene = cornell_box(b)
    white = b.lambertian(b.constant(vfrom(0.73)))
    scene.append(b.translate(v(130.0, 0.0, 65.0),
                             b.rotate_y(-18.0, b.rect_prism(v(0, 0, 0), v(165.0, 165.0, 165.0), white))))
    scene.append(b.translate(v(265.0, 0.0, 295.0),
                             b.rotate_y(15.0, b.rect_prism(v(0, 0, 0), v(165.0, 330.0, 165.0), white))))
    return scene


# ---------------------------------------------------------------------------------------------
# src/main.rs
# ---------------------------------------------------------------------------------------------
def _cornell_camera(be, nx, ny):
    """main.rs:12-27 (shared by cornell_box_scene, motion_test, volume_test, simple_light_scene)"""
    exposure = (0.0, 1.0)
    cam = be.camera_look(v(278.0, 278.0, -800.0), v(278.0, 278.0, 0.0), v(0.0, 1.0, 0.0), 40.0,
                         float(f32(nx) / f32(ny)), 0.0, 10.0, exposure)
    return cam, exposure


def cornell_box_scene(b, nx, ny):
    """main.rs:11-30"""
    cam, exposure = _cornell_camera(b.be, nx, ny)
    return cornell_box_with_boxes(b), cam, exposure


def motion_test(b, nx, ny):
    """main.rs:33-67"""
    cam, exposure = _cornell_camera(b.be, nx, ny)
    scene = cornell_box(b)
    mat = b.lambertian(b.constant(vfrom(0.73)))
    scene.append(b.translate(v(278.0, 278.0, 278.0),
                             b.linear_move(b.sphere(65.0, mat), v(0.0, 100.0, 0.0))))
    return scene, cam, exposure


def volume_test(b, nx, ny):
    """main.rs:70-108"""
    cam, exposure = _cornell_camera(b.be, nx, ny)
    scene = cornell_box(b)
    boundary = b.sphere(180.0, b.lambertian(b.constant(vfrom(0.73))))  # material does not matter
    medium = b.constant_medium(boundary, 0.01, b.isotropic(b.constant(v(0.2, 0.2, 1.0))))
    scene.append(b.translate(v(278.0, 278.0, 278.0), medium))
    return scene, cam, exposure


def simple_light_scene(b, nx, ny, rng, spheres=1000):
    """main.rs:111-159"""
    cam, exposure = _cornell_camera(b.be, nx, ny)
    world = cornell_box(b)
    for _ in range(spheres):
        mat = b.lambertian(b.constant(vfrom(0.3)))
        offset = f32(277.0) + f32(257.0) * rng.gen_vec3()
        world.append(b.translate(offset, b.sphere(20.0, mat)))
    world.append(b.flip_normals(b.sphere(1000.0, b.diffuse_light(b.constant(vfrom(0.1)), 1.0))))
    return world, cam, exposure


def book_final_scene(b, nx, ny, rng, perlin_seed=0xDEADBEEF):
    """main.rs:161-319 (book 2 final scene).  Perlin tables: fixed, from SmallRng(perlin_seed)."""
    exposure = (0.0, 1.0)
    cam = b.be.camera_look(v(478.0, 278.0, -600.0), v(278.0, 278.0, 0.0), v(0.0, 1.0, 0.0), 40.0,
                           float(f32(nx) / f32(ny)), 0.0, 10.0, exposure)
    b.set_perlin_tables(*perlin_tables(perlin_seed))

    ground = b.lambertian(b.constant(v(0.48, 0.83, 0.53)))
    world = []

    # Make random floor. main.rs:192-203
    boxes = []
    W = f32(100.0)
    for i in range(20):
        for j in range(20):
            c0 = v(f32(-1000.0) + f32(i) * W, 0.0, f32(-1000.0) + f32(j) * W)
            c1 = c0 + v(W, f32(100.0) * (rng.gen_f32() + f32(0.01)), W)
            boxes.append(b.rect_prism(c0, c1, ground))
    world.append(b.bvh(boxes, exposure))

    # Make light. main.rs:206-215
    world.append(b.rect(Y, (123.0, 423.0), (147.0, 412.0), 554.0,
                        b.diffuse_light(b.constant(vfrom(1.0)), 7.0)))

    # Brown blurry sphere. main.rs:218-229
    world.append(b.translate(v(400.0, 400.0, 200.0),
                             b.linear_move(b.sphere(50.0, b.lambertian(b.constant(v(0.7, 0.3, 0.1)))),
                                           v(30.0, 0.0, 0.0))))
    glass = b.dielectric(1.5)
    # Glass sphere. main.rs:234-240
    world.append(b.translate(v(260.0, 150.0, 45.0), b.sphere(50.0, glass)))
    # Silvery sphere. main.rs:243-252
    world.append(b.translate(v(0.0, 150.0, 145.0), b.sphere(50.0, b.metal(v(0.8, 0.8, 0.9), 1.0))))
    # Blue glass sphere. main.rs:255-269
    boundary = b.translate(v(360.0, 150.0, 145.0), b.sphere(70.0, glass))
    world.append(boundary)
    world.append(b.constant_medium(boundary, 0.2, b.isotropic(b.constant(v(0.2, 0.4, 0.9)))))
    # Fog. main.rs:272-281
    world.append(b.constant_medium(b.sphere(5000.0, glass), 0.0001, b.isotropic(b.constant(vfrom(1.0)))))
    # Perlin marbled sphere. main.rs:284-292
    world.append(b.translate(v(220.0, 280.0, 300.0), b.sphere(80.0, b.lambertian(b.perlin(0.05)))))
    # Cube made of random spheres. main.rs:295-316
    white = b.lambertian(b.constant(vfrom(0.73)))
    spheres = [b.translate(f32(165.0) * rng.gen_vec3(), b.sphere(10.0, white)) for _ in range(1000)]
    world.append(b.translate(v(-100.0, 270.0, 395.0), b.rotate_y(15.0, b.bvh(spheres, exposure))))
    return world, cam, exposure


def bench_scene(b, nx, ny):
    """benches/scene.rs:8-36: Cornell box + prisms in a BVH, book-1 camera."""
    world = [b.bvh(cornell_box_with_boxes(b), (0.0, 1.0))]
    cam = b.be.camera_look(v(13.0, 2.0, 3.0), v(0, 0, 0), v(0.0, 1.0, 0.0), 20.0, float(f32(nx) / f32(ny)),
                           0.1, 10.0, (0.0, 1.0))
    return world, cam, (0.0, 1.0)


# ---------------------------------------------------------------------------------------------
# Book-1 random spheres (BASELINE.json north-star workload)
# ---------------------------------------------------------------------------------------------
def random_scene_objects(b, rng):
    """lib.rs:238-319 re-expressed in the live API: each sphere is
    Translate{offset: center, Sphere{radius, material}}; no motion (book 1); constant grey ground;
    the three r=1 spheres follow the book layout (glass / Lambertian / metal); a FlipNormals sky-dome
    emitter (same idiom as main.rs:150-156) replaces the sky, because a miss is black (lib.rs:100)."""
    world = [b.translate(v(0.0, -1000.0, 0.0), b.sphere(1000.0, b.lambertian(b.constant(vfrom(0.5)))))]
    for a in range(-11, 11):
        for bb in range(-11, 11):
            center = v(f32(a) + f32(0.9) * rng.gen_f32(), 0.2, f32(bb) + f32(0.9) * rng.gen_f32())
            d = center - v(4.0, 0.2, 0.0)
            dist = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
            if dist > f32(0.9):
                choose_mat = rng.gen_f32()
                if choose_mat < f32(0.8):
                    mat = b.lambertian(b.constant(rng.gen_vec3() * rng.gen_vec3()))
                elif choose_mat < f32(0.95):
                    albedo = f32(0.5) * (f32(1.0) + rng.gen_vec3())
                    mat = b.metal(albedo, float(f32(0.5) * rng.gen_f32()))
                else:
                    mat = b.dielectric(1.5)
                world.append(b.translate(center, b.sphere(0.2, mat)))
    world.append(b.translate(v(0.0, 1.0, 0.0), b.sphere(1.0, b.dielectric(1.5))))
    world.append(b.translate(v(-4.0, 1.0, 0.0), b.sphere(1.0, b.lambertian(b.constant(v(0.4, 0.2, 0.1))))))
    world.append(b.translate(v(4.0, 1.0, 0.0), b.sphere(1.0, b.metal(v(0.7, 0.6, 0.5), 0.0))))
    world.append(b.flip_normals(b.sphere(10000.0, b.diffuse_light(b.constant(v(0.7, 0.8, 1.0)), 1.0))))
    return world


def random_scene(b, nx, ny, rng=None, use_bvh=True):
    """Book-1 random spheres + the benches/scene.rs:16-30 camera; whole scene under bvh::from_scene
    (the USE_BVH = true path of main.rs:340-345)."""
    if rng is None:
        rng = SmallRng(0xDEADBEEF)  # main.rs:333
    exposure = (0.0, 1.0)
    objs = random_scene_objects(b, rng)
    if use_bvh == "sah":      # non-parity option: same image, ~36 % fewer Aabb tests per ray
        world = [b.bvh_sah(objs, exposure)]
    else:
        world = [b.bvh(objs, exposure)] if use_bvh else objs
    cam = b.be.camera_look(v(13.0, 2.0, 3.0), v(0, 0, 0), v(0.0, 1.0, 0.0), 20.0, float(f32(nx) / f32(ny)),
                           0.1, 10.0, exposure)
    return world, cam, exposure
