"""Scene cases shared by the oracle tests, the GPU parity tests and the golden-fixture generator."""
import numpy as np


def _volume_in_bvh(pkg, b, nx, ny):
    """volume_test's world under bvh::from_scene (the USE_BVH = true path, main.rs:340-345):
    a ConstantMedium as a Bvh leaf exercises the hl.t < hr.t merge rule (bvh.rs:104-112)."""
    world, cam, exp = pkg.scenes.volume_test(b, nx, ny)
    return [b.bvh(world, exp)], cam, exp


def _book2_bvh(pkg, b, nx, ny):
    world, cam, exp = pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF))
    return [b.bvh(world, exp)], cam, exp


def _checker_scale(pkg, b, nx, ny):
    """Surface not used by any live reference scene: checker texture (texture.rs:12) and Scale
    (object.rs:296) -- SURVEY.md 8(f) item 4."""
    S = pkg.scenes
    world = S.cornell_box(b)
    chk = b.lambertian(b.checker(b.constant(S.v(0.2, 0.3, 0.1)), b.constant(S.vfrom(0.9))))
    world.append(b.translate(S.v(278.0, 120.0, 278.0), b.scale(S.v(1.0, 0.5, 1.5), b.sphere(120.0, chk))))
    world.append(b.translate(S.v(150.0, 300.0, 200.0),
                             b.flip_normals(b.flip_normals(b.sphere(60.0, b.metal(S.v(0.8, 0.85, 0.88), 0.3))))))
    cam, exp = S._cornell_camera(b.be, nx, ny)
    return world, cam, exp


def _big_lean(pkg, b, nx, ny):
    """3000 spheres under one Bvh: a lean scene whose flat program (8999 records = 288 KB) does NOT fit
    LDS, so the ray-pool kernel runs its global-memory-program variant."""
    S = pkg.scenes
    rng = pkg.small_rng.SmallRng(42)
    objs = [b.translate(S.v(0.0, -1000.0, 0.0), b.sphere(1000.0, b.lambertian(b.constant(S.vfrom(0.5)))))]
    mats = [b.lambertian(b.constant(S.v(0.7, 0.3, 0.2))), b.metal(S.v(0.8, 0.8, 0.9), 0.1), b.dielectric(1.5)]
    for i in range(2998):
        c = S.f32(24.0) * rng.gen_vec3() - S.f32(12.0)
        objs.append(b.translate(S.v(c[0], S.f32(0.1) + S.f32(3.0) * rng.gen_f32(), c[2]), b.sphere(0.1, mats[i % 3])))
    objs.append(b.flip_normals(b.sphere(10000.0, b.diffuse_light(b.constant(S.v(0.7, 0.8, 1.0)), 1.0))))
    cam = b.be.camera_look(S.v(13, 2, 3), S.v(0, 0, 0), S.v(0.0, 1.0, 0.0), 20.0, float(S.f32(nx) / S.f32(ny)), 0.1, 10.0)
    return [b.bvh(objs, (0.0, 1.0))], cam, (0.0, 1.0)


def _cornell_smoke(pkg, b, nx, ny):
    """The Cornell box with its two prisms turned into smoke: ConstantMedium<O> (object.rs:441) over
    Translate<RotateY<And<...rects>>> boundaries -- the general (object-graph) boundary walk."""
    S = pkg.scenes
    world = S.cornell_box(b)
    white = b.lambertian(b.constant(S.vfrom(0.73)))
    box1 = b.translate(S.v(130.0, 0.0, 65.0), b.rotate_y(-18.0, b.rect_prism(S.v(0, 0, 0), S.v(165.0, 165.0, 165.0), white)))
    box2 = b.translate(S.v(265.0, 0.0, 295.0), b.rotate_y(15.0, b.rect_prism(S.v(0, 0, 0), S.v(165.0, 330.0, 165.0), white)))
    world.append(b.constant_medium(box1, 0.01, b.isotropic(b.constant(S.vfrom(1.0)))))
    world.append(b.constant_medium(box2, 0.01, b.isotropic(b.constant(S.vfrom(0.0)))))
    cam, exp = S._cornell_camera(b.be, nx, ny)
    return world, cam, exp


CASES = {
    # name: (builder fn (pkg, b, nx, ny) -> (world, cam, exposure), nx, ny, ns)
    "cornell": (lambda pkg, b, nx, ny: pkg.scenes.cornell_box_scene(b, nx, ny), 32, 32, 16),
    "book1": (lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny), 48, 32, 8),
    "book1_list": (lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny, use_bvh=False), 24, 16, 4),
    "book2": (lambda pkg, b, nx, ny: pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF)),
              32, 32, 8),
    "book2_bvh": (_book2_bvh, 32, 32, 8),
    "bench": (lambda pkg, b, nx, ny: pkg.scenes.bench_scene(b, nx, ny), 10, 10, 4),
    "motion": (lambda pkg, b, nx, ny: pkg.scenes.motion_test(b, nx, ny), 32, 32, 8),
    "volume": (lambda pkg, b, nx, ny: pkg.scenes.volume_test(b, nx, ny), 32, 32, 8),
    "volume_bvh": (_volume_in_bvh, 32, 32, 8),
    "simple_light": (lambda pkg, b, nx, ny: pkg.scenes.simple_light_scene(
        b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF), spheres=200), 24, 24, 4),
    # simple_light_scene at the reference's own size: 1000 spheres in a LIST world (main.rs:137, USE_BVH = false)
    "simple_light_1000": (lambda pkg, b, nx, ny: pkg.scenes.simple_light_scene(
        b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF), spheres=1000), 24, 24, 4),
    "checker_scale": (_checker_scale, 32, 32, 8),
    "big_lean": (_big_lean, 40, 24, 6),
    "cornell_smoke": (_cornell_smoke, 32, 32, 8),
    "book1_sah": (lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny, use_bvh="sah"), 48, 32, 8),
}


def build_case(pkg, backend, name, nx=None, ny=None):
    fn, dnx, dny, dns = CASES[name]
    nx, ny = nx or dnx, ny or dny
    b = backend.builder()
    world, cam, _ = fn(pkg, b, nx, ny)
    return b.scene(world), cam, nx, ny, dns
