#!/usr/bin/env python3
"""What the parts of book-2's object list cost the full-feature pool kernel (GPU box): the scene of main.rs:161-319 with parts of
its top-level list left out, timed per ray.  Different scenes render different images -- this is a cost probe for the
schedule model, not a parity run.  usage: probe_book2_levers.py [nx ny ns]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
gpu = pkg.load()
S = pkg.scenes
nx, ny, ns = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (800, 800, 100)

# indices into book_final_scene's world list: 0 floor Bvh, 1 light rect, 2 moving sphere, 3 glass, 4 metal, 5 blue glass, 6 blue
# medium, 7 fog, 8 perlin sphere, 9 cube of spheres
VARIANTS = [
    ("all ten objects", range(10)),
    ("without the four plain spheres 2-5", [0, 1, 6, 7, 8, 9]),
    ("without spheres 2-5 and the perlin sphere", [0, 1, 6, 7, 9]),
    ("without the two media", [0, 1, 2, 3, 4, 5, 8, 9]),
    ("without the fog", [0, 1, 2, 3, 4, 5, 6, 8, 9]),
    ("floor + light + cube", [0, 1, 9]),
    ("floor + light", [0, 1]),
    ("light + cube", [1, 9]),
    ("light + the seven list-level objects", [1, 2, 3, 4, 5, 6, 7, 8]),
]
print("%-46s %9s %9s %9s %10s %8s %8s" % ("book-2 %dx%dx%d" % (nx, ny, ns), "ms", "Mrays", "ns/ray", "rays/smp", "box/ray", "prim/ray"))
# the cube of spheres (main.rs:295-316) under other wrappers: what Translate{RotateY{Bvh}} costs beside the Bvh itself
CUBE = [("floor + light + cube, cube = bare Bvh", "bare"), ("floor + light + cube, cube = Translate{Bvh}", "translate"),
        ("floor + light + cube, cube = RotateY{Bvh}", "rotate")]
for name, keep in VARIANTS + CUBE:
    b = gpu.builder()
    rng = pkg.small_rng.SmallRng(0xDEADBEEF)
    world, cam, _ = S.book_final_scene(b, nx, ny, rng)
    if isinstance(keep, str):
        rng = pkg.small_rng.SmallRng(0xDEADBEEF)
        for _ in range(400):
            rng.gen_f32()
        white = b.lambertian(b.constant(S.vfrom(0.73)))
        bvh = b.bvh([b.translate(S.f32(165.0) * rng.gen_vec3(), b.sphere(10.0, white)) for _ in range(1000)], (0.0, 1.0))
        cube = bvh if keep == "bare" else b.translate(S.v(-100.0, 270.0, 395.0), bvh) if keep == "translate" else b.rotate_y(15.0, bvh)
        sc = b.scene([world[0], world[1], cube])
    else:
        sc = b.scene([world[i] for i in keep])
    sc.par_cast(cam, nx, ny, 1)
    _, st = sc.par_cast(cam, nx, ny, ns, stats=True)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        sc.par_cast(cam, nx, ny, ns)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print("%-46s %9.2f %9.1f %9.3f %10.2f %8.1f %8.1f" % (name, dt * 1e3, st["rays"] / 1e6, dt * 1e9 / st["rays"],
                                                       st["rays"] / st["samples"], st["aabb_tests"] / st["rays"],
                                                       st["prim_tests"] / st["rays"]), flush=True)
