"""An anchor the REFERENCE itself publishes: img/rttnw-final.jpg, the book-2 final scene of its README (rendered by the Rust
binary at 5000 spp; the scene is built by `book_final_scene` with `SmallRng::seed_from_u64(0xDEADBEEF)`, src/main.rs:333, so the
heights of the 400 floor boxes and the positions of the 1000 small spheres are a function of rand 0.6.5's seed expansion, its PCG
stream and its float conversions).  tests/golden/ref_rttnw_final_200.npz is that JPEG box-filtered to 200x200
(tools/gen_ref_image_fixture.py).  No bit-exactness is possible against a JPEG of a 5000-spp render with thread_rng Perlin tables;
what IS decidable: our render of the same scene correlates with the reference's picture where the construction randomness shows
(floor, sphere cube) only if the emulated SmallRng stream is the real one -- any other construction seed drops the floor
correlation from 0.91 to ~0.5.  This pins small_rng.py (`seed_from_u64`, Pcg64Mcg, gen::<f32>(), gen_range) and the scene
transliteration against an output of the reference, which the recorded-but-unverified KATs could not.
(Levels are NOT compared: bright regions agree within 1-4 of 255, but the JPEG's dark regions sit ~10 levels above ours -- the
README does not say which revision, light or tone curve produced it -- so the test uses correlation, which an offset cannot fake.)"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N = 200


def corr(a, b):
    return float(np.corrcoef(a.ravel(), b.ravel())[0, 1])


def render_u8(pkg, backend, seed, ns):
    b = backend.builder()
    world, cam, _ = pkg.scenes.book_final_scene(b, N, N, pkg.small_rng.SmallRng(seed))
    return pkg.ppm.to_u8(b.scene(world).par_cast(cam, N, N, ns)).astype(np.float32)   # print_ppm's quantisation, lib.rs:348-356


def check(pkg, backend, ns):
    ref = np.load(os.path.join(GOLD, "ref_rttnw_final_200.npz"))["rgb"].astype(np.float32)
    floor = slice(N * 2 // 3, N)                       # the 400 boxes of random height
    cube = (slice(N // 3, N * 2 // 3), slice(N // 2, N))   # the rotated cube of 1000 random spheres
    ours = render_u8(pkg, backend, 0xDEADBEEF, ns)
    assert corr(ours, ref) > 0.90
    assert corr(ours[floor], ref[floor]) > 0.87 and corr(ours[cube], ref[cube]) > 0.85
    for other in (0xDEADBEEE, 1):                       # a neighbouring and an unrelated construction seed: the picture is another one
        ctl = render_u8(pkg, backend, other, ns)
        assert corr(ctl[floor], ref[floor]) < 0.65 and corr(ctl[cube], ref[cube]) < 0.78


def test_oracle_render_matches_the_reference_readme_image(pkg, oracle):
    check(pkg, oracle, 150)


@pytest.mark.gpu
def test_gpu_render_matches_the_reference_readme_image(pkg, gpu):
    check(pkg, gpu, 1000)


# ---- the reference's OTHER published picture: img/demo-scene.jpg = the book-1 random-spheres scene, i.e. the scene of the
# headline workload (BASELINE.json configs[1] / [2]).  tests/golden/ref_demo_scene_300x200.npz is that JPEG box-filtered to
# 300x200 (tools/gen_ref_image_fixture.py).  It was rendered by an older revision (gradient sky instead of our sky-dome emitter),
# so levels are not compared either -- but the ~480 small spheres on the ground sit where, and have the materials and colours
# that, `random_scene` (src/lib.rs:236-319) draws from SmallRng::seed_from_u64(0xDEADBEEF): our transliteration with the emulated
# stream correlates with the reference's picture at 0.99 (0.98 on the ground alone); any other construction seed gives 0.6 overall
# (camera, big spheres, horizon) and 0.0-0.25 on the ground.  That pins, against an OUTPUT OF THE REFERENCE, the seed expansion,
# Pcg64Mcg, gen::<f32>() / gen::<Vec3>(), the draw ORDER of the scene builder and the camera of the workload bench.py times.
NX2, NY2 = 300, 200


def render_book1_u8(pkg, backend, seed, ns):
    b = backend.builder()
    world, cam, _ = pkg.scenes.random_scene(b, NX2, NY2, rng=pkg.small_rng.SmallRng(seed))
    return pkg.ppm.to_u8(b.scene(world).par_cast(cam, NX2, NY2, ns)).astype(np.float32)


def check_book1(pkg, backend, ns):
    ref = np.load(os.path.join(GOLD, "ref_demo_scene_300x200.npz"))["rgb"].astype(np.float32)
    ground = slice(NY2 * 55 // 100, NY2)                # below the horizon: the field of small random spheres
    ours = render_book1_u8(pkg, backend, 0xDEADBEEF, ns)
    assert corr(ours, ref) > 0.97 and corr(ours[ground], ref[ground]) > 0.95
    for other in (0xDEADBEEE, 1):
        ctl = render_book1_u8(pkg, backend, other, ns)
        assert corr(ctl, ref) < 0.75 and corr(ctl[ground], ref[ground]) < 0.45


def test_oracle_book1_render_matches_the_reference_demo_image(pkg, oracle):
    check_book1(pkg, oracle, 16)


@pytest.mark.gpu
def test_gpu_book1_render_matches_the_reference_demo_image(pkg, gpu):
    check_book1(pkg, gpu, 200)
