"""CPU-only tests of the oracle (the restatement of the reference's hot path): golden fixtures,
self-consistency properties and the edge cases SURVEY.md H5/H8 lists."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import assert_bit_equal
from scene_cases import CASES, build_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_framebuffer(pkg, oracle, name):
    gold = np.load(os.path.join(GOLD, "framebuffers.npz"))[name]
    scene, cam, nx, ny, ns = build_case(pkg, oracle, name)
    assert_bit_equal(scene.par_cast(cam, nx, ny, ns), gold, name)
    # thread count must not change a single bit (rows are independent, RNG keyed by pixel/sample)
    assert_bit_equal(scene.par_cast(cam, nx, ny, ns, threads=1), gold, name + " 1 thread")


@pytest.mark.parametrize("name", ["cornell", "book1", "book2", "volume_bvh"])
def test_golden_sample_traces(pkg, oracle, name):
    g = np.load(os.path.join(GOLD, "samples.npz"))
    xs, ys, ss = g[name + ".keys"]
    scene, cam, nx, ny, ns = build_case(pkg, oracle, name)
    rgb, info = scene.debug_samples(cam, nx, ny, ns, xs, ys, ss)
    assert_bit_equal(rgb, g[name + ".rgb"], name)
    assert np.array_equal(info, g[name + ".info"])


def test_pixel_is_ordered_fold_of_its_samples(pkg, oracle):
    """par_cast (lib.rs:365-374): pixel = ((0 + c0) + c1 + ...) / ns, in sample order."""
    scene, cam, nx, ny, ns = build_case(pkg, oracle, "book1")
    img = scene.par_cast(cam, nx, ny, ns)
    for (x, row) in [(0, 0), (17, 5), (nx - 1, ny - 1), (20, 30)]:
        y = ny - 1 - row  # row 0 is the TOP scanline = y = ny-1 (lib.rs:328)
        rgb, _ = scene.debug_samples(cam, nx, ny, ns, [x] * ns, [y] * ns, list(range(ns)))
        acc = np.zeros(3, dtype=np.float32)
        for s in range(ns):
            acc = acc + rgb[s]
        assert_bit_equal(acc / np.float32(ns), img[row, x], "pixel (%d,%d)" % (x, row))


def test_bvh_world_equals_list_world(pkg, oracle):
    """Closest-hit results do not depend on the tree (only exact-t ties and medium RNG order could):
    the book-1 scene gives the same image as a flat list and under bvh::from_scene."""
    sb, cam, nx, ny, ns = build_case(pkg, oracle, "book1", 24, 16)
    sl, cam2, _, _, _ = build_case(pkg, oracle, "book1_list", 24, 16)
    assert_bit_equal(sb.par_cast(cam, nx, ny, 4), sl.par_cast(cam2, nx, ny, 4), "bvh vs list")


def test_tile_shards_sum_to_full_frame(pkg, oracle):
    scene, cam, nx, ny, ns = build_case(pkg, oracle, "cornell", 48, 48)
    full = scene.par_cast(cam, nx, ny, 4)
    for nranks in (2, 3, 8):
        acc = np.zeros_like(full)
        for r in range(nranks):
            part = scene.par_cast(cam, nx, ny, 4, rank=r, nranks=nranks, tile_w=16, tile_h=16)
            assert ((part != 0).any(axis=2) & (acc != 0).any(axis=2)).sum() == 0  # disjoint
            acc = acc + part
        assert_bit_equal(acc, full, "%d shards" % nranks)


def test_cast_sequential_small_rng_fixture(pkg, oracle):
    """cast() (lib.rs:378-397) with SmallRng 0xDEADBEEF on the Criterion bench config
    (benches/scene.rs:8-36): the reference's only deterministic path, restated; fixture is self-minted."""
    gold = np.load(os.path.join(GOLD, "cast_bench_10x10x4.npz"))["image"]
    scene, cam, _, _, _ = build_case(pkg, oracle, "bench")
    assert_bit_equal(scene.cast(cam, 10, 10, 4, 0xDEADBEEF), gold, "cast")


# ---- per-function known answers ---------------------------------------------------------------------
def aabb(oracle, mn, mx, o, d, t0, t1):
    f3 = lambda v: (C.c_float * 3)(*v)
    return oracle.lib.rto_debug_aabb_hit(f3(mn), f3(mx), f3(o), f3(d), C.c_float(t0), C.c_float(t1))


def test_aabb_hit_edge_cases(oracle):
    """aabb.rs:16-27"""
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, -5), (0, 0, 1), 0.001, 3e38) == 1
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, -5), (0, 0, -1), 0.001, 3e38) == 0   # behind
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, -5), (0, 0, 1), 0.001, 3.9) == 0     # t_max before entry
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, -5), (0, 0, 1), 0.001, 4.5) == 1
    # zero direction components: inv = +inf; inside the slab -> (-inf, +inf), outside -> rejected
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0.5, 0.5, -5), (0, 0, 1), 0.001, 3e38) == 1
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (2.0, 0.5, -5), (0, 0, 1), 0.001, 3e38) == 0
    # origin exactly on a slab plane with d = 0: (min - o) * inf = 0 * inf = NaN, ignored by f32::max/min
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (1.0, 0.0, -5), (0, 0, 1), 0.001, 3e38) == 1
    # negative direction swaps t0/t1
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (5, 0, 0), (-1, 0, 0), 0.001, 3e38) == 1
    # degenerate (flat) box: end == start -> `end > start` is false
    assert aabb(oracle, (-1, -1, 0), (1, 1, 0), (0, 0, -5), (0, 0, 1), 0.001, 3e38) == 0


def test_bounding_boxes(pkg, oracle):
    """object.rs:113,220,285,372,514"""
    S = pkg.scenes
    b = oracle.builder()
    m = b.lambertian(b.constant(S.vfrom(0.5)))

    def bb(o, e=(0.0, 1.0)):
        out = (C.c_float * 6)()
        oracle.check(oracle.lib.rto_debug_bounding_box(b.h, C.c_uint32(o), C.c_float(e[0]), C.c_float(e[1]), out))
        return np.array(out, dtype=np.float32)

    s = b.sphere(2.0, m)
    assert bb(s).tolist() == [-2, -2, -2, 2, 2, 2]
    assert bb(b.translate(S.v(1, 2, 3), s)).tolist() == [-1, 0, 1, 3, 4, 5]
    r = b.rect(S.Y, (1.0, 2.0), (3.0, 4.0), 5.0, m)
    got = bb(r)
    assert got[0] == 1 and got[3] == 2 and got[2] == 3 and got[5] == 4
    assert got[1] == np.float32(5.0) - np.float32(0.0001) and got[4] == np.float32(5.0) + np.float32(0.0001)
    mv = bb(b.linear_move(s, S.v(0, 10, 0)), (0.0, 1.0))
    assert mv.tolist() == [-2, -2, -2, 2, 12, 2]
    rot = bb(b.rotate_y(90.0, b.translate(S.v(10, 0, 0), s)))
    # 90 degrees about Y: (x, z) -> (z*sin, -x*sin): the box moves from x~10 to z~-10
    assert abs(rot[2] + 12) < 1e-4 and abs(rot[5] + 8) < 1e-4
    sc = bb(b.scale(S.v(2, 1, 0.5), s))
    assert sc.tolist() == [-4, -2, -1, 4, 2, 1]


def test_hit_top_known_answers(pkg, oracle):
    """Sphere::hit (object.rs:84-111), Rect::hit (:185-218), FlipNormals, Translate on hand-computed rays."""
    S = pkg.scenes
    b = oracle.builder()
    m0 = b.lambertian(b.constant(S.vfrom(0.5)))
    m1 = b.metal(S.vfrom(0.5), 0.0)
    world = [b.translate(S.v(0, 0, 10), b.sphere(2.0, m0)),
             b.flip_normals(b.rect(S.Z, (-1.0, 1.0), (-1.0, 1.0), 20.0, m1))]
    scene = b.scene(world)
    rays = np.array([
        [0, 0, 0, 0, 0, 1, 0],        # hits the sphere front at t=8
        [0, 0, 10, 0, 0, 1, 0],       # from the centre: far root t=2, normal +z
        [0, 0, 0, 0, 0, 2, 0],        # un-normalised direction: t=4
        [2, 0, 0, 0, 0, 1, 0],        # exactly tangent: discriminant == 0 -> NOT > 0 -> falls through to the rect? x=2 outside
        [0.5, 0.5, 13, 0, 0, 1, 0],   # starts beyond the sphere: rect at t=7, flipped normal -z
        [1.0, 0.0, 13, 0, 0, 1, 0],   # x == range0.end is outside (half-open)
        [-1.0, 0.0, 13, 0, 0, 1, 0],  # x == range0.start is inside
        [0, 0, 0, 0, 0, -1, 0],       # away from everything
    ], dtype=np.float32)
    out, mat = scene.debug_hit_top(rays)
    assert out[0, 0] == 1 and out[0, 1] == 8 and out[0, 2:5].tolist() == [0, 0, 8] and out[0, 5:8].tolist() == [0, 0, -1]
    assert mat[0] == m0
    assert out[1, 1] == 2 and out[1, 5:8].tolist() == [0, 0, 1]
    assert out[2, 1] == 4
    assert out[3, 0] == 0
    assert out[4, 0] == 1 and out[4, 1] == 7 and mat[4] == m1 and out[4, 5:8].tolist() == [0, 0, -1]
    assert out[5, 0] == 0
    assert out[6, 0] == 1
    assert out[7, 0] == 0 and mat[7] == 0xffffffff


def test_nan_polarity_of_rect_and_sphere(pkg, oracle):
    """H5: Rect::hit rejects with `t < start || t >= end` (NaN passes); Sphere accepts with
    `t < end && t >= start` (NaN fails).  A ray lying in the rect's plane from a point on it: t = 0/0."""
    S = pkg.scenes
    b = oracle.builder()
    m = b.lambertian(b.constant(S.vfrom(0.5)))
    scene = b.scene([b.rect(S.Y, (0.0, 10.0), (0.0, 10.0), 0.0, m)])
    out, _ = scene.debug_hit_top(np.array([[5, 0, 5, 1, 0, 0, 0]], dtype=np.float32))
    assert out[0, 0] == 1 and np.isnan(out[0, 1]) and np.isnan(out[0, 2])  # the reference reports a NaN hit


def test_quirks_are_preserved(pkg, oracle):
    """H8: a miss returns black and discards accum (lib.rs:100); DiffuseLight does not scatter
    (material.rs:108); LinearMove does not move hit.p back (object.rs:504-511)."""
    S = pkg.scenes
    b = oracle.builder()
    cam = oracle.camera_look(S.v(0, 0, -10), S.v(0, 0, 0), S.v(0, 1, 0), 20.0, 1.0, 0.0, 10.0)
    light = b.diffuse_light(b.constant(S.vfrom(1.0)), 3.0)
    img = b.scene([b.sphere(1000.0, light)]).par_cast(cam, 4, 4, 2)   # inside an emitting sphere
    assert (img == 3.0).all()
    img = b.scene([b.translate(S.v(0, 0, 500), b.sphere(1.0, light))]).par_cast(cam, 4, 4, 2)  # nothing hit
    assert (img == 0.0).all()
    mv = b.scene([b.linear_move(b.sphere(1.0, light), S.v(0, 5, 0))])
    out, _ = mv.debug_hit_top(np.array([[0, 5, -10, 0, 0, 1, 1.0]], dtype=np.float32))  # time = 1
    assert out[0, 0] == 1 and out[0, 1] == 9 and out[0, 3] == 0.0   # p.y is the un-moved 0, not 5


def test_perlin_turb_reference_properties(pkg, oracle):
    """perlin.rs:49-75: turb >= 0, deterministic, noise vanishes on the integer lattice."""
    b = oracle.builder()
    b.set_perlin_tables(*pkg.small_rng.perlin_tables(0xDEADBEEF))
    f3 = lambda v: (C.c_float * 3)(*v)
    oracle.lib.rto_debug_perlin_turb.restype = C.c_float
    t = oracle.lib.rto_debug_perlin_turb
    assert t(b.h, f3((3.0, -2.0, 7.0)), 1) == 0.0
    vals = [t(b.h, f3((0.37 * i, 1.1 * i, -0.73 * i)), 7) for i in range(1, 50)]
    assert min(vals) >= 0.0 and max(vals) > 0.05 and max(vals) < 4.0
    assert vals == [t(b.h, f3((0.37 * i, 1.1 * i, -0.73 * i)), 7) for i in range(1, 50)]


def test_oracle_error_codes(pkg, oracle):
    b = oracle.builder()
    with pytest.raises(pkg.RtError) as e:
        b.bvh([])
    assert "zero objects" in str(e.value)   # bvh.rs:60
    with pytest.raises(pkg.RtError):
        b.perlin(1.0)                        # tables not set


def test_print_ppm_quantisation(pkg, oracle):
    """lib.rs:348-356: sqrt gamma, (255.99 * x) as i32 with Rust's saturating cast, clamp to 0..=255."""
    x = np.array([0.0, 1.0, 0.25, 4.0, -1.0, np.nan, np.inf, 1e-12, 0.999, 0.5], dtype=np.float32)
    q = oracle.tonemap(x)
    assert q.tolist() == [0, 255, 127, 255, 0, 0, 255, 0, 255, 181]
    rs = np.random.RandomState(1)
    img = rs.rand(16, 8, 3).astype(np.float32) * 1.3
    assert np.array_equal(oracle.tonemap(img), pkg.ppm.to_u8(img).astype(np.uint8))
    assert pkg.ppm.format_ppm(img).startswith("P3\n8 16\n255\n")


def test_sah_tree_gives_the_reference_image(pkg, oracle):
    """a SAH-built Bvh changes the tree, not the closest hits (only exact-t ties could
    differ): same image as Bvh::new's median tree, with fewer Aabb::hit calls."""
    ref, cam, nx, ny, ns = build_case(pkg, oracle, "book1", 64, 40)
    sah, cam2, _, _, _ = build_case(pkg, oracle, "book1_sah", 64, 40)
    a, sa = ref.par_cast(cam, nx, ny, 6, stats=True)
    b_, sb = sah.par_cast(cam2, nx, ny, 6, stats=True)
    assert_bit_equal(a, b_, "sah vs median tree")
    assert sb["rays"] == sa["rays"] and sb["aabb_tests"] < 0.75 * sa["aabb_tests"]


def test_par_cast_multi_mirror_equals_one_frame(pkg, oracle):
    """rto_par_cast_multi (the checker for rtg_par_cast_multi): n shard renders summed == the unsharded frame."""
    from scene_cases import build_case
    scenes = []
    for _ in range(3):
        s, cam, nx, ny, ns = build_case(pkg, oracle, "book1", 64, 48)
        scenes.append(s)
    whole, st1 = scenes[0].par_cast(cam, nx, ny, ns, stats=True)
    multi, stn = oracle.par_cast_multi(scenes, cam, nx, ny, ns, stats=True)
    assert_bit_equal(multi, whole, "3 shards")
    for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st1[k] == stn[k], k
    with pytest.raises(pkg.RtError):
        oracle.par_cast_multi(scenes, cam, nx, ny, ns, rank=1, nranks=2)   # the call shards by itself


def test_non_finite_exposure_is_refused(pkg, oracle):
    """rand 0.6.5 panics on non-finite gen_range bounds; scale = inf would never accept a value."""
    b = oracle.builder()
    world, cam, _ = pkg.scenes.cornell_box_scene(b, 8, 8)
    sc = b.scene(world)
    for e0, e1 in ((0.0, float("inf")), (float("-inf"), 0.0), (-3e38, 3e38), (float("nan"), 1.0), (1.0, 1.0)):
        cam.exposure_start, cam.exposure_end = e0, e1
        with pytest.raises(pkg.RtError) as e:
            sc.par_cast(cam, 8, 8, 1)
        assert e.value.code == -4, (e0, e1)


def test_perlin_at_coordinates_beyond_i32(pkg, oracle):
    """perlin.rs:53-55 `(p.x.floor() as i32 + di) & 255`: `as i32` saturates beyond 2^31 and i32::MAX + 1 wraps in a
    release build.  A Perlin texture scaled by 1e10 puts every lookup there (the sanitizer leg watches this one)."""
    S = pkg.scenes
    b = oracle.builder()
    world, cam, _ = S.book_final_scene(b, 16, 16, pkg.small_rng.SmallRng(0xDEADBEEF))   # installs Perlin tables
    far = b.lambertian(b.perlin(1e10))
    img = b.scene([b.translate(S.v(0.0, 0.0, 0.0), b.sphere(200.0, far))] + world).par_cast(cam, 16, 16, 2)
    assert np.isfinite(img).all()
