"""Randomised object graphs: the flattened HIP path against the recursive oracle, bit for bit."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from fuzz_scenes import random_camera, random_world

N_SCENES = 48
N_BOUNDARY_SCENES = 16


def _build(pkg, backend, seed, nx, ny, general_boundaries=False, deep_shapes=False):
    rs = np.random.RandomState(seed)
    b = backend.builder()
    world = random_world(pkg, b, rs, general_boundaries=general_boundaries, deep_shapes=deep_shapes)
    cam = random_camera(pkg, backend, rs, nx, ny)
    return b, world, cam


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_graphs_flatten_on_host(pkg, seed):
    """CPU-only: every random graph flattens (no GPU needed) and the program is well formed."""
    b, world, _ = _build(pkg, pkg.load(), 1000 + seed, 24, 16)
    words, feat = b.flatten(world)
    ops = words[:, 7] & 0xff
    assert ops[-1] == 0 and (ops[:-1] != 0).all()
    box = ops == 1
    assert (words[box, 6] > np.nonzero(box)[0]).all() and (words[box, 6] < len(words)).all()
    assert int((ops == 4).sum()) == int((ops == 5).sum())          # PUSH / POP balanced
    med = np.nonzero(ops == 6)[0]
    assert np.isin(ops[med + 1], (2, 3)).all()                     # a medium's boundary record follows it
    assert (words[med, 4] == med + 2).all()                        # ... and the medium names where it ends


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_graph_boundaries_flatten_on_host(pkg, seed):
    """CPU-only: media bounded by object graphs flatten to MEDIUM + a closed boundary stream."""
    b, world, _ = _build(pkg, pkg.load(), 5000 + seed, 24, 16, general_boundaries=True)
    words, feat = b.flatten(world)
    ops, flags = words[:, 7] & 0xff, words[:, 7]
    med = np.nonzero(ops == 6)[0]
    for m in med:
        end = int(words[m, 4])
        assert m + 1 < end <= len(words) - 1
        inner = ops[m + 1:end]
        assert not np.isin(inner, (0, 6)).any()                    # no END / nested medium inside a boundary
        assert int((inner == 4).sum()) == int((inner == 5).sum())  # its PUSH / POP pairs close inside it
        general = bool(flags[m] & (1 << 14))
        assert general == (end != m + 2) or general
        assert (not general) or (feat & 16)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_SCENES))
def test_fuzz_graphs_bit_exact(pkg, gpu, oracle, seed):
    nx, ny, ns = 40, 24, 5
    bg, wg, cam_g = _build(pkg, gpu, 1000 + seed, nx, ny)
    bo, wo, cam_o = _build(pkg, oracle, 1000 + seed, nx, ny)
    assert bytes(cam_g) == bytes(cam_o)
    sg, so = bg.scene(wg), bo.scene(wo)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    # a frame this small is routed to the lock-step kernel by default (rtg_launch.inc: tiny frames): both schedules, explicitly
    # (programs the pool kernels do not take -- lean ones, FEAT_DEEP -- ignore the option)
    # (pool2 = 2: graphs with a second program -- flat_scene.h "the list level, hoisted" -- also on the pool-2 kernel at this size)
    for sync, pool2 in ((0, 0), (0, 2), (1, 0)):
        sg.set_option("sync", sync)
        sg.set_option("pool2", pool2)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "fuzz scene %d sync=%d pool2=%d" % (seed, sync, pool2))
        for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (seed, sync, k, st_g[k], st_o[k])
        # ... and the timed instantiation of the same kernel (no counters: other register allocation, other spills)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "fuzz scene %d sync=%d, production variant" % (seed, sync))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_BOUNDARY_SCENES))
def test_fuzz_graph_boundaries_bit_exact(pkg, gpu, oracle, seed):
    """ConstantMedium<O> with O an object graph (object.rs:441-497): nested boundary walk vs the oracle."""
    nx, ny, ns = 40, 24, 5
    bg, wg, cam_g = _build(pkg, gpu, 5000 + seed, nx, ny, True)
    bo, wo, cam_o = _build(pkg, oracle, 5000 + seed, nx, ny, True)
    sg, so = bg.scene(wg), bo.scene(wo)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    for sync in (0, 1):   # pool kernel and lock-step kernel (see test_fuzz_graphs_bit_exact)
        sg.set_option("sync", sync)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "boundary fuzz scene %d sync=%d" % (seed, sync))
        for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (seed, sync, k, st_g[k], st_o[k])
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "boundary fuzz scene %d sync=%d, production variant" % (seed, sync))


N_DEEP_SCENES = 24


@pytest.mark.parametrize("seed", range(N_DEEP_SCENES))
def test_fuzz_deep_shapes_flatten_on_host(pkg, seed):
    """CPU-only: the shapes round 2 refused (a medium inside a medium's boundary, a medium below And below Bvh, 5-7 nested
    wrappers) flatten; a program that holds one carries FEAT_DEEP, SAVE / MERGE pairs nest properly, media inside boundary
    streams are closed streams themselves."""
    b, world, _ = _build(pkg, pkg.load(), 9000 + seed, 24, 16, general_boundaries=True, deep_shapes=True)
    words, feat = b.flatten(world)
    ops = words[:, 7] & 0xff
    OP_SAVE, OP_MERGE = 10, 11   # flat_scene.h
    assert int((ops == OP_SAVE).sum()) == int((ops == OP_MERGE).sum())
    level = 0
    for o in ops:
        level += 1 if o == OP_SAVE else -1 if o == OP_MERGE else 0
        assert 0 <= level <= 4
    if (ops == OP_SAVE).any():
        assert feat & 128
    depth = max_depth = 0
    for o in ops:
        depth += 1 if o == 4 else -1 if o == 5 else 0
        max_depth = max(max_depth, depth)
    med = np.nonzero(ops == 6)[0]
    nested = any(np.isin(ops[m + 1:int(words[m, 4])], (6,)).any() for m in med)
    assert bool(feat & 128) == bool((ops == OP_SAVE).any() or nested or _deep_wrappers(words)), (seed, feat)


def _deep_wrappers(words):
    """more than 4 PUSH levels open at once, counted per stream (a boundary stream starts a fresh count)"""
    ops = words[:, 7] & 0xff

    def scan(lo, hi):
        depth, i, deep = 0, lo, False
        while i < hi:
            o = ops[i]
            if o == 4:
                depth += 1
                deep |= depth > 4
            elif o == 5:
                depth -= 1
            elif o == 6:
                end = int(words[i, 4])
                deep |= scan(i + 1, end)
                i = end
                continue
            i += 1
        return deep
    return scan(0, len(ops))


def test_fuzz_deep_shapes_cover_all_three(pkg):
    """the generator really draws the three shapes (else the GPU test below proves nothing)"""
    seen = set()
    for seed in range(N_DEEP_SCENES):
        b, world, _ = _build(pkg, pkg.load(), 9000 + seed, 24, 16, general_boundaries=True, deep_shapes=True)
        words, feat = b.flatten(world)
        ops = words[:, 7] & 0xff
        if (ops == 10).any():   # OP_SAVE (flat_scene.h)
            seen.add("medium below And below Bvh")
        if any(np.isin(ops[m + 1:int(words[m, 4])], (6,)).any() for m in np.nonzero(ops == 6)[0]):
            seen.add("medium inside a boundary")
        if _deep_wrappers(words):
            seen.add("deep wrappers")
    assert len(seen) == 3, seen


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_DEEP_SCENES))
def test_fuzz_deep_shapes_bit_exact(pkg, gpu, oracle, seed):
    """Every graph the reference's types allow renders: the three shapes the scheduled kernels do not walk go through the
    general walk (rt_trace.h walk_deep, baseline kernel) -- bit-exact against the oracle's recursion, counters included."""
    nx, ny, ns = 40, 24, 5
    bg, wg, cam_g = _build(pkg, gpu, 9000 + seed, nx, ny, True, True)
    bo, wo, cam_o = _build(pkg, oracle, 9000 + seed, nx, ny, True, True)
    sg, so = bg.scene(wg), bo.scene(wo)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    assert_bit_equal(img_g, img_o, "deep fuzz scene %d" % seed)
    for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st_g[k] == st_o[k], (seed, k, st_g[k], st_o[k])
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "deep fuzz scene %d, production variant" % seed)


@pytest.mark.gpu
@pytest.mark.parametrize("wrappers,media_levels", [(6, 1), (12, 1), (30, 1), (6, 3), (12, 3)])
def test_general_walk_sized_by_the_graph(pkg, gpu, oracle, wrappers, media_levels):
    """FEAT_DEEP graphs run an instantiation of the general walk sized for what they need (flat_scene.h FEAT_DEEP_FEW_WRAPPERS =
    256: <= 8 wrappers open at once, FEAT_DEEP_ONE_LEVEL = 512: no medium inside a medium's boundary).  Every combination --
    few / many wrappers x one / three levels -- against the oracle, and against the largest instantiation (option deep_sized = 0)."""
    S = pkg.scenes
    nx, ny, ns = 48, 32, 6

    def build(be):
        b = be.builder()
        m = b.lambertian(b.constant(S.v(0.6, 0.5, 0.4)))
        iso = b.isotropic(b.constant(S.vfrom(0.9)))
        inner = b.sphere(1.0, m)
        for i in range(wrappers):   # alternating wrapper kinds, `wrappers` levels deep
            inner = (b.translate(S.v(0.01 * i, 0.0, 0.0), inner) if i % 3 == 0 else
                     b.rotate_y(3.0, inner) if i % 3 == 1 else b.scale(S.v(1.0, 1.02, 1.0), inner))
        level = b.sphere(0.8, m)
        for _ in range(media_levels):   # a medium whose boundary holds a medium whose boundary ...
            level = b.constant_medium(b.and_(b.rect_prism(S.v(-1.5, -1.5, -1.5), S.v(1.5, 1.5, 1.5), m), level), 0.4, iso)
        world = [inner, b.translate(S.v(3.0, 0.0, 0.0), level),
                 b.flip_normals(b.sphere(50.0, b.diffuse_light(b.constant(S.v(0.7, 0.8, 1.0)), 1.0)))]
        cam = be.camera_look(S.v(1.5, 1.0, 9.0), S.v(1.5, 0.0, 0.0), S.v(0.0, 1.0, 0.0), 35.0, nx / ny, 0.0, 10.0)
        return b, world, cam
    bg, wg, cam_g = build(gpu)
    bo, wo, cam_o = build(oracle)
    _, feat = bg.flatten(wg)
    assert feat & 128
    assert bool(feat & 256) == (wrappers <= 8) and bool(feat & 512) == (media_levels <= 1), hex(feat)
    sg, so = bg.scene(wg), bo.scene(wo)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    for sized in (1, 0):
        sg.set_option("deep_sized", sized)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "wrappers %d, media levels %d, deep_sized %d" % (wrappers, media_levels, sized))
        for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (wrappers, media_levels, sized, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "production variant")
