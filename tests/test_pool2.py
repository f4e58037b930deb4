"""The second flat program (flat_scene.h "the list level, hoisted") and the pool-2 kernel that walks it (rt_pool2.h).

CPU part: the flattener -- which worlds get a second program, what stands in it.  GPU part: render_full_pool2 (option pool2 = 2: every
frame size) against the oracle and against render_full_pool (pool2 = 0) on book-2, on its parts and on a world built to exercise
every record kind of the second program: two OP_LIST records, a FlipNormals-wrapped and a Translate{RotateY}-wrapped Bvh, a Bvh
BEHIND a wrapper (the POP restores the ray from the slot), a moving and a checker-textured sphere under a Bvh, a medium, a sky dome.
"""
import numpy as np
import pytest

from conftest import assert_bit_equal
from scene_cases import build_case

OP_BOX, OP_SPHERE, OP_RECT, OP_PUSH, OP_POP, OP_MEDIUM, OP_PRISM, OP_SEG, OP_EXT, OP_LIST = 1, 2, 3, 4, 5, 6, 7, 9, 12, 13
P2_PRIMS, P2_MEDIUM, P2_WRAPPED = 1, 2, 3
F_P2_DEAD_POP = 1 << 12
KEYS = ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")


def ops(words):
    return [int(w) & 0xff for w in words[:, 7]]


def mixed_world(pkg, b, nx, ny):
    """P, B, W(flip), M, W(translate{rotate_y}), B, P P: five items, two OP_LIST records in front of wrappers, one at the end."""
    S = pkg.scenes
    rng = pkg.small_rng.SmallRng(7)
    white = b.lambertian(b.constant(S.vfrom(0.73)))
    chk = b.lambertian(b.checker(b.constant(S.v(0.2, 0.3, 0.1)), b.constant(S.vfrom(0.9))))
    glass, metal = b.dielectric(1.5), b.metal(S.v(0.8, 0.8, 0.9), 0.2)

    def blobs(n, centre, spread, mats, moving=False):
        out = []
        for i in range(n):
            c = centre + S.f32(spread) * (rng.gen_vec3() - S.f32(0.5))
            sp = b.sphere(float(S.f32(12.0) + S.f32(10.0) * rng.gen_f32()), mats[i % len(mats)])
            if moving and i % 5 == 0:
                sp = b.linear_move(sp, S.v(0.0, 25.0, 0.0))
            out.append(b.translate(c, sp))
        return out

    world = [b.translate(S.v(120.0, 90.0, 120.0), b.sphere(70.0, metal))]
    world.append(b.bvh(blobs(40, S.v(300.0, 120.0, 250.0), 260.0, [white, chk, glass], moving=True)
                       + [b.rect_prism(S.v(0.0, -20.0, 0.0), S.v(555.0, 0.0, 555.0), white)], (0.0, 1.0)))
    world.append(b.flip_normals(b.bvh(blobs(24, S.v(420.0, 330.0, 330.0), 150.0, [white, metal]), (0.0, 1.0))))
    world.append(b.constant_medium(b.translate(S.v(278.0, 278.0, 278.0), b.sphere(170.0, glass)), 0.004, b.isotropic(b.constant(S.v(0.3, 0.5, 0.9)))))
    world.append(b.translate(S.v(80.0, 260.0, 300.0), b.rotate_y(25.0, b.bvh(blobs(48, S.v(60.0, 60.0, 60.0), 130.0, [white, chk]), (0.0, 1.0)))))
    world.append(b.bvh(blobs(33, S.v(200.0, 420.0, 150.0), 200.0, [glass, white, metal]), (0.0, 1.0)))
    world.append(b.rect(S.Y, (113.0, 443.0), (127.0, 432.0), 554.0, b.diffuse_light(b.constant(S.vfrom(1.0)), 7.0)))
    world.append(b.flip_normals(b.sphere(3000.0, b.diffuse_light(b.constant(S.v(0.5, 0.6, 0.8)), 0.6))))
    cam, exp = S._cornell_camera(b.be, nx, ny)
    return world, cam, exp


def random_p2_world(pkg, b, seed, nx, ny):
    """A random world of the shape the second program takes: a list of plain primitives (spheres incl. moving ones, rects, prisms),
    media over one primitive, Bvhs of primitives and once-wrapped such Bvhs (Translate, RotateY, Translate{RotateY}, Scale, FlipNormals,
    LinearMove), at most five runs of list-level items and two media; every material and texture kind."""
    S = pkg.scenes
    rs = np.random.RandomState(seed)
    b.set_perlin_tables(*pkg.small_rng.perlin_tables(int(rs.randint(1, 1 << 30))))

    def f(lo, hi):
        return float(np.float32(rs.uniform(lo, hi)))

    def vec(lo, hi):
        return S.v(f(lo, hi), f(lo, hi), f(lo, hi))

    def texture():
        k = rs.randint(0, 4)
        if k <= 1:
            return b.constant(vec(0.1, 0.95))
        if k == 2:
            return b.perlin(f(0.02, 0.3))
        return b.checker(b.constant(vec(0.1, 0.9)), b.constant(vec(0.1, 0.9)))

    def material(light=True):
        k = rs.randint(0, 5 if light else 4)
        if k == 0:
            return b.lambertian(texture())
        if k == 1:
            return b.metal(vec(0.3, 0.95), f(0.0, 1.0))
        if k == 2:
            return b.dielectric(f(1.1, 2.0))
        if k == 3:
            return b.isotropic(texture())
        return b.diffuse_light(texture(), f(0.5, 5.0))

    def primitive(spread=260.0):
        k = rs.randint(0, 5)
        c = S.v(278.0, 278.0, 278.0) + S.f32(spread) * vec(-1.0, 1.0)
        if k <= 1:
            return b.translate(c, b.sphere(f(15, 70), material()))
        if k == 2:
            return b.translate(c, b.linear_move(b.sphere(f(15, 50), material()), vec(-40, 40)))
        if k == 3:
            a0, b0 = f(0, 400), f(0, 400)
            r = b.rect(int(rs.randint(0, 3)), (a0, a0 + f(40, 250)), (b0, b0 + f(40, 250)), f(0, 555), material())
            return b.flip_normals(r) if rs.rand() < 0.3 else r
        p0 = c - S.v(40.0, 40.0, 40.0)
        return b.rect_prism(p0, p0 + S.v(f(30, 120), f(30, 120), f(30, 120)), material())

    def bvh():
        return b.bvh([primitive() for _ in range(int(rs.randint(34, 70)))], (0.0, 1.0))  # (> 32 boxes: the pool kernels, not the lock-step one)

    def wrapped():
        k = rs.randint(0, 6)
        inner = bvh()
        if k == 0:
            return b.translate(vec(-60, 60), inner)
        if k == 1:
            return b.rotate_y(f(-40, 40), inner)
        if k == 2:
            return b.translate(vec(-60, 60), b.rotate_y(f(-40, 40), inner))
        if k == 3:
            return b.scale(S.v(f(0.7, 1.3), f(0.7, 1.3), f(0.7, 1.3)), inner)
        if k == 4:
            return b.flip_normals(inner)
        return b.translate(vec(-30, 30), b.linear_move(inner, vec(-20, 20)))

    def medium():
        bound = b.translate(S.v(278.0, 278.0, 278.0) + S.f32(150.0) * vec(-1.0, 1.0), b.sphere(f(60, 400), b.dielectric(1.5)))
        if rs.rand() < 0.3:
            bound = b.rect(int(rs.randint(0, 3)), (0.0, 555.0), (0.0, 555.0), f(100, 500), b.dielectric(1.5))  # a Rect never hits twice: no crossing
        return b.constant_medium(bound, f(0.001, 0.02), b.isotropic(texture()))

    world, items, media, have_bvh = [], 0, 0, False
    last = None  # kind of the last list-level item run ("P" runs merge)
    for _ in range(int(rs.randint(3, 9))):
        k = rs.choice(["P", "P", "M", "B", "W"])
        if k == "P":
            if last != "P" and items >= 5:
                continue
            world.append(primitive(300.0))
            items += last != "P"
        elif k == "M":
            if media >= 2 or items >= 5:
                continue
            world.append(medium())
            media += 1
            items += 1
        elif k == "B":
            world.append(bvh())
            have_bvh = True
        else:
            if items >= 5:
                continue
            world.append(wrapped())
            items += 1
            have_bvh = True
        last = k
    if not have_bvh:
        world.append(bvh())
        last = "B"
    if last == "P" or items < 5:
        world.append(b.flip_normals(b.sphere(4000.0, b.diffuse_light(b.constant(S.v(0.6, 0.7, 0.9)), 0.8))))  # a sky dome: a plain primitive
    cam, exp = S._cornell_camera(b.be, nx, ny)
    return world, cam, exp


N_P2_FUZZ = 32


@pytest.mark.parametrize("seed", range(N_P2_FUZZ))
def test_random_worlds_of_the_second_program_flatten(pkg, seed):
    b = pkg.load().builder()
    world, _, _ = random_p2_world(pkg, b, 7000 + seed, 32, 24)
    w, items, n_media, n_wrapped = b.flatten_pool2(world)
    assert len(w) != 0 and 1 <= len(items) <= 5 and n_media <= 2, (seed, len(w), items)
    o = ops(w)
    end = o.index(0)
    assert all(x in (OP_BOX, OP_SPHERE, OP_RECT, OP_PRISM, OP_PUSH, OP_POP, OP_LIST, OP_EXT) for x in o[:end])   # the walk: Bvh streams, wrappers, list records
    assert sum(int(w[i, 5]) for i, x in enumerate(o[:end]) if x == OP_LIST) == len(items)                        # every item is committed exactly once
    assert o[:end].count(OP_PUSH) == o[:end].count(OP_POP) == n_wrapped


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_P2_FUZZ))
def test_random_worlds_of_the_second_program_bit_exact(pkg, gpu, oracle, seed):
    nx, ny, ns = 48, 32, 5
    bo, bg = oracle.builder(), gpu.builder()
    world_o, cam_o, _ = random_p2_world(pkg, bo, 7000 + seed, nx, ny)
    world_g, cam_g, _ = random_p2_world(pkg, bg, 7000 + seed, nx, ny)
    so, sg = bo.scene(world_o), bg.scene(world_g)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    sg.set_option("sync", 0)
    for pool2 in (2, 0):
        sg.set_option("pool2", pool2)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "random second-program world %d, pool2=%d" % (seed, pool2))
        for k in KEYS:
            assert st_g[k] == st_o[k], (seed, pool2, k, st_g[k], st_o[k])
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "random second-program world %d, pool2=%d (timed variant)" % (seed, pool2))


def book2_part(keep):
    def fn(pkg, b, nx, ny):
        world, cam, exp = pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF))
        return [world[i] for i in keep], cam, exp
    return fn


def test_second_program_of_book2(pkg):
    b = pkg.load().builder()
    world, _, _ = pkg.scenes.book_final_scene(b, 32, 32, pkg.small_rng.SmallRng(0xDEADBEEF))
    w, items, n_media, n_wrapped = b.flatten_pool2(world)
    w1, _ = b.flatten(world)
    o = ops(w)
    assert len(w) == len(w1) == 4213 and (n_media, n_wrapped) == (2, 1)
    # the walk: the floor Bvh (799 BOX + 400 PRISM), ONE list record, the wrapped cube, END; the item records behind END
    assert o[:1199].count(OP_BOX) == 799 and o[:1199].count(OP_PRISM) == 400
    assert o[1199] == OP_LIST and (int(w[1199, 4]), int(w[1199, 5])) == (0, 5)
    assert o[1200] == OP_PUSH and o[1201] == OP_BOX and o[4200] == OP_POP and o[4201] == 0
    assert int(w[4200, 7]) & F_P2_DEAD_POP  # nothing behind the cube needs the ray
    assert [k for k, _, _, _ in items] == [P2_PRIMS, P2_MEDIUM, P2_MEDIUM, P2_PRIMS, P2_WRAPPED]
    assert items[0][1:3] == (4202, 4208) and [o[i] for i in range(4202, 4208)] == [OP_RECT, OP_SPHERE, OP_EXT, OP_SPHERE, OP_SPHERE, OP_SPHERE]
    assert items[1][1] == 4208 and o[4208] == OP_MEDIUM and o[4209] == OP_SPHERE and items[2][1] == 4210
    assert items[3][1:3] == (4212, 4213) and items[4][1:] == (1200, 1201, 4201)
    # 1 / density rides in the MEDIUM record (object.rs:562's divide, done once on the host: the same f32 quotient)
    for pc in (4208, 4210):
        dens = w[pc, 0:1].view(np.float32)[0]
        assert w[pc, 1:2].view(np.float32)[0] == np.float32(1.0) / dens
    # the first program keeps its hoisted segment: OP_SEG in front of the five plain primitives, skip pointer behind them
    o1 = ops(w1)
    seg = o1.index(OP_SEG)
    assert o1[seg + 1:seg + 7] == [OP_RECT, OP_SPHERE, OP_EXT, OP_SPHERE, OP_SPHERE, OP_SPHERE] and int(w1[seg, 6]) == seg + 7


def test_which_worlds_get_a_second_program(pkg):
    be = pkg.load()
    for name, want in (("book2", True), ("book2_bvh", False), ("cornell", False), ("bench", False), ("book1", False), ("volume_bvh", False),
                       ("simple_light", False), ("cornell_smoke", False)):
        b = be.builder()
        from scene_cases import CASES
        world, _, _ = CASES[name][0](pkg, b, 32, 32)
        w, items, _, _ = b.flatten_pool2(world)
        assert (len(w) != 0) == want, name
        w1, _ = b.flatten(world)
        if not want and name not in ("book1",):
            assert OP_LIST not in ops(w1)
    # lean, FEAT_DEEP and Bvh-less programs get no OP_SEG either
    for name in ("book1", "cornell", "simple_light"):
        b = be.builder()
        world, _, _ = CASES[name][0](pkg, b, 32, 32)
        assert OP_SEG not in ops(b.flatten(world)[0]), name


def test_second_program_with_two_list_records_and_a_live_pop(pkg):
    b = pkg.load().builder()
    world, _, _ = mixed_world(pkg, b, 32, 32)
    w, items, n_media, n_wrapped = b.flatten_pool2(world)
    o = ops(w)
    assert [k for k, _, _, _ in items] == [P2_PRIMS, P2_WRAPPED, P2_MEDIUM, P2_WRAPPED, P2_PRIMS] and (n_media, n_wrapped) == (1, 2)
    lists = [i for i, x in enumerate(o) if x == OP_LIST]
    end = o.index(0)
    assert [(int(w[i, 4]), int(w[i, 5])) for i in lists] == [(0, 1), (1, 1), (2, 2), (4, 1)]
    assert lists[0] == 0 and o[1] == OP_BOX and lists[-1] == end - 1
    pops = [i for i, x in enumerate(o[:end]) if x == OP_POP]
    assert len(pops) == 2 and not (int(w[pops[0], 7]) & F_P2_DEAD_POP) and not (int(w[pops[1], 7]) & F_P2_DEAD_POP)  # a Bvh / a list record follows
    for k, a, bb, c in items:
        if k == P2_WRAPPED:
            assert o[a] == OP_PUSH and bb == a + 1 and o[bb] == OP_BOX and o[c - 1] == OP_POP
    # six items are one too many: the world falls back to the first program
    extra = world[:1] + [world[3]] + world
    assert len(b.flatten_pool2(extra)[0]) == 0


PARTS = {
    "book2": (book2_part(range(10)), 72, 56, 6),
    "floor_light": (book2_part([0, 1]), 64, 48, 5),
    "floor_light_cube": (book2_part([0, 1, 9]), 64, 48, 5),
    "no_media": (book2_part([0, 1, 2, 3, 4, 5, 8, 9]), 64, 48, 5),
    "media_first": (book2_part([6, 7, 0, 9, 1]), 48, 40, 4),
    "mixed": (mixed_world, 72, 56, 6),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PARTS))
def test_pool2_kernel_against_the_oracle_and_the_first_kernel(pkg, gpu, oracle, name):
    fn, nx, ny, ns = PARTS[name]
    bo = oracle.builder()
    world_o, cam_o, _ = fn(pkg, bo, nx, ny)
    so = bo.scene(world_o)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    bg = gpu.builder()
    world_g, cam_g, _ = fn(pkg, bg, nx, ny)
    assert len(bg.flatten_pool2(world_g)[0]) != 0, "%s has no second program" % name
    sg = bg.scene(world_g)
    sg.set_option("sync", 0)  # small frames: the pool kernels, not the lock-step one
    xs, ys = np.meshgrid(np.arange(0, nx, 7, dtype=np.uint32), np.arange(0, ny, 5, dtype=np.uint32))
    xs, ys = xs.ravel(), ys.ravel()
    ss = (xs + ys) % ns
    rgb_o, info_o = so.debug_samples(cam_o, nx, ny, ns, xs, ys, ss)
    for pool2 in (2, 0):
        sg.set_option("pool2", pool2)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "%s pool2=%d" % (name, pool2))
        for k in KEYS:
            assert st_g[k] == st_o[k], (name, pool2, k, st_g[k], st_o[k])
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "%s pool2=%d (timed variant)" % (name, pool2))
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, rank=2, nranks=3, tile_w=8, tile_h=8),
                         so.par_cast(cam_o, nx, ny, ns, rank=2, nranks=3, tile_w=8, tile_h=8), "%s shard" % name)
        rgb, info = sg.debug_samples(cam_g, nx, ny, ns, xs, ys, ss, trace_kernel=True)  # per-path traces of the production kernel
        assert np.array_equal(info, info_o), (name, pool2)
        assert_bit_equal(rgb, rgb_o, "%s traces pool2=%d" % (name, pool2))


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [
    {"drain_share": 0}, {"p2_refill": 4, "p2_box_leave": 2, "p2_park": 3}, {"p2_sphere": 1, "p2_prism": 1, "p2_list": 1, "p2_push": 1},
    {"p2_sphere": 64, "p2_prism": 64, "p2_list": 64, "p2_push": 64, "p2_park": 64}, {"lpt": 0}, {"lpt": 2, "lpt_phase1": 2, "lpt_deep": 0},
    {"mat_lds": 0}, {"scratch_mb": 1}, {"block": 256}, {"bounces": 3},
])
def test_pool2_schedule_switches_never_change_a_bit(pkg, gpu, oracle, opts):
    for name, nx, ny, ns in (("book2", 160, 112, 9), ("mixed", 96, 64, 7)):
        fn = PARTS[name][0]
        bo, bg = oracle.builder(), gpu.builder()
        world_o, cam_o, _ = fn(pkg, bo, nx, ny)
        world_g, cam_g, _ = fn(pkg, bg, nx, ny)
        so, sg = bo.scene(world_o), bg.scene(world_g)
        kw = {"max_bounces": opts["bounces"]} if "bounces" in opts else {}
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True, **kw)
        sg.set_option("sync", 0)
        sg.set_option("pool2", 2)
        for k, v in opts.items():
            if k != "bounces":
                sg.set_option(k, v)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True, **kw)
        assert_bit_equal(img_g, img_o, "%s %s" % (name, opts))
        for k in KEYS:
            assert st_g[k] == st_o[k], (name, opts, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, **kw), img_o, "%s %s (timed variant)" % (name, opts))


@pytest.mark.gpu
@pytest.mark.parametrize("hoist,mat_lds", [(0, 1), (1, 0), (0, 0)])
def test_first_kernel_switches_hoist_and_mat_lds(pkg, gpu, oracle, hoist, mat_lds):
    """The FIRST full-feature pool kernel with its hoisted segment (OP_SEG) and its LDS material records switched off, one at a time
    and together: the same bits and the same counters as the oracle."""
    nx, ny, ns = 96, 72, 6
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book2", nx, ny)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book2", nx, ny)
    for k, v in (("sync", 0), ("pool2", 0), ("hoist", hoist), ("mat_lds", mat_lds)):
        sg.set_option(k, v)
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    assert_bit_equal(img_g, img_o, "hoist=%d mat_lds=%d" % (hoist, mat_lds))
    for k in KEYS:
        assert st_g[k] == st_o[k], (hoist, mat_lds, k)
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "hoist=%d mat_lds=%d (timed variant)" % (hoist, mat_lds))


@pytest.mark.gpu
def test_pool2_takes_the_large_frames_by_default(pkg, gpu, capfd):
    """pool2 = 1 (default): the pool-2 kernel renders frames of >= 32 M samples, the first kernel the smaller ones; the fuzz graphs
    with a second program run on both (tests/test_fuzz.py sets pool2 = 2 through RTG_POOL2)."""
    sg, cam, _, _, _ = build_case(pkg, gpu, "book2", 800, 800)
    sg.set_option("verbose", 1)
    sg.par_cast(cam, 800, 800, 64)
    big = capfd.readouterr().err
    sg.par_cast(cam, 800, 800, 8)
    small = capfd.readouterr().err
    assert "pool 2" in big and "pool 2" not in small
