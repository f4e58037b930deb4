"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Bar: BIT-EXACT float32 framebuffers (north_star allows 1 ULP per channel; the tests demand 0)."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from scene_cases import CASES, build_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_framebuffer_bit_exact(pkg, gpu, oracle, name):
    sg, cam_g, nx, ny, ns = build_case(pkg, gpu, name)
    so, cam_o, _, _, _ = build_case(pkg, oracle, name)
    assert bytes(cam_g) == bytes(cam_o), "Camera::look differs between product and oracle"
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    assert_bit_equal(img_g, img_o, name)
    for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st_g[k] == st_o[k], (name, k, st_g[k], st_o[k])
    # the non-instrumented kernel variant is the one that is timed: must give the same image
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, name + " (timed variant)")


# ---- golden fixtures (committed, minted by tools/gen_golden.py from the oracle) ----------------------
import os  # noqa: E402

from probe_rays import probe_rays  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden_framebuffer(pkg, gpu, name):
    gold = np.load(os.path.join(GOLD, "framebuffers.npz"))[name]
    sg, cam, nx, ny, ns = build_case(pkg, gpu, name)
    assert_bit_equal(sg.par_cast(cam, nx, ny, ns), gold, name)


@pytest.mark.parametrize("name", sorted(CASES))
def test_per_sample_traces(pkg, gpu, name):
    """64 fixed (pixel, sample) keys per scene: colour, bounce count, RNG draws consumed, Aabb tests,
    primitive tests -- all must equal the oracle's (same traversal order, same draw order)."""
    g = np.load(os.path.join(GOLD, "samples.npz"))
    xs, ys, ss = g[name + ".keys"]
    sg, cam, nx, ny, ns = build_case(pkg, gpu, name)
    rgb, info = sg.debug_samples(cam, nx, ny, ns, xs, ys, ss)
    assert np.array_equal(info, g[name + ".info"]), name
    assert_bit_equal(rgb, g[name + ".rgb"], name)


@pytest.mark.parametrize("name", sorted(CASES))
def test_per_sample_traces_of_the_production_kernels(pkg, gpu, name):
    """The same 64 keys, read out of the per-sample trace table of the ray-pool kernel par_cast really runs for the scene
    (instrumented variant of render_lean_pool / render_full_pool, RTG_FLAG_TRACE_KERNEL): colour, bounce count, RNG draws,
    Aabb tests and primitive tests of every traced (pixel, sample) equal the oracle's -- a break in the schedule is
    localised to a path, not just to a frame."""
    g = np.load(os.path.join(GOLD, "samples.npz"))
    xs, ys, ss = g[name + ".keys"]
    sg, cam, nx, ny, ns = build_case(pkg, gpu, name)
    rgb, info = sg.debug_samples(cam, nx, ny, ns, xs, ys, ss, trace_kernel=True)
    assert np.array_equal(info, g[name + ".info"]), name
    assert_bit_equal(rgb, g[name + ".rgb"], name)
    # and on a shard: the table is indexed by the rank's own pixel work items
    keep = [i for i in range(len(xs)) if (((ny - 1 - int(ys[i])) // 16) * ((nx + 15) // 16) + int(xs[i]) // 16) % 2 == 1]
    if keep:
        rgb2, info2 = sg.debug_samples(cam, nx, ny, ns, xs[keep], ys[keep], ss[keep], trace_kernel=True, rank=1, nranks=2)
        assert np.array_equal(info2, g[name + ".info"][keep]) and np.array_equal(rgb2.view(np.uint32), g[name + ".rgb"][keep].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell", "book1", "book2", "volume_bvh", "checker_scale", "motion"])
def test_hit_top_probes(pkg, gpu, oracle, name):
    """World::hit_top on random + edge-case rays (zero direction components, NaN t, in-plane rays)."""
    g = np.load(os.path.join(GOLD, "hit_top.npz"))
    rays = g[name + ".rays"]
    assert np.array_equal(rays, probe_rays(name))
    sg, _, _, _, _ = build_case(pkg, gpu, name)
    out, mat = sg.debug_hit_top(rays, seed=5)
    assert np.array_equal(mat, g[name + ".mat"]), name
    assert_bit_equal(out, g[name + ".out"], name)
    # and live against the oracle on a different seed / ray set
    so, _, _, _, _ = build_case(pkg, oracle, name)
    rays2 = probe_rays(name, n_random=512, seed=99)
    og, mg = sg.debug_hit_top(rays2, seed=77)
    oo, mo = so.debug_hit_top(rays2, seed=77)
    assert np.array_equal(mg, mo)
    assert_bit_equal(og, oo, name + " live")


# ---- scalar building blocks --------------------------------------------------------------------------
def test_logf_full_rng_domain(gpu, oracle):
    """rt_logf on EVERY value rng.gen::<f32>() can produce (k * 2^-24, k < 2^24): GPU == CPU bitwise."""
    x = (np.arange(1 << 24, dtype=np.float64) / (1 << 24)).astype(np.float32)
    assert_bit_equal(gpu.debug_math(0, x), oracle.debug_math(0, x), "rt_logf")


def test_libm_restatements_and_ieee_ops(gpu, oracle):
    rs = np.random.RandomState(3)
    bits32 = rs.randint(0, 2 ** 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    anyf = bits32.view(np.float32)                    # every class: normals, denormals, inf, nan
    unit = rs.uniform(-0.5, 1.0, 1 << 20).astype(np.float32)
    wide = (rs.standard_normal(1 << 20) * 3000).astype(np.float32)
    assert_bit_equal(gpu.debug_math(0, anyf), oracle.debug_math(0, anyf), "rt_logf any")
    assert_bit_equal(gpu.debug_math(1, unit), oracle.debug_math(1, unit), "rt_pow5f")
    assert_bit_equal(gpu.debug_math(1, anyf), oracle.debug_math(1, anyf), "rt_pow5f any")
    assert_bit_equal(gpu.debug_math(2, wide), oracle.debug_math(2, wide), "rt_sinf")
    assert_bit_equal(gpu.debug_math(2, anyf), oracle.debug_math(2, anyf), "rt_sinf any")
    # correctly rounded sqrt / reciprocal / divide incl. denormals: the parity of everything else rests on these
    assert_bit_equal(gpu.debug_math(3, anyf), oracle.debug_math(3, anyf), "sqrt")
    assert_bit_equal(gpu.debug_math(4, anyf), oracle.debug_math(4, anyf), "1/x")
    other = np.roll(anyf, 7)
    assert_bit_equal(gpu.debug_math(5, anyf, other), oracle.debug_math(5, anyf, other), "x/y")
    den = (rs.randint(1, 1 << 23, 1 << 16).astype(np.uint32)).view(np.float32)   # denormal operands
    assert_bit_equal(gpu.debug_math(5, den, np.roll(den, 1)), oracle.debug_math(5, den, np.roll(den, 1)), "den/den")
    assert_bit_equal(gpu.debug_math(3, den), oracle.debug_math(3, den), "sqrt(den)")


# ---- sharding (multi-GPU path emulated on one device) --------------------------------------------------
@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_tile_shards_sum_to_full_frame(pkg, gpu, nranks):
    sg, cam, nx, ny, ns = build_case(pkg, gpu, "book1", 112, 80)
    full = sg.par_cast(cam, nx, ny, ns)
    acc = np.zeros_like(full)
    for r in range(nranks):
        part = sg.par_cast(cam, nx, ny, ns, rank=r, nranks=nranks)
        acc = acc + part
    assert_bit_equal(acc, full, "%d shards" % nranks)


def test_other_ranks_pixels_are_left_untouched(pkg, gpu):
    sg, cam, nx, ny, ns = build_case(pkg, gpu, "cornell", 64, 64)
    canvas = np.full((ny, nx, 3), 7.0, dtype=np.float32)
    out = sg.par_cast(cam, nx, ny, 2, rank=1, nranks=4, tile_w=32, tile_h=16, out=canvas)
    tiles_x = 2
    for row in range(ny):
        for x in range(0, nx, 16):
            tile = (row // 16) * tiles_x + x // 32
            if tile % 4 != 1:
                assert (out[row, x:x + 16] == 7.0).all()


# ---- error behaviour -------------------------------------------------------------------------------------
def test_gen_range_assertion(pkg, gpu):
    """camera.rs:55: gen_range(lo, hi) asserts lo < hi -> RTG_ERR_RANGE instead of a panic."""
    b = gpu.builder()
    world, cam, _ = pkg.scenes.cornell_box_scene(b, 16, 16)
    cam.exposure_start, cam.exposure_end = 1.0, 1.0
    with pytest.raises(pkg.RtError) as e:
        b.scene(world).par_cast(cam, 16, 16, 1)
    assert e.value.code == -4


def test_non_finite_exposure_is_refused(pkg, gpu):
    """exposure_end = inf or an overflowing (end - start): scale = inf in gen_range would spin every lane of the
    persistent kernel forever -> RTG_ERR_RANGE up front (rand 0.6.5 panics "non-finite boundaries")."""
    b = gpu.builder()
    world, cam, _ = pkg.scenes.cornell_box_scene(b, 16, 16)
    sc = b.scene(world)
    for e0, e1 in ((0.0, float("inf")), (float("-inf"), 0.0), (-3e38, 3e38), (float("nan"), 1.0)):
        cam.exposure_start, cam.exposure_end = e0, e1
        with pytest.raises(pkg.RtError) as e:
            sc.par_cast(cam, 16, 16, 1)
        assert e.value.code == -4, (e0, e1)


@pytest.mark.parametrize("name,nx,ny,ns", [("book1", 176, 112, 12), ("book2", 96, 80, 6), ("cornell", 64, 64, 8)])
def test_par_cast_multi_in_library_shard_and_reduce(pkg, gpu, oracle, name, nx, ny, ns):
    """rtg_par_cast_multi: one scene handle per device, tiles sharded inside the library, frames summed.
    This box has ONE GPU, so the handles share device 0 (summed on the device; the RCCL clique needs >= 2 distinct
    devices) -- 1, 2, 3 and 8 handles must all give the one-GPU frame and the oracle's, counters included."""
    so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
    ref, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    for n in (1, 2, 3, 8):
        scenes = []
        for _ in range(n):
            sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
            scenes.append(sg)
        img, st = gpu.par_cast_multi(scenes, cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img, ref, "%s, %d handles" % (name, n))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st[k] == st_o[k], (name, n, k)
        assert st["kernel_ms"] > 0
        assert_bit_equal(gpu.par_cast_multi(scenes, cam_g, nx, ny, ns), ref, "%s, %d handles (timed variant)" % (name, n))
    with pytest.raises(pkg.RtError):
        gpu.par_cast_multi(scenes, cam_g, nx, ny, ns, rank=1, nranks=2)


def test_par_cast_multi_through_the_real_rccl_symbols(pkg, gpu, oracle):
    """The RCCL leg of rtg_par_cast_multi EXECUTED (round 2 had only ever run the same-device add): with the scene option
    `force_rccl` the call goes dlopen(librccl) -> ncclCommInitAll -> ncclGroupStart / in-place ncclReduce(sum, root 0) /
    ncclGroupEnd even when all handles sit on one device (a clique of one on this one-GPU box; with >= 2 GPUs visible the
    handles are spread over two of them and the clique is a real one).  The frame must still be the oracle's, bit for bit;
    rtg_multi_reset reports how many ncclReduce calls were issued, destroys the communicators and unloads the library.
    A library that cannot be loaded is RTG_ERR_DEVICE with dlopen's reason -- not a crash (round 2: dlerror() called
    twice -> std::string(NULL))."""
    nx, ny, ns = 96, 64, 6
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", nx, ny)
    ref, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    n_dev = gpu.device_count()
    gpu.multi_reset()
    for n in (1, 2, 3):
        scenes = []
        for i in range(n):
            b = gpu.builder()
            world, cam_g, _ = pkg.scenes.random_scene(b, nx, ny)
            sg = b.scene(world, device=i % min(n_dev, 2))
            sg.set_option("force_rccl", 1)
            scenes.append(sg)
        img, st = gpu.par_cast_multi(scenes, cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img, ref, "force_rccl, %d handles" % n)
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st[k] == st_o[k], (n, k)
        assert_bit_equal(gpu.par_cast_multi(scenes, cam_g, nx, ny, ns), ref, "force_rccl, %d handles, second frame (cached clique)" % n)
    issued = gpu.multi_reset()
    # one ncclReduce per distinct device and frame: 3 handle counts x 2 frames x (1 device, or 2 once n >= 2)
    assert issued == (6 if n_dev < 2 else 2 + 4 + 4), issued
    # the same handle twice: refused (it would wipe its own tiles), nothing rendered
    with pytest.raises(pkg.RtError) as ei:
        gpu.par_cast_multi([scenes[0], scenes[0]], cam_g, nx, ny, ns)
    assert ei.value.code == pkg.capi.ERR_INVALID and "twice" in str(ei.value)
    # an unloadable library: a clean error with the loader's reason, and the next call with the default library works again
    gpu.multi_reset("/nonexistent/librccl-missing.so")
    with pytest.raises(pkg.RtError) as ei:
        gpu.par_cast_multi(scenes, cam_g, nx, ny, ns)
    assert ei.value.code == pkg.capi.ERR_DEVICE and "librccl" in str(ei.value) and "nonexistent" in str(ei.value)
    assert gpu.multi_reset() == 0
    assert_bit_equal(gpu.par_cast_multi(scenes, cam_g, nx, ny, ns), ref, "after the failed load")
    assert gpu.multi_reset() == (1 if n_dev < 2 else 2)


@pytest.mark.timeout(300)
def test_par_cast_multi_packed_collective(pkg, gpu, oracle):
    """Scene option multi_gather = 1: every handle ships only the tiles it owns (packed in work-item order) and the first device
    scatters them into the frame -- grouped ncclSend / ncclRecv instead of the full-frame ncclReduce.  Copies only: the frame is the
    oracle's bit for bit with 1, 2, 3 and 8 handles, ragged sizes and both tile sizes; with force_rccl the first handle's tiles take
    the send / recv path on this one-GPU box too (a clique of one), and rtg_multi_reset counts the transfers."""
    for name, nx, ny, ns in (("book1", 176, 112, 6), ("book2", 100, 76, 4)):
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        ref, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
        for n in (1, 2, 3, 8):
            scenes = []
            for _ in range(n):
                sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
                sg.set_option("multi_gather", 1)
                scenes.append(sg)
            img, st = gpu.par_cast_multi(scenes, cam_g, nx, ny, ns, stats=True)
            assert_bit_equal(img, ref, "%s, %d handles, packed" % (name, n))
            for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
                assert st[k] == st_o[k], (name, n, k)
            assert_bit_equal(gpu.par_cast_multi(scenes, cam_g, nx, ny, ns, tile_w=8, tile_h=8), ref, "%s, %d handles, packed, 8x8 tiles" % (name, n))
    n_dev = gpu.device_count()
    gpu.multi_reset()
    nx, ny, ns = 96, 64, 6
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", nx, ny)
    ref = so.par_cast(cam_o, nx, ny, ns)
    for n in (1, 3):
        scenes = []
        for i in range(n):
            b = gpu.builder()
            world, cam_g, _ = pkg.scenes.random_scene(b, nx, ny)
            sg = b.scene(world, device=i % min(n_dev, 2))
            sg.set_option("force_rccl", 1)
            sg.set_option("multi_gather", 1)
            scenes.append(sg)
        assert_bit_equal(gpu.par_cast_multi(scenes, cam_g, nx, ny, ns), ref, "force_rccl + packed, %d handles" % n)
    issued = gpu.multi_reset()
    assert issued == (2 if n_dev < 2 else 1 + 1), issued   # one transfer per handle that travels (one GPU: the first handle, to itself)


@pytest.mark.parametrize("name,nx,ny,ns,mb", [("cornell", 300, 300, 400, 16), ("book1", 300, 300, 400, 16), ("book2", 160, 160, 60, 1),
                                               ("book1", 200, 120, 37, 1)])
def test_bounded_sample_scratch_renders_in_passes(pkg, gpu, oracle, name, nx, ny, ns, mb, capfd):
    """The per-sample colour scratch is O(budget), not O(spp): with a budget (option scratch_mb) smaller than the frame's
    sample colours the frame is rendered in sample passes and the fold kernel carries the running per-pixel sum from pass
    to pass -- still the reference's left fold (lib.rs:365-374): bit-exact against the oracle, counters included, on all
    three pool kernels (lock-step / lean / full-feature), also sharded.  (Round 2 sent such frames to the 10x slower
    baseline kernel.)"""
    sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
    ref, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    sg.set_option("scratch_mb", mb)
    sg.set_option("verbose", 1)
    img, st = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    err = capfd.readouterr().err
    n_pass = len([l for l in err.splitlines() if "pool: samples [" in l])
    per_sample = ((nx + 15) // 16) * ((ny + 15) // 16) * 256 * 12
    assert n_pass >= 2 and n_pass == -(-ns // -(-ns // -(-ns // max(1, (mb << 20) // per_sample)))), (n_pass, err[-400:])
    assert "baseline" not in err
    sg.set_option("verbose", 0)
    assert_bit_equal(img, ref, "%s in %d passes" % (name, n_pass))
    for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st[k] == st_o[k], (name, k, st[k], st_o[k])
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), ref, name + " (timed variant)")
    # a shard of the frame in passes, into a caller-owned canvas
    canvas = np.zeros_like(ref)
    for r in range(3):
        canvas = sg.par_cast(cam_g, nx, ny, ns, rank=r, nranks=3, out=canvas)
    assert_bit_equal(canvas, ref, name + " 3 shards in passes")
    # back to one pass on the same handle
    sg.set_option("scratch_mb", 0)
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), ref, name + " (one pass again)")


@pytest.mark.parametrize("frames", [1, 3])
def test_frames_in_flight_on_one_handle(pkg, gpu, oracle, frames):
    """rtg_par_cast_device is asynchronous; round 2 allowed ONE frame in flight per handle and left the rest to the caller.
    Now a handle owns a ring of launch contexts: six frames (several seeds and sample counts) enqueued back to back on three
    streams WITHOUT any wait in between must all equal the oracle's, with one context (the calls serialise on the host) and
    with three (they overlap), on the lean and on the full-feature pool kernel.  (Device buffers and streams straight from
    the HIP runtime the library itself uses: no second runtime in the process.)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    for name, nx, ny in (("book1", 160, 96), ("book2", 96, 96)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        sg.set_option("frames_in_flight", frames)
        streams = []
        for _ in range(3):
            st = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(st)) == 0
            streams.append(st)
        jobs = [(7, 0xDEADBEEF), (20, 0xDEADBEEF), (3, 1234), (12, 99), (7, 5), (33, 0xDEADBEEF)]
        nbytes = nx * ny * 3 * 4
        outs = []
        for _ in jobs:
            d = C.c_void_p()
            assert hip.hipMalloc(C.byref(d), nbytes) == 0
            outs.append(d)
        assert hip.hipDeviceSynchronize() == 0
        for k, (ns, seed) in enumerate(jobs):
            p = pkg.make_params(nx, ny, ns, seed=seed)
            sg.par_cast_device(cam_g, p, outs[k], streams[k % 3])
        assert hip.hipDeviceSynchronize() == 0
        for k, (ns, seed) in enumerate(jobs):
            img = np.empty((ny, nx, 3), dtype=np.float32)
            assert hip.hipMemcpy(img.ctypes.data_as(C.c_void_p), outs[k], nbytes, 2) == 0   # hipMemcpyDeviceToHost
            assert_bit_equal(img, so.par_cast(cam_o, nx, ny, ns, seed=seed), "%s frame %d (%d contexts)" % (name, k, frames))
        for d in outs:
            hip.hipFree(d)
        for st in streams:
            hip.hipStreamDestroy(st)
    with pytest.raises(pkg.RtError):
        sg.set_option("frames_in_flight", 9)


def test_ragged_image_sizes(pkg, gpu, oracle):
    """Sizes that are not multiples of the 16x16 block / 8x8 wave tile, and 1-pixel images."""
    for (nx, ny) in [(1, 1), (17, 9), (33, 47), (15, 64)]:
        sg, cam_g, _, _, _ = build_case(pkg, gpu, "cornell", nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, "cornell", nx, ny)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, 3), so.par_cast(cam_o, nx, ny, 3), "%dx%d" % (nx, ny))


def test_bounce_cap_and_near_are_parameters(pkg, gpu, oracle):
    for mb, near in [(0, 0.001), (3, 0.001), (50, 0.1)]:
        sg, cam_g, nx, ny, _ = build_case(pkg, gpu, "book1", 40, 24)
        so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", 40, 24)
        a = sg.par_cast(cam_g, nx, ny, 4, max_bounces=mb, t_near=near)
        b_ = so.par_cast(cam_o, nx, ny, 4, max_bounces=mb, t_near=near)
        assert_bit_equal(a, b_, "max_bounces=%d" % mb)


def _albedo_scene(pkg, b, albedo, lean):
    """A mirror-ish pit of high-albedo spheres under a sky dome; `lean` = spheres under one Bvh, else a list
    world with a rect (full-feature kernel)."""
    S = pkg.scenes
    m = b.lambertian(b.constant(S.vfrom(albedo)))
    objs = [b.translate(S.v(0.0, -100.5, -1.0), b.sphere(100.0, m)), b.translate(S.v(0.0, 0.0, -1.0), b.sphere(0.5, m)),
            b.translate(S.v(1.0, 0.0, -1.0), b.sphere(0.5, b.metal(S.vfrom(albedo), 0.0))),
            b.flip_normals(b.sphere(50.0, b.diffuse_light(b.constant(S.v(0.7, 0.8, 1.0)), 1.0)))]
    world = [b.bvh(objs, (0.0, 1.0))] if lean else objs + [b.rect(1, (-1.0, 1.0), (-2.0, 0.0), 1.5, m)]
    cam = b.be.camera_look(S.v(-2, 2, 1), S.v(0, 0, -1), S.v(0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 1.0)
    return b.scene(world), cam


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("albedo,max_bounces", [(1.0, 50), (1.5, 50), (1.5, 200), (3.9, 63), (3.9, 64), (7.0, 50), (-0.5, 50)])
def test_albedo_range_routing(pkg, gpu, oracle, lean, albedo, max_bounces):
    """The pool kernels keep no accum (it is provably +0 while the path strength stays finite and non-negative);
    scenes or bounce caps outside that argument run on the baseline kernel.  Either way: the oracle's bits
    (albedo 7 overflows the strength to inf, albedo < 0 makes strength * 0 = -0)."""
    sg, cam_g = _albedo_scene(pkg, gpu.builder(), albedo, lean)
    so, cam_o = _albedo_scene(pkg, oracle.builder(), albedo, lean)
    a = sg.par_cast(cam_g, 36, 24, 6, max_bounces=max_bounces)
    b_ = so.par_cast(cam_o, 36, 24, 6, max_bounces=max_bounces)
    assert_bit_equal(a, b_, "albedo %g, %d bounces, lean=%s" % (albedo, max_bounces, lean))


def test_cxx_crate_mirror_example(pkg, gpu, tmp_path):
    """host/examples/crate_mirror_demo.cpp = the reference's src/main.rs transliterated against host/rtiow.hpp.
    Its PPM (print_ppm, lib.rs:344-361) must equal the one made from the ctypes path's framebuffer."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLD), "..", "rtiow-rust_amd", "host", "examples", "rtiow_main")
    exe = os.path.abspath(exe)
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    for which, case in (("cornell", "cornell"), ("motion", "motion"), ("volume", "volume")):
        out = subprocess.run([exe, which, "40", "24", "6"], check=True, capture_output=True, text=True).stdout
        sg, cam, nx, ny, _ = build_case(pkg, gpu, case, 40, 24)
        assert out == pkg.ppm.format_ppm(sg.par_cast(cam, nx, ny, 6)), which


@pytest.mark.parametrize("lean", [True, False])
def test_albedo_range_routing_through_a_checker(pkg, gpu, oracle, lean):
    """a Lambertian whose albedo is checker(constant(100), constant(-60)) inside a CLOSED sphere with a small
    light.  The strength overflows to +-inf along deep paths; the next non-emitter hit makes accum = inf * 0 = NaN,
    which the reference carries to the end (lib.rs:76).  The flattener has to see through the checker and route the
    scene off the accum-free pool kernels -- which would return +-inf (or 0) for those paths."""
    S = pkg.scenes

    def build(be):
        b = be.builder()
        chk = b.lambertian(b.checker(b.constant(S.vfrom(100.0)), b.constant(S.vfrom(-60.0))))
        objs = [b.flip_normals(b.sphere(10.0, chk)), b.translate(S.v(0.0, 0.0, -1.0), b.sphere(0.5, chk)),
                b.translate(S.v(1.0, 0.0, -1.0), b.sphere(0.5, b.metal(S.vfrom(0.9), 0.0))),
                b.translate(S.v(0.0, 5.0, 0.0), b.sphere(2.5, b.diffuse_light(b.constant(S.v(0.7, 0.8, 1.0)), 1.0)))]
        world = [b.bvh(objs, (0.0, 1.0))] if lean else objs + [b.rect(1, (-1.0, 1.0), (-2.0, 0.0), 1.5, chk)]
        cam = be.camera_look(S.v(-2, 2, 1), S.v(0, 0, -1), S.v(0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 1.0)
        return b.scene(world), cam
    sg, cam_g = build(gpu)
    so, cam_o = build(oracle)
    a = sg.par_cast(cam_g, 36, 24, 2)
    b_ = so.par_cast(cam_o, 36, 24, 2)
    assert_bit_equal(a, b_, "checker(100, -60), lean=%s" % lean)
    assert np.isnan(b_).any() and np.isinf(b_).any() and np.isfinite(b_).any(), "the case must mix NaN, inf and finite pixels"


@pytest.mark.parametrize("env", [
    {"RTG_KERNEL": "1"},                       # one-lane-per-pixel baseline kernel
    {"RTG_KERNEL": "3", "RTG_CHUNKS": "1"},    # ray-pool kernel, a slot folds its own pixel
    {"RTG_KERNEL": "3", "RTG_CHUNKS": "3"},    # ray-pool kernel, 3 sample chunks per pixel + fold kernel
    {"RTG_KERNEL": "3"},                       # ray-pool kernel, automatic chunking
    {"RTG_KERNEL": "3", "RTG_REFILL_MIN": "1", "RTG_SPHERE_MIN": "1", "RTG_BOX_LEAVE": "1"},   # degenerate schedules
    {"RTG_KERNEL": "3", "RTG_REFILL_MIN": "64", "RTG_SPHERE_MIN": "64", "RTG_BOX_LEAVE": "64"},
])
def test_every_kernel_variant_and_schedule_gives_the_same_bits(pkg, gpu, oracle, env, monkeypatch):
    """The schedule (kernel generation, chunking, thresholds) must never change a bit."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)   # capi.Scene forwards RTG_* to rtg_scene_set_option
    for name, nx, ny, ns in (("book1", 72, 40, 7), ("book1_list", 24, 16, 4)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "%s %s" % (name, env))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, env, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, rank=1, nranks=3), so.par_cast(cam_o, nx, ny, ns, rank=1, nranks=3),
                         "%s shard %s" % (name, env))


@pytest.mark.parametrize("sync", ["0", "1"])
def test_full_feature_scenes_on_either_schedule(pkg, gpu, oracle, sync, monkeypatch):
    """Full-feature scenes run on the ray-pool kernel (rt_pool_full.h) or, when the program holds no Bvh, on the lock-step
    kernel (rt_sync_full.h); `sync` = 1 / 0 forces one or the other for every scene.  Same bits, same counters, same traces."""
    monkeypatch.setenv("RTG_SYNC", sync)
    g = np.load(os.path.join(GOLD, "samples.npz"))
    for name in ("cornell", "volume", "volume_bvh", "cornell_smoke", "checker_scale", "book2", "motion", "simple_light"):
        sg, cam_g, nx, ny, ns = build_case(pkg, gpu, name)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name)
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "%s sync=%s" % (name, sync))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, sync, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "%s sync=%s (timed variant)" % (name, sync))
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, rank=1, nranks=3), so.par_cast(cam_o, nx, ny, ns, rank=1, nranks=3), "%s shard" % name)
        xs, ys, ss = g[name + ".keys"]
        rgb, info = sg.debug_samples(cam_g, nx, ny, ns, xs, ys, ss, trace_kernel=True)
        assert np.array_equal(info, g[name + ".info"]), (name, sync)
        assert_bit_equal(rgb, g[name + ".rgb"], name)
    # a frame large enough for the cost-ordered queue, and one with more waves than work
    for name, nx, ny, ns in (("cornell", 144, 128, 12), ("cornell_smoke", 160, 112, 9), ("volume", 17, 9, 3)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), so.par_cast(cam_o, nx, ny, ns), "%s %dx%d sync=%s" % (name, nx, ny, sync))


@pytest.mark.parametrize("env", [
    {"RTG_LPT": "0"},                                                    # natural order throughout
    {"RTG_LPT": "1", "RTG_LPT_PHASE1": "2", "RTG_LPT_DEEP": "0"},     # cost-ordered queue, block-major, every scatter counts
    {"RTG_LPT": "2", "RTG_LPT_PHASE1": "2", "RTG_LPT_DEEP": "0"},     # ... class-major, chunk-major inside a class
    {"RTG_LPT": "2", "RTG_LPT_PHASE1": "3", "RTG_LPT_DEEP": "2", "RTG_LPT_SHIFT": "3"},
    {"RTG_LPT": "2", "RTG_LPT_PHASE1": "2", "RTG_RAY_LDS": "0"},      # slot rays in global memory
])
def test_cost_ordered_queue_never_changes_a_bit(pkg, gpu, oracle, env, monkeypatch):
    """The cost-ordered work queue (rt_pool.h LptQueue) only engages on frames with >= 64 blocks and enough chunks;
    RTG_LPT_PHASE1 forces it on at a size the oracle renders in a second.  Lean and full-feature kernel, one rank
    and a shard."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)   # capi.Scene forwards RTG_* to rtg_scene_set_option
    for name, nx, ny, ns in (("book1", 176, 112, 12), ("cornell", 144, 128, 12), ("book2", 160, 112, 9)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_g, img_o, "%s %s" % (name, env))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, env, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "%s %s (timed variant)" % (name, env))
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", 352, 224)   # 308 tiles, 154 per rank
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", 352, 224)
    assert_bit_equal(sg.par_cast(cam_g, 352, 224, 9, rank=1, nranks=2), so.par_cast(cam_o, 352, 224, 9, rank=1, nranks=2),
                     "book1 shard %s" % (env,))


def test_tile_shapes_with_the_cost_ordered_queue(pkg, gpu, oracle, monkeypatch):
    """Cost blocks are 256 consecutive work items: a whole 16x16 tile, or a quarter / sixth ... of a larger one."""
    monkeypatch.setenv("RTG_LPT_PHASE1", "2")
    for name, nx, ny, ns in (("book1", 352, 224, 9), ("book2", 224, 160, 9)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        for kw in ({"tile_w": 32, "tile_h": 32}, {"tile_w": 48, "tile_h": 16, "rank": 2, "nranks": 3},
                   {"tile_w": 16, "tile_h": 64, "rank": 0, "nranks": 2}):
            assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, **kw), so.par_cast(cam_o, nx, ny, ns, **kw), "%s %s" % (name, kw))


def test_scene_reuse_across_sizes_and_sample_counts(pkg, gpu, oracle):
    """One scene handle, many calls: the library's scratch / path-slot buffers are grown lazily (a
    use-after-free here once produced a GPU memory fault when a second call needed a bigger scratch)."""
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", 96, 64)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", 96, 64)
    for (nx, ny, ns) in [(64, 64, 1), (64, 64, 2), (96, 64, 1), (96, 64, 9), (32, 16, 40), (96, 64, 3)]:
        cg = gpu.camera_look(pkg.scenes.v(13, 2, 3), pkg.scenes.v(0, 0, 0), pkg.scenes.v(0, 1, 0), 20.0, nx / ny, 0.1, 10.0)
        assert_bit_equal(sg.par_cast(cg, nx, ny, ns), so.par_cast(cg, nx, ny, ns), "%dx%dx%d" % (nx, ny, ns))


@pytest.mark.parametrize("scaling,mode", [("strong", "reduce"), ("weak", "reduce"), ("strong", "gather")])
def test_bench_multi_rank_path_on_one_gpu(tmp_path, scaling, mode):
    """bench.py's N>1 leg (tile sharding by rank through rtiow_rust_amd.parallel.ShardedFrame + framebuffer reduce +
    max-over-ranks timing), run as 2 ranks that share the single GPU of this box (RTG_BENCH_BACKEND=gloo test hook:
    RCCL refuses two ranks per device).  --verify makes rank 0 compare the reduced frame with an unsharded render.
    Default = strong scaling on the fixed 500-spp frame of configs[2]; --scaling weak = 50 spp per GPU."""
    import json
    import subprocess
    import sys
    from conftest import retry_on_busy_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTG_BENCH_BACKEND="gloo")
    runs = []

    def run(port):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
               "--gpus", "2", "--steps", "2", "--warmup", "1", "--nx", "320", "--ny", "192", "--verify", "--no-cpu-baseline",
               "--scaling", scaling, "--reduce-mode", mode]
        runs.append(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600))
        return runs[-1].returncode, runs[-1].stderr
    retry_on_busy_port(run)
    out = runs[-1]
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["verified_bit_exact_vs_unsharded"] is True
    assert line["config"]["workload"].endswith("x500spp" if scaling == "strong" else "x100spp") and line["scaling"] == scaling
    assert line["roofline"]["bound"] == "valu" and line["value"] > 0
    if line["roofline"]["frac"] is not None:
        assert 0.0 < line["roofline"]["frac"] <= 1.0
    # the N > 1 line explains itself: every rank's render kernel and samples, and the reduce as each rank saw it
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert all(r["kernel_ms_avg"] > 0 and r["reduce_ms_avg"] >= 0 and r["samples"] > 0 for r in line["per_rank"])
    assert sum(r["samples"] for r in line["per_rank"]) == 320 * 192 * (500 if scaling == "strong" else 100)
    assert line["reduce_ms_avg"] == line["per_rank"][0]["reduce_ms_avg"]
    # which collective assembled the frame, and what a rank handed to it: the whole zero-padded frame, or its own tiles only
    assert line["reduce_mode"] == mode and line["reduce_bytes_per_rank"] == (320 * 192 * 12 if mode == "reduce" else 320 * 192 * 12 // 2)
    assert line["slowest_rank_kernel_ms"] == max(r["kernel_ms_avg"] for r in line["per_rank"]) <= line["ms_per_step"] * 1.05


def test_bench_gpus_flag_is_self_sufficient(tmp_path, gpu):
    """`python bench.py --gpus N` WITHOUT a launcher starts its own N ranks (torch.distributed.run, one per GPU) and reports
    n_gpus == N; it never reports another N than the one asked for: with fewer GPUs than ranks (RCCL: one rank per device)
    and with a WORLD_SIZE that contradicts --gpus it exits non-zero and prints no JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    small = ["--steps", "1", "--warmup", "0", "--nx", "160", "--ny", "96", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # (a) plain launch, 2 ranks on the one GPU of this box through the gloo test hook
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--verify", "--spp", "20"] + small, env=dict(env, RTG_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["verified_bit_exact_vs_unsharded"] is True
    # (b) more ranks than GPUs over RCCL: refused before anything is rendered
    n_too_many = gpu.device_count() + 7   # (no torch in THIS process: its first import can take minutes on a fresh box)
    out = subprocess.run([sys.executable, bench, "--gpus", str(n_too_many)] + small, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "GPU(s)" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # (c) a launcher that disagrees with --gpus (the silent N = 1 of round 2: `bench.py --gpus 8` run as one process
    # printed n_gpus: 1): WORLD_SIZE = 1 with --gpus 2
    out = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_degenerate_inputs(pkg, gpu, oracle):
    """Empty world (lib.rs:100: every ray misses -> black), ranks that own no tile, 1x1 images, ns = 1."""
    S = pkg.scenes
    b = gpu.builder()
    cam = gpu.camera_look(S.v(0, 0, -5), S.v(0, 0, 0), S.v(0, 1, 0), 40.0, 1.0, 0.0, 10.0)
    img = b.scene([]).par_cast(cam, 20, 12, 3)
    assert img.shape == (12, 20, 3) and (img == 0).all()
    # more ranks than tiles: ranks 2.. own nothing and must leave the canvas untouched
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", 24, 16)     # 2 tiles of 16x16
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", 24, 16)
    ref = so.par_cast(cam_o, 24, 16, 2)
    acc = np.zeros_like(ref)
    for r in range(5):
        canvas = np.full_like(ref, 0.0)
        acc += sg.par_cast(cam_g, 24, 16, 2, rank=r, nranks=5, out=canvas)
    assert_bit_equal(acc, ref, "5 ranks, 2 tiles")
    for name in ("book1", "cornell", "book2"):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, 1, 1)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, 1, 1)
        assert_bit_equal(sg.par_cast(cam_g, 1, 1, 1), so.par_cast(cam_o, 1, 1, 1), name + " 1x1x1")


def test_output_stage_tonemap(pkg, gpu, oracle):
    """print_ppm's quantisation as a GPU kernel, byte-exact against the oracle."""
    rs = np.random.RandomState(5)
    bits32 = rs.randint(0, 2 ** 32, size=1 << 18, dtype=np.uint64).astype(np.uint32).view(np.float32)  # every class
    unit = (rs.rand(1 << 18) * 1.5).astype(np.float32)
    for x in (bits32, unit):
        assert np.array_equal(gpu.tonemap(x), oracle.tonemap(x))
    sg, cam, nx, ny, ns = build_case(pkg, gpu, "cornell")
    img = sg.par_cast(cam, nx, ny, ns)
    assert np.array_equal(gpu.tonemap(img), pkg.ppm.to_u8(img).astype(np.uint8))


@pytest.mark.gpu
def test_bvh4_mode_renders_the_same_image(pkg, gpu, oracle):
    """Non-parity traversal mode: the reference Bvh collapsed to 4-wide nodes.  Boxes are tested earlier
    (against a `best` that is no smaller than the reference's), leaves in the reference's order: same closest hits, same
    image, same rays / shaded hits / draws -- other Aabb / sphere test counts, fewer dependent steps."""
    nx, ny, ns = 120, 80, 8
    bo = oracle.builder()
    wo, cam_o, _ = pkg.scenes.random_scene(bo, nx, ny)
    img_o, st_o = bo.scene(wo).par_cast(cam_o, nx, ny, ns, stats=True)
    bg = gpu.builder()
    wg, cam_g, _ = pkg.scenes.random_scene(bg, nx, ny)
    sg = bg.scene(wg)
    sg.set_option("bvh4", 1)
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    assert_bit_equal(img_g, img_o, "bvh4, instrumented variant")
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), img_o, "bvh4, timed variant")
    for k in ("shaded_hits", "rays", "draws"):
        assert st_g[k] == st_o[k], (k, st_g[k], st_o[k])
    assert st_g["prim_tests"] >= st_o["prim_tests"]          # a superset of the reference's leaves
    # a scene that is not ONE Bvh of spheres has no 4-wide image: the option is refused, nothing is rendered differently
    b2 = gpu.builder()
    w2, _, _ = pkg.scenes.cornell_box_scene(b2, 30, 30)
    with pytest.raises(pkg.capi.RtError):
        b2.scene(w2).set_option("bvh4", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("small_frames", [0, 1])
def test_small_frame_launch_geometry_never_changes_a_bit(pkg, gpu, oracle, small_frames, capfd):
    """rtg_launch.inc pool_geometry: a frame with fewer work items than the chip has lanes runs as small workgroups with one
    work item per lane (reservations of 64 instead of 256); option small_frames = 0 keeps one geometry for every size.  Either
    way the bits are the oracle's -- the reference's own benchmark frame (benches/scene.rs: 10 x 10 x 4) included -- on the
    full-feature pool kernel (bench), the lean one (book1) and the lock-step one (cornell), production and instrumented variant."""
    for name, nx, ny, ns in (("bench", 10, 10, 4), ("bench", 40, 24, 3), ("book1", 10, 10, 4), ("book1", 100, 60, 2),
                             ("cornell", 24, 24, 5), ("book2", 16, 16, 6)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        sg.set_option("small_frames", small_frames)
        sg.set_option("verbose", 1)
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
        capfd.readouterr()
        img_p = sg.par_cast(cam_g, nx, ny, ns)
        err = capfd.readouterr().err
        sg.set_option("verbose", 0)
        if name in ("bench", "book1", "book2"):   # the pool kernels report their launch geometry
            assert ("x 1024 threads" in err) == (small_frames == 0), (name, nx, ny, err)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
        assert_bit_equal(img_p, img_o, "%s %dx%dx%d production, small_frames %d" % (name, nx, ny, ns, small_frames))
        assert_bit_equal(img_g, img_o, "%s %dx%dx%d instrumented, small_frames %d" % (name, nx, ny, ns, small_frames))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, small_frames, k)


@pytest.mark.gpu
@pytest.mark.parametrize("share", [0, 1])
def test_drain_work_sharing_never_changes_a_bit(pkg, gpu, oracle, share):
    """rt_pool_full.h RT_DRAIN_SHARE: once the work queue is empty, waves that have run dry adopt rays from waves that still
    hold some (a path's RNG streams are keyed by pixel, sample and event: which wave runs it is invisible).  Frames large
    enough for every wave of a workgroup to hold paths when the queue runs dry, with the deep bounce cap that makes the
    drain long; sharing on and off, shards included."""
    for name, nx, ny, ns, mb in (("book2", 96, 96, 24, 50), ("book2_bvh", 64, 64, 16, 50), ("volume_bvh", 96, 64, 12, 50),
                                 ("bench", 128, 96, 10, 12)):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        sg.set_option("drain_share", share)
        sg.set_option("small_frames", 0)   # 16-wave workgroups: sharing is between the waves of a workgroup
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True, max_bounces=mb)
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True, max_bounces=mb)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, max_bounces=mb), img_o, "%s production, sharing %d" % (name, share))
        assert_bit_equal(img_g, img_o, "%s instrumented, sharing %d" % (name, share))
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, share, k)
        assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, rank=2, nranks=3, max_bounces=mb),
                         so.par_cast(cam_o, nx, ny, ns, rank=2, nranks=3, max_bounces=mb), "%s shard, sharing %d" % (name, share))


@pytest.mark.gpu
def test_frames_of_different_lds_needs_alternate_on_one_kernel(pkg, gpu, oracle):
    """The dynamic-LDS cap of a kernel belongs to the function, not to a scene handle or a frame size, and the last
    hipFuncSetAttribute wins: a small frame (a 64-thread workgroup asks the lean kernel for half the LDS of a 1024-thread one)
    between two large ones, and a second scene on the same kernel, must not lower it under the next large launch."""
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", 512, 384)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", 512, 384)
    sg2, cam_g2, _, _, _ = build_case(pkg, gpu, "book1_sah", 16, 16)
    so2, cam_o2, _, _, _ = build_case(pkg, oracle, "book1_sah", 16, 16)
    big = so.par_cast(cam_o, 512, 384, 3)     # 589 824 work items: every lane busy, 1024-thread workgroups
    cs_g = gpu.camera_look(pkg.scenes.v(13, 2, 3), pkg.scenes.v(0, 0, 0), pkg.scenes.v(0, 1, 0), 20.0, 1.0, 0.1, 10.0)
    small = so.par_cast(cs_g, 16, 16, 2)      # 512 work items: 64-thread workgroups
    for _ in range(2):
        assert_bit_equal(sg.par_cast(cam_g, 512, 384, 3), big, "large frame")
        assert_bit_equal(sg.par_cast(cs_g, 16, 16, 2), small, "small frame on the same handle")
        assert_bit_equal(sg2.par_cast(cam_g2, 16, 16, 2), so2.par_cast(cam_o2, 16, 16, 2), "another scene on the same kernel")


@pytest.mark.gpu
def test_tiles_of_8_pixels_shard_like_tiles_of_16(pkg, gpu, oracle):
    """rtg_params.tile_w / tile_h may be any multiples of 8 (8x16 tiles balance 8 ranks better than 16x16: parallel.shard_tile).
    A rank's tiles then need not add up to whole 256-item reservations (the work-item count is padded; the padding lies outside
    the image) and a 16x16 block of the baseline kernel spans several tiles.  Every kernel, ragged image sizes, shard by shard
    against the oracle, and the shards sum to the unsharded frame."""
    for name, nx, ny, ns, env in (("book1", 100, 60, 5, {}), ("book2", 72, 40, 4, {}), ("cornell", 50, 50, 6, {}),
                                  ("book1", 90, 52, 3, {"kernel": 1}), ("bench", 44, 36, 7, {})):
        sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
        so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
        for k, v in env.items():
            sg.set_option(k, v)
        whole = so.par_cast(cam_o, nx, ny, ns)
        for tw, th, n in ((8, 8, 3), (8, 16, 8), (24, 8, 5)):
            acc = np.zeros_like(whole)
            for r in range(n):
                kw = {"tile_w": tw, "tile_h": th, "rank": r, "nranks": n}
                part = sg.par_cast(cam_g, nx, ny, ns, **kw)
                assert_bit_equal(part, so.par_cast(cam_o, nx, ny, ns, **kw), "%s %s" % (name, kw))
                acc += part
            assert_bit_equal(acc, whole, "%s: %d shards of %dx%d tiles summed" % (name, n, tw, th))
        img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True, tile_w=8, tile_h=16, rank=1, nranks=2)
        img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True, tile_w=8, tile_h=16, rank=1, nranks=2)
        assert_bit_equal(img_g, img_o, name + " instrumented, 8x16 tiles")
        for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
            assert st_g[k] == st_o[k], (name, k)


def test_sah_tree_renders_the_reference_tree_frame_at_c2(pkg, gpu):
    """bench.py's `also.book1_..._sah_tree` anchor renders configs[1]'s frame through a NON-reference Bvh (the SAH builder,
    SURVEY.md 8 f2).  At C2's full size: the SAH-tree frame equals the reference-tree frame bit for bit on the GPU, and both
    equal the oracle's committed digest of C2 (tests/golden/config_hashes.json); rays / shaded hits / draws are the same,
    only the Aabb::hit count differs (that is the point of the tree)."""
    import hashlib
    import json
    nx, ny, ns = 1200, 800, 50
    ref_scene, cam, _, _, _ = build_case(pkg, gpu, "book1", nx, ny)
    sah_scene, cam2, _, _, _ = build_case(pkg, gpu, "book1_sah", nx, ny)
    a, st_a = ref_scene.par_cast(cam, nx, ny, ns, stats=True)
    b, st_b = sah_scene.par_cast(cam2, nx, ny, ns, stats=True)
    assert_bit_equal(a, b, "SAH tree vs reference tree at C2")
    assert_bit_equal(sah_scene.par_cast(cam2, nx, ny, ns), a, "SAH tree, timed variant")
    for k in ("samples", "rays", "shaded_hits", "draws"):
        assert st_a[k] == st_b[k], (k, st_a[k], st_b[k])
    assert st_b["aabb_tests"] < 0.8 * st_a["aabb_tests"]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_hashes.json")) as f:
        want = json.load(f)["C2_book1_1200x800x50"]["frame"]
    assert hashlib.sha256(np.ascontiguousarray(b, dtype=np.float32).tobytes()).hexdigest() == want   # (no NaN in this frame)


def test_bench_default_line_carries_the_anchors_rooflines(gpu):
    """The driver's N = 1 line: `also` holds C3's frame, C4 and the SAH-tree frame; the two reference-tree anchors carry
    their own roofline objects (counters collected AT those configs, null + reason when the profiles belong to another build)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    also = line["also"]
    assert set(also) == {"book1_random_spheres_1200x800x500spp", "book2_final_scene_800x800x1000spp", "book1_random_spheres_1200x800x50spp_sah_tree"}
    for k in ("book1_random_spheres_1200x800x500spp", "book2_final_scene_800x800x1000spp"):
        r = also[k]["roofline"]
        assert r["bound"] == "valu" and "frac" in r and "traffic" in r and "pmc_source" in r and "lane_utilization" in r
        if r["frac"] is not None:
            assert 0.0 < r["frac"] < 1.0 and 0.0 < r["valu_issue"]["frac"] < 1.0 and r["extrapolated"] is False
        else:
            assert r.get("stale_profile") or r["pmc_source"] is None
    sah = also["book1_random_spheres_1200x800x50spp_sah_tree"]
    assert "non-reference" in sah["note"] and sah["value"] > line["value"] * 0.9
    # the fraction of USEFUL work: on the headline and on every anchor; the SAH tree's with the reference walk's counters,
    # so it exceeds the headline's exactly when the same image takes less time
    cpl = line["roofline"]["counters_per_launch"]
    for r in [line["roofline"]] + [also[k]["roofline"] for k in also]:
        a = r["algorithmic_valu"]
        assert r["valu_lane_utilisation"] == r["frac"] and set(a) >= {"frac", "achieved", "peak", "unit", "definition"}
        if a["frac"] is not None:
            assert 0.0 < a["frac"] < 1.0 and a["c_box"] > 0 and a["lane_instructions_per_launch"] > 0
        else:
            assert a.get("stale_costs")
    a0, a1 = line["roofline"]["algorithmic_valu"], sah["roofline"]["algorithmic_valu"]
    if a0["frac"] is not None:
        assert a1["lane_instructions_per_launch"] == a0["lane_instructions_per_launch"] and sah["roofline"]["counters_per_launch"]["aabb_tests"] < cpl["aabb_tests"]
        assert (a1["frac"] > a0["frac"]) == (sah["kernel_ms_avg"] < line["roofline"]["kernel_ms_avg"])
