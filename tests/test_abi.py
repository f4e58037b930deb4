"""CPU-only checks of the product library: it loads, exports every symbol include/rtiow_gpu.h declares,
the host-side builder/flattener behaves like the reference's constructors (incl. error behaviour), and
compute entry points FAIL LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OP_END, OP_BOX, OP_SPHERE, OP_RECT, OP_PUSH, OP_POP, OP_MEDIUM, OP_PRISM = range(8)
OP_SEG, OP_EXT = 9, 12   # flat_scene.h


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rtiow_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rtg_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(pkg):
    be = pkg.load()
    names = header_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(be.lib, n), "librtiow_gpu.so does not export %s" % n
    # and the binding covers the header
    assert sorted("rtg_" + s for s in pkg.capi.ABI_SYMBOLS) == names


def test_rust_sys_crate_declares_every_header_symbol():
    """No Rust toolchain exists here, so the `-sys` crate cannot be compiled; what can be checked is that it declares
    exactly the header's entry points, each with the header's number of arguments, and the three repr(C) structs with
    the header's field order."""
    rs = open(os.path.join(ROOT, "rtiow-rust_amd", "host", "rust", "rtiow-gpu-sys", "src", "lib.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "rtiow_gpu.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs_nc = re.sub(r"//[^\n]*", "", rs)
    c_fns = {m.group(1): m.group(2) for m in re.finditer(r"\b(rtg_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", hdr_nc)}
    rs_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rtg_[a-z_0-9]+)\s*\(([^)]*)\)", rs_nc, flags=re.S)}
    assert sorted(c_fns) == sorted(rs_fns) == header_symbols()

    def argc(a):
        a = a.strip()
        return 0 if a in ("", "void") else len([x for x in a.split(",") if x.strip()])
    for name in c_fns:
        assert argc(c_fns[name]) == argc(rs_fns[name]), name
    for struct in ("rtg_camera", "rtg_params", "rtg_stats"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr_nc, flags=re.S).group(1)
        c_fields = [f for decl in c_body.split(";") if decl.strip()
                    for f in re.findall(r"([a-z_0-9]+)(?:\[\d+\])?\s*(?:,|$)", decl.strip().split(None, 1)[1])]
        rs_body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs_nc, flags=re.S).group(1)
        rs_fields = re.findall(r"pub ([a-z_0-9]+):", rs_body)
        assert c_fields == rs_fields, (struct, c_fields, rs_fields)


def test_oracle_mirrors_the_abi(pkg, oracle):
    for s in pkg.capi.ABI_SYMBOLS:
        if s in ("device_count", "scene_set_option", "scene_info", "par_cast_device", "debug_flatten", "debug_flatten_pool2", "tonemap_device", "multi_reset"):
            continue  # device plumbing has no CPU counterpart
        assert hasattr(oracle.lib, "rto_" + s), s


def test_struct_layouts_match_header(pkg):
    assert C.sizeof(pkg.Camera) == 21 * 4
    assert C.sizeof(pkg.Params) == 56
    assert C.sizeof(pkg.Stats) == 56


def test_camera_look_matches_oracle_bitwise(pkg, oracle):
    be = pkg.load()
    S = pkg.scenes
    for args in [(S.v(278, 278, -800), S.v(278, 278, 0), S.v(0, 1, 0), 40.0, 1.0, 0.0, 10.0),
                 (S.v(13, 2, 3), S.v(0, 0, 0), S.v(0, 1, 0), 20.0, 1.5, 0.1, 10.0),
                 (S.v(478, 278, -600), S.v(278, 278, 0), S.v(0, 1, 0), 40.0, 1.0, 0.0, 10.0),
                 (S.v(-3, 7, 2), S.v(1, -2, 0.5), S.v(0.1, 1, 0.2), 63.0, 1.7777, 0.3, 4.2)]:
        assert bytes(be.camera_look(*args)) == bytes(oracle.camera_look(*args))


def test_flat_program_of_book1(pkg):
    be = pkg.load()
    b = be.builder()
    world, _, _ = pkg.scenes.random_scene(b, 120, 80)
    words, feat = b.flatten(world)
    ops = words[:, 7] & 0xff
    n_sph = int((ops == OP_SPHERE).sum())
    assert feat == 0                                   # lean kernel: spheres under one Bvh
    assert int((ops == OP_BOX).sum()) == 2 * n_sph - 1  # one leaf per object (bvh.rs:61-65)
    assert ops[-1] == OP_END and ops[0] == OP_BOX
    assert words[0, 7] & (1 << 13)                      # F_BVH_ROOT
    assert words[0, 6] == len(words) - 1                # root skip -> END
    # skip pointers always point forward and never past END
    box = ops == OP_BOX
    assert (words[box, 6] > np.nonzero(box)[0]).all() and (words[box, 6] <= len(words) - 1).all()
    # every Translate{Sphere} fused into one record, the sky dome flipped and un-translated
    sph = words[ops == OP_SPHERE]
    assert int(((sph[:, 7] >> 8) & 1).sum()) == n_sph - 1 and int(((sph[:, 7] >> 9) & 1).sum()) == 1


def test_flat_program_of_cornell_and_book2(pkg):
    be = pkg.load()
    b = be.builder()
    world, _, _ = pkg.scenes.cornell_box_scene(b, 32, 32)
    words, feat = b.flatten(world)
    ops = (words[:, 7] & 0xff).tolist()
    # 6 walls, then 2 x [PUSH (Translate + RotateY in one wrapper level), PRISM, POP]
    assert ops == [OP_RECT] * 6 + [OP_PUSH, OP_PRISM, OP_POP] * 2 + [OP_END]
    assert feat == (1 | 4)
    push = words[6]
    assert (push[7] >> 8) & 7 == 1 and push[7] & (1 << 11)          # XF_ROTATE_Y | F_PRE_TRANSLATE
    assert push[[3, 4, 5]].view(np.float32).tolist() == [130.0, 0.0, 65.0]
    assert words[7, :6].view(np.float32).tolist() == [0.0, 165.0, 0.0, 165.0, 0.0, 165.0]   # (p0.x, p1.x, p0.y, p1.y, p0.z, p1.z)
    # an And-tree that is NOT rect_prism's (here: one face of another material) stays six RECT records,
    # in rect_prism's order (object.rs:420-473): +Z, +Y, +X, then flipped -Z, -Y, -X
    m1, m2 = b.lambertian(b.constant(pkg.scenes.vfrom(0.5))), b.lambertian(b.constant(pkg.scenes.vfrom(0.25)))
    p0, p1 = (0.0, 0.0, 0.0), (1.0, 2.0, 3.0)
    faces = [b.rect(2, (p0[0], p1[0]), (p0[1], p1[1]), p1[2], m1), b.rect(1, (p0[0], p1[0]), (p0[2], p1[2]), p1[1], m1),
             b.rect(0, (p0[1], p1[1]), (p0[2], p1[2]), p1[0], m1),
             b.flip_normals(b.rect(2, (p0[0], p1[0]), (p0[1], p1[1]), p0[2], m1)),
             b.flip_normals(b.rect(1, (p0[0], p1[0]), (p0[2], p1[2]), p0[1], m1)),
             b.flip_normals(b.rect(0, (p0[1], p1[1]), (p0[2], p1[2]), p0[0], m2))]
    odd = b.and_(b.and_(faces[0], b.and_(faces[1], faces[2])), b.and_(faces[3], b.and_(faces[4], faces[5])))
    prism, _ = b.flatten([odd])
    assert (prism[:, 7] & 0xff).tolist() == [OP_RECT] * 6 + [OP_END]
    assert ((prism[:6, 7] >> 10) & 3).tolist() == [2, 1, 0, 2, 1, 0]
    assert ((prism[:6, 7] >> 9) & 1).tolist() == [0, 0, 0, 1, 1, 1]
    same = b.and_(b.and_(faces[0], b.and_(faces[1], faces[2])),
                  b.and_(faces[3], b.and_(faces[4], b.flip_normals(b.rect(0, (p0[1], p1[1]), (p0[2], p1[2]), p0[0], m1)))))
    assert (b.flatten([same])[0][:, 7] & 0xff).tolist() == [OP_PRISM, OP_END]
    b = be.builder()
    world, _, _ = pkg.scenes.book_final_scene(b, 32, 32, pkg.small_rng.SmallRng(0xDEADBEEF))
    words, feat = b.flatten(world)
    ops = words[:, 7] & 0xff
    assert feat == 15 | 64                             # every geometry / texture feature + "an albedo (Perlin) may exceed 1"
    assert int((ops == OP_MEDIUM).sum()) == 2 and int((ops == OP_RECT).sum()) == 1 and int((ops == OP_PRISM).sum()) == 400
    assert int((ops == OP_PUSH).sum()) == 1               # Translate{RotateY{Bvh}}: one wrapper level
    # Translate{LinearMove{Sphere}} (main.rs:218-229) is ONE fused SPHERE record (F_TRANSLATE | F_MOVE) + its motion vector
    mv = np.nonzero((ops == OP_SPHERE) & ((words[:, 7] >> 20) & 1 == 1))[0]
    assert len(mv) == 1 and ops[mv[0] + 1] == OP_EXT and words[mv[0], 7] & (1 << 8)
    assert words[mv[0], :4].view(np.float32).tolist() == [400.0, 400.0, 200.0, 50.0]
    assert words[mv[0] + 1, :3].view(np.float32).tolist() == [30.0, 0.0, 0.0]
    # ... only in the order the reference applies them: LinearMove{Translate{Sphere}} keeps its wrapper (around a fused Translate)
    b2 = be.builder()
    m = b2.lambertian(b2.constant(pkg.scenes.vfrom(0.5)))
    w2, _ = b2.flatten([b2.linear_move(b2.translate(pkg.scenes.v(1, 2, 3), b2.sphere(1.0, m)), pkg.scenes.v(1, 0, 0))])
    assert (w2[:, 7] & 0xff).tolist() == [OP_PUSH, OP_SPHERE, OP_POP, OP_END] and not (w2[1, 7] >> 20) & 1
    w3, _ = b2.flatten([b2.flip_normals(b2.linear_move(b2.sphere(1.0, m), pkg.scenes.v(0, 1, 0)))])
    assert (w3[:, 7] & 0xff).tolist() == [OP_SPHERE, OP_EXT, OP_END] and (w3[0, 7] >> 20) & 1 and (w3[0, 7] >> 9) & 1 and not w3[0, 7] & (1 << 8)
    assert int((ops == OP_BOX).sum()) == (2 * 400 - 1) + (2 * 1000 - 1)
    med = np.nonzero(ops == OP_MEDIUM)[0]
    assert (ops[med + 1] == OP_SPHERE).all()            # boundary record follows its medium
    assert not (words[med, 7] & (1 << 12)).any()        # list world: not under a Bvh


def test_albedo_range_flags_walk_the_texture_tree(pkg):
    """The pool kernels keep no accum field; the flattener must prove every reachable albedo stays in range -- through
    checker children (texture.rs:12-21) and Perlin tables (perlin.rs), not only for top-level constants."""
    be = pkg.load()
    S = pkg.scenes
    WIDE, BRIGHT = 32, 64

    def feat(make):
        b = be.builder()
        return b.flatten([b.sphere(1.0, make(b))])[1] & (WIDE | BRIGHT)

    assert feat(lambda b: b.lambertian(b.constant(S.vfrom(0.5)))) == 0
    assert feat(lambda b: b.lambertian(b.constant(S.vfrom(100.0)))) == WIDE
    assert feat(lambda b: b.lambertian(b.checker(b.constant(S.vfrom(0.2)), b.constant(S.vfrom(0.9))))) == 0
    assert feat(lambda b: b.lambertian(b.checker(b.constant(S.vfrom(100.0)), b.constant(S.vfrom(float("nan")))))) == WIDE
    assert feat(lambda b: b.lambertian(b.checker(b.constant(S.vfrom(7.0)), b.constant(S.vfrom(-0.5))))) == WIDE
    assert feat(lambda b: b.isotropic(b.checker(b.constant(S.vfrom(0.5)), b.checker(b.constant(S.vfrom(0.1)), b.constant(S.vfrom(1.5)))))) == BRIGHT
    assert feat(lambda b: b.diffuse_light(b.checker(b.constant(S.vfrom(100.0)), b.constant(S.vfrom(-3.0))), 15.0)) == 0  # emission: no albedo
    rs = np.random.RandomState(1)
    v = rs.standard_normal((256, 3)).astype(np.float32)
    unit = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    perm = [rs.permutation(256).astype(np.uint8) for _ in range(3)]

    def with_tables(vecs):
        def make(b):
            b.set_perlin_tables(vecs, *perm)
            return b.lambertian(b.checker(b.constant(S.vfrom(0.5)), b.perlin(4.0)))   # Perlin as a CHILD of a checker
        return make
    assert feat(with_tables(unit)) == BRIGHT
    assert feat(with_tables(unit * np.float32(3.0))) == WIDE       # no turbulence bound for over-long gradient vectors


def test_reference_error_behaviour(pkg):
    be = pkg.load()
    S = pkg.scenes
    b = be.builder()
    with pytest.raises(pkg.RtError) as e:               # bvh.rs:60 panic
        b.bvh([])
    assert e.value.code == -1 or "zero objects" in str(e.value)
    assert "zero objects" in be.last_error()
    m = b.lambertian(b.constant(S.vfrom(0.5)))
    with pytest.raises(pkg.RtError):                     # bvh.rs:45/56 partial_cmp().unwrap() on NaN
        b.bvh([b.sphere(float("nan"), m), b.sphere(1.0, m)])
    assert "NaN" in be.last_error()
    with pytest.raises(pkg.RtError):
        b.sphere(1.0, 12345)                             # dangling material handle
    with pytest.raises(pkg.RtError):
        b.translate(S.v(0, 0, 0), 999)
    with pytest.raises(pkg.RtError):
        b.perlin(4.0)                                    # tables not supplied
    with pytest.raises(pkg.RtError):
        b.rect(5, (0, 1), (0, 1), 0.0, m)
    # Every graph the reference's types allow flattens.  Shapes the scheduled kernels do not walk (a medium inside a medium's
    # boundary, a medium below an And below a Bvh, more than 4 nested wrappers) set FEAT_DEEP (128): the general walk of the
    # baseline kernel renders them.  RTG_ERR_UNSUPPORTED is left for nesting beyond that walk's (documented) stack bounds.
    FEAT_BOUNDARY, FEAT_DEEP, OP_BEND, OP_SAVE, OP_MERGE = 16, 128, 8, 10, 11
    iso = b.isotropic(b.constant(S.vfrom(1.0)))
    box = b.rect_prism(S.v(0, 0, 0), S.v(1, 1, 1), m)
    words, feat = b.flatten([b.constant_medium(box, 0.1, iso)])   # a boundary may be any object graph ...
    assert (feat & FEAT_BOUNDARY) and not (feat & FEAT_DEEP) and words[0, 4] == len(words) - 1
    nested = b.constant_medium(b.and_(box, b.constant_medium(b.sphere(1.0, m), 0.1, iso)), 0.1, iso)
    words, feat = b.flatten([nested])                             # ... also one that itself holds a medium
    ops = [int(w[7]) & 0xff for w in words]
    assert (feat & FEAT_DEEP) and ops == [OP_MEDIUM, OP_PRISM, OP_MEDIUM, OP_SPHERE, OP_BEND, OP_END]
    level = b.sphere(1.0, m)
    for _ in range(3):                                            # 3 levels of media in media boundaries; the 4th is refused
        level = b.constant_medium(b.and_(box, level), 0.1, iso)
    b.flatten([level])
    with pytest.raises(pkg.RtError) as e:
        b.flatten([b.constant_medium(b.and_(box, b.constant_medium(b.and_(box, level), 0.1, iso)), 0.1, iso)])
    assert e.value.code == -5
    # a medium below an And below a Bvh: SAVE / MERGE around the And's stream, its media replace like a list's
    smoke = b.constant_medium(b.sphere(2.0, m), 0.1, iso)
    words, feat = b.flatten([b.bvh([b.and_(b.sphere(1.0, m), smoke), b.sphere(3.0, m)])])
    ops = [int(w[7]) & 0xff for w in words]
    assert (feat & FEAT_DEEP) and ops.count(OP_SAVE) == 1 and ops.count(OP_MERGE) == 1
    i_med = ops.index(OP_MEDIUM)
    assert ops.index(OP_SAVE) < i_med < ops.index(OP_MERGE) and not (int(words[i_med, 7]) & (1 << 12))   # no F_UNDER_BVH inside the And
    words, feat = b.flatten([b.bvh([smoke, b.sphere(3.0, m)])])   # directly below the Bvh: the scheduled kernels' compare rule
    assert not (feat & FEAT_DEEP) and (int(words[[int(w[7]) & 0xff for w in words].index(OP_MEDIUM), 7]) & (1 << 12))
    # wrappers: 4 levels for the scheduled kernels, 32 for the general walk
    deep = b.sphere(1.0, m)
    for i in range(33):
        deep = b.scale(S.v(1, 2, 1), deep)
        if i + 1 in (4, 5, 32):
            _, feat = b.flatten([deep])
            assert bool(feat & FEAT_DEEP) == (i + 1 > 4), i
    with pytest.raises(pkg.RtError) as e:
        b.flatten([deep])
    assert e.value.code == -5


def test_no_cpu_fallback(pkg):
    """Without a GPU the compute path must fail loudly (RTG_ERR_DEVICE), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    be = pkg.load()
    b = be.builder()
    world, cam, _ = pkg.scenes.cornell_box_scene(b, 16, 16)
    with pytest.raises(pkg.RtError) as e:
        b.scene(world)
    assert e.value.code == -6 and "no CPU fallback" in str(e.value)
    with pytest.raises(pkg.RtError) as e:
        be.debug_math(0, np.ones(4, dtype=np.float32))
    assert e.value.code == -6


def test_product_does_not_link_or_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkgdir = os.path.join(ROOT, "rtiow-rust_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "rto_" not in txt and "pyoracle" not in txt, os.path.join(dirpath, f)
