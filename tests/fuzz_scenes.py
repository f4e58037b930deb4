"""Random object graphs built through the public builder API -- exercised identically on the oracle and
on the HIP product.  Every reference constructor can appear: Sphere, Rect, FlipNormals, Translate, Scale,
RotateY, And, rect_prism, LinearMove, ConstantMedium, nested Bvh, list or Bvh world; all five materials;
constant / checker / Perlin textures.  By default the shapes only the general walk handles are avoided (at most 4 nested
non-fused wrappers, no medium below And below Bvh, no medium inside a medium's boundary), so that these scenes run on the
scheduled kernels.  `general_boundaries=True` additionally draws ConstantMedium boundaries that are object graphs (prisms,
And, Bvh, transform wrappers) instead of one primitive; `deep_shapes=True` draws exactly the avoided shapes as well (up to 7
nested wrappers, media inside boundary graphs, media anywhere below And below Bvh): FEAT_DEEP programs, baseline kernel."""
import numpy as np


def random_world(pkg, b, rs, n_top=6, general_boundaries=False, deep_shapes=False):
    S = pkg.scenes
    b.set_perlin_tables(*pkg.small_rng.perlin_tables(int(rs.randint(1, 1 << 30))))

    def f(lo, hi):
        return float(np.float32(rs.uniform(lo, hi)))

    def vec(lo, hi):
        return S.v(f(lo, hi), f(lo, hi), f(lo, hi))

    def texture(depth=0):
        k = rs.randint(0, 4 if depth < 2 else 2)
        if k <= 1:
            return b.constant(vec(0.1, 0.95))
        if k == 2:
            return b.perlin(f(0.02, 0.3))
        return b.checker(texture(depth + 1), texture(depth + 1))

    def material(allow_light=True):
        k = rs.randint(0, 5 if allow_light else 4)
        if k == 0:
            return b.lambertian(texture())
        if k == 1:
            return b.metal(vec(0.3, 0.95), f(0.0, 1.0))
        if k == 2:
            return b.dielectric(f(1.1, 2.0))
        if k == 3:
            return b.isotropic(texture())
        return b.diffuse_light(texture(), f(0.5, 6.0))

    def primitive():
        if rs.rand() < 0.6:
            return b.sphere(f(20, 90), material())
        a0, b0 = f(-120, 0), f(-120, 0)
        return b.rect(int(rs.randint(0, 3)), (a0, a0 + f(40, 220)), (b0, b0 + f(40, 220)), f(-60, 60), material())

    max_wrappers = 6 if deep_shapes else 3

    def graph_boundary(level=0):
        """object.rs:441 `ConstantMedium<O: Object>`: any object can bound a medium (main.rs only uses spheres)."""
        def solid():
            if deep_shapes and level < 2 and rs.rand() < 0.3:   # a medium inside the boundary of a medium
                return b.constant_medium(graph_boundary(level + 1), f(0.002, 0.05), b.isotropic(texture()))
            if rs.rand() < 0.5:
                p0 = vec(-80, 0)
                return b.rect_prism(p0, p0 + S.v(f(60, 200), f(60, 200), f(60, 200)), material())
            return b.translate(vec(-60, 60), b.sphere(f(40, 120), material()))
        k = rs.randint(0, 5)
        if k == 0:
            return solid()
        if k == 1:
            return b.and_(solid(), solid())
        if k == 2:
            return b.bvh([solid() for _ in range(rs.randint(1, 5))], (0.0, 1.0))
        if k == 3:
            return b.rotate_y(f(-170, 170), solid())
        return b.linear_move(b.scale(S.v(f(0.5, 2), f(0.5, 2), f(0.5, 2)), solid()), vec(-40, 40))

    def obj(depth, wrappers, under_bvh, in_and_under_bvh):
        if deep_shapes and wrappers == 0 and rs.rand() < 0.1:   # 5-7 wrappers around one object (object.rs:241-512, any order)
            o = obj(depth + 1, 7, under_bvh, in_and_under_bvh)
            for _ in range(rs.randint(5, 8)):
                w = rs.randint(0, 5)
                o = (b.scale(S.v(f(0.7, 1.5), f(0.7, 1.5), f(0.7, 1.5)), o) if w == 0 else b.linear_move(o, vec(-20, 20)) if w == 1 else
                     b.flip_normals(o) if w == 2 else b.rotate_y(f(-170, 170), o) if w == 3 else b.translate(vec(-60, 60), o))
            return o
        k = rs.randint(0, 10)
        if depth >= (5 if deep_shapes else 3) or k <= 1:
            o = primitive()
        elif k == 2:
            p0 = vec(-80, 0)
            o = b.rect_prism(p0, p0 + S.v(f(30, 120), f(30, 120), f(30, 120)), material())
        elif k == 3:
            o = b.and_(obj(depth + 1, wrappers, under_bvh, under_bvh), obj(depth + 1, wrappers, under_bvh, under_bvh))
        elif k == 4 and wrappers < max_wrappers:
            o = b.translate(vec(-150, 150), obj(depth + 1, wrappers + 1, under_bvh, in_and_under_bvh))
        elif k == 5 and wrappers < max_wrappers:
            o = b.rotate_y(f(-170, 170), obj(depth + 1, wrappers + 1, under_bvh, in_and_under_bvh))
        elif k == 6 and wrappers < max_wrappers:
            w = rs.randint(0, 3)
            inner = obj(depth + 1, wrappers + 1, under_bvh, in_and_under_bvh)
            o = (b.scale(S.v(f(0.5, 2), f(0.5, 2), f(0.5, 2)), inner) if w == 0 else
                 b.linear_move(inner, vec(-40, 40)) if w == 1 else b.flip_normals(inner))
        elif k == 7 and (deep_shapes or not in_and_under_bvh):
            boundary = graph_boundary() if general_boundaries else b.sphere(f(40, 160), material())
            if rs.rand() < 0.5:
                boundary = b.translate(vec(-100, 100), boundary)
            if rs.rand() < 0.3:
                boundary = b.flip_normals(boundary)
            o = b.constant_medium(boundary, f(0.002, 0.05), b.isotropic(texture()))
        elif k == 8:
            o = b.bvh([obj(depth + 1, wrappers, True, False) for _ in range(rs.randint(1, 6))], (0.0, 1.0))
        else:
            o = b.translate(vec(-200, 200), b.sphere(f(10, 60), material()))
        return o

    bvh_world = rs.rand() < 0.4  # lib.rs:51 `impl World for Bvh`: then every top-level object sits below a Bvh
    world = [obj(0, 0, bvh_world, False) for _ in range(n_top)]
    world.append(b.flip_normals(b.sphere(3000.0, b.diffuse_light(b.constant(S.v(0.6, 0.7, 0.9)), 1.0))))
    if bvh_world:
        world = [b.bvh(world, (0.0, 1.0))]
    return world


def random_camera(pkg, be, rs, nx, ny):
    S = pkg.scenes
    frm = S.v(float(rs.uniform(-500, 500)), float(rs.uniform(-100, 400)), float(rs.uniform(-900, -400)))
    return be.camera_look(frm, S.v(0, 0, 0), S.v(0, 1, 0), float(rs.uniform(20, 60)), nx / ny,
                          float(rs.choice([0.0, 5.0, 30.0])), float(rs.uniform(300, 900)),
                          (0.0, float(rs.choice([1.0, 0.25]))))
