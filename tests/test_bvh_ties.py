"""Tie-order audit of Bvh::new (bvh.rs:22-81).  The reference sorts with `sort_unstable_by` (bvh.rs:51-57), whose order
of EQUAL keys is rustc-version specific; oracle and product use a stable sort (ties keep input order).
This test puts the error bar on "the counters equal the reference's": it counts how many sorts of the named scenes have
tied keys at all, how many of those ties STRADDLE the median split (only there can the tie order change which leaf goes
left / right, i.e. the tree's shape), and renders the affected scene with the tied runs in random orders -- any outcome an
unstable sort could produce, at every level independently (oracle-only hook rto_builder_bvh_ties).

  * the image must not change (a tree's shape only matters at exact-t ties between two primitives);
  * rays / shaded hits / draws must not change (they follow from the image-relevant hits and the RNG streams);
  * N (Aabb::hit calls) and P (primitive tests) may: the measured spread is asserted against the bound DESIGN.md states.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bit_equal

# |N(tie order) - N(stable)| / N(stable), the bound DESIGN.md section 2 states.  Measured over 40 random tie orders
# (tools: this file's _scene with seeds 1..40): book-2 64x64x4 -1.31 % .. +1.37 % (40 distinct trees); book-1 96x64x4
# -3.84 % .. +2.44 % (5 distinct trees: its ONE tie across a median sits in the small subtree where the three objects with the
# x-key 0 meet -- ground sphere, sky dome, the glass sphere at the origin: the two largest objects of the scene change places).
N_BOUND = {"book1": 0.05, "book2": 0.02}


def _ties(oracle, b, seed):
    fn = oracle.lib.rto_builder_bvh_ties
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    assert fn(b.h, seed, out) == 0
    return dict(zip(("sorts", "sorts_with_ties", "tied_keys", "straddling"), [int(x) for x in out]))


def _scene(pkg, oracle, name, nx, ny, tie_seed):
    b = oracle.builder()
    _ties(oracle, b, tie_seed)
    if name == "book2":
        world, cam, _ = pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF))
    else:
        world, cam, _ = pkg.scenes.random_scene(b, nx, ny)
    return b, b.scene(world), cam


def test_which_sorts_tie(pkg, oracle):
    """Measured on the scenes BASELINE.json names (construction seed 0xDEADBEEF):
    book-1 (484 sorts for 485 leaves): 7 sorts see equal keys -- small subtrees whose widest axis is y, where every
    r = 0.2 sphere has the key 0.4, and the nodes above ground sphere, sky dome and the glass sphere at the origin (x-key 0
    each) -- and in ONE of them the tie lies across the median.
    book-2: the cube of 1000 random spheres has no tie at all; the floor (20 x 20 boxes on a 100-unit grid,
    main.rs:194-200: whole rows / columns share a centroid) ties in 223 of its 399 sorts, 144 of them across the median
    (first in the 16 subtrees of 5 x 5 boxes, where `len / 2` = 12 cuts through the third row of five)."""
    b = oracle.builder()
    _ties(oracle, b, 0)
    pkg.scenes.random_scene(b, 60, 40)
    assert _ties(oracle, b, 0) == {"sorts": 484, "sorts_with_ties": 7, "tied_keys": 21, "straddling": 1}
    b = oracle.builder()
    _ties(oracle, b, 0)
    pkg.scenes.book_final_scene(b, 40, 40, pkg.small_rng.SmallRng(0xDEADBEEF))
    assert _ties(oracle, b, 0) == {"sorts": 399 + 999, "sorts_with_ties": 223, "tied_keys": 2976, "straddling": 144}
    # the cube alone: no tie
    b = oracle.builder()
    _ties(oracle, b, 0)
    rng = pkg.small_rng.SmallRng(1)
    S = pkg.scenes
    b.bvh([b.translate(np.float32(165.0) * rng.gen_vec3(), b.sphere(10.0, b.dielectric(1.5))) for _ in range(1000)])
    assert _ties(oracle, b, 0) == {"sorts": 999, "sorts_with_ties": 0, "tied_keys": 0, "straddling": 0}


@pytest.mark.parametrize("name,nx,ny,ns", [("book2", 64, 64, 6), ("book1", 96, 64, 6)])
def test_tie_order_changes_no_pixel_and_bounds_the_counters(pkg, oracle, name, nx, ny, ns):
    _, s0, cam = _scene(pkg, oracle, name, nx, ny, 0)
    ref, st0 = s0.par_cast(cam, nx, ny, ns, stats=True)
    spread = []
    for seed in (1, 2, 3, 0xC0FFEE, 0xDEADBEEF, 77):
        _, s1, cam1 = _scene(pkg, oracle, name, nx, ny, seed)
        img, st = s1.par_cast(cam1, nx, ny, ns, stats=True)
        assert_bit_equal(img, ref, "%s, tie order seed %#x" % (name, seed))
        for k in ("samples", "shaded_hits", "rays", "draws"):
            assert st[k] == st0[k], (seed, k)
        spread.append((st["aabb_tests"] - st0["aabb_tests"]) / st0["aabb_tests"])
        assert abs(st["prim_tests"] - st0["prim_tests"]) / st0["prim_tests"] <= N_BOUND[name]
    assert any(x != 0.0 for x in spread), "the shuffles never changed the tree: the audit hook is not wired"
    assert max(abs(x) for x in spread) <= N_BOUND[name], spread
    print("%s: N spread over tie orders: %s" % (name, ", ".join("%+.4f%%" % (100 * x) for x in spread)))
