"""Traversal-order model on the ORACLE (test infrastructure): what would a near-child-first walk of the
reference's own Bvh cost, and does it ever find another hit?

    python tests/model_traversal_order.py [--spp-scale 1.0] > profiles/r06_experiments/r06a_traversal_order_model.txt

For every Bvh root without a ConstantMedium below it (oracle/rto_scene.hpp OrderModel) the oracle walks the tree a second time,
near child first by the sign of the ray direction along the node's split axis, with the tie-safe rule (enter when
min(best, far) >= start; on equal t the leaf that comes first in the reference's depth-first order wins), and counts: Aabb::hit
calls and primitive tests of both walks, results that differ, and accepted hits whose t lies below the entry distance of their
own leaf box (the roundoff cases in which the orders CAN differ).  CPU only; C2 at its named size takes ~1 min on 8 cores.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def run(pkg, be, name, fn, nx, ny, ns, max_bounces=50):
    b = be.builder()
    world, cam, _ = fn(pkg, b, nx, ny)
    scene = b.scene(world)
    f = be.lib.rto_debug_order_model
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(pkg.capi.Camera), C.POINTER(pkg.capi.Params), C.c_int, C.POINTER(C.c_uint64), C.c_int]
    out = (C.c_uint64 * (9 * 8))()
    p = pkg.capi.make_params(nx, ny, ns, max_bounces=max_bounces)
    t0 = time.time()
    n = f(scene.h, C.byref(cam), C.byref(p), 0, out, 8)
    assert n >= 0, be.last_error()
    dt = time.time() - t0
    rows = np.array(list(out), dtype=np.uint64).reshape(8, 9)[:n]
    print("%s  %dx%dx%d  (%.0f s)" % (name, nx, ny, ns, dt))
    print("  %-8s %12s %14s %14s %7s %14s %14s %7s %8s %12s" % ("leaves", "calls", "N ref", "N near-first", "ratio", "P ref", "P near-first", "ratio",
                                                                  "differ", "below-entry"))
    for r in rows:
        leaves, calls, n_ref, p_ref, n_near, p_near, differ, below, hits = [int(v) for v in r]
        print("  %-8d %12d %14d %14d %7.3f %14d %14d %7.3f %8d %12d" % (leaves, calls, n_ref, n_near, n_near / max(n_ref, 1), p_ref, p_near,
                                                                         p_near / max(p_ref, 1), differ, below))
    sys.stdout.flush()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp-scale", type=float, default=1.0, help="scale the sample counts (1.0 = C2 at its named size)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    pkg = graft.load_package()
    be = graft.load_oracle()
    S = pkg.scenes
    rng = lambda: pkg.small_rng.SmallRng(0xDEADBEEF)  # noqa: E731
    cases = [
        ("book1 (C2: the reference's median-split tree)", lambda pkg, b, nx, ny: S.random_scene(b, nx, ny), 1200, 800, 50),
        ("book1, SAH tree (non-reference)", lambda pkg, b, nx, ny: S.random_scene(b, nx, ny, use_bvh="sah"), 1200, 800, 50),
        ("book2, list world (C4's scene: floor Bvh of 400 prisms, cube Bvh of 1000 spheres)",
         lambda pkg, b, nx, ny: S.book_final_scene(b, nx, ny, rng()), 800, 800, 20),
        ("bench (benches/scene.rs: Cornell + prisms under one Bvh)", lambda pkg, b, nx, ny: S.bench_scene(b, nx, ny), 300, 300, 20),
    ]
    for name, fn, nx, ny, ns in cases:
        if args.only and args.only not in name:
            continue
        run(pkg, be, name, fn, nx, ny, max(1, int(round(ns * args.spp_scale))))


if __name__ == "__main__":
    main()
