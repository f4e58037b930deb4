#!/usr/bin/env python3
"""A/B of the two full-feature pool kernels on book-2 and its parts (GPU box): every scene is rendered by render_full_pool
(option pool2 = 0) and by render_full_pool2 (pool2 = 1) -- the frames must be bit-equal and the counters equal -- and timed on both,
interleaved.  `floor + light` is the stage-A gate of VERDICT r5 #1 (a Bvh-only scene): ps per Aabb::hit call.
usage: probe_pool2.py [nx ny ns] [--only substring]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
gpu = pkg.load()
S = pkg.scenes
args = [a for a in sys.argv[1:] if not a.startswith("--")]
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
if "--only" in sys.argv:
    args = [a for a in args if a != only]
nx, ny, ns = (int(a) for a in args[:3]) if len(args) >= 3 else (800, 800, 100)
VARIANTS = [
    ("floor + light", [0, 1]),
    ("floor + light + cube", [0, 1, 9]),
    ("without the two media", [0, 1, 2, 3, 4, 5, 8, 9]),
    ("all ten objects", range(10)),
]
print("%-26s %9s %9s %7s %9s %9s %8s %8s  %s" % ("book-2 %dx%dx%d" % (nx, ny, ns), "pool ms", "pool2 ms", "ratio", "ps/box 1", "ps/box 2", "box/ray", "Mrays", "bits / counters"))
for name, keep in VARIANTS:
    if only and only not in name:
        continue
    b = gpu.builder()
    world, cam, _ = S.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF))
    sc = b.scene([world[i] for i in keep])
    res = {}
    for v in (0, 1):
        sc.set_option("pool2", 2 * v)
        sc.par_cast(cam, nx, ny, 1)
        if "--verbose" in sys.argv:
            print("-- %s, pool2 = %d" % (name, v), file=sys.stderr, flush=True)
            sc.set_option("verbose", 1)
        res[v] = sc.par_cast(cam, nx, ny, ns, stats=True)
        sc.set_option("verbose", 0)
    same = np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    keys = ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")
    cnt = all(res[0][1][k] == res[1][1][k] for k in keys)
    ts = {0: [], 1: []}
    for _ in range(4):
        for v in (0, 1):
            sc.set_option("pool2", 2 * v)
            t0 = time.perf_counter()
            sc.par_cast(cam, nx, ny, ns)
            ts[v].append(time.perf_counter() - t0)
    st = res[0][1]
    t0, t1 = min(ts[0]), min(ts[1])
    print("%-26s %9.2f %9.2f %7.3f %9.2f %9.2f %8.1f %8.1f  %s / %s" % (
        name, t0 * 1e3, t1 * 1e3, t1 / t0, t0 * 1e12 / st["aabb_tests"], t1 * 1e12 / st["aabb_tests"], st["aabb_tests"] / st["rays"], st["rays"] / 1e6,
        "equal" if same else "DIFFER (%d px)" % int((res[0][0] != res[1][0]).any(axis=2).sum()),
        "equal" if cnt else "DIFFER " + str({k: (res[0][1][k], res[1][1][k]) for k in keys if res[0][1][k] != res[1][1][k]})), flush=True)
