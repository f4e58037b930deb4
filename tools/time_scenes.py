#!/usr/bin/env python3
"""Time par_cast on the non-north-star configs (GPU box).  usage: time_scenes.py [case nx ny ns]..."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load()
cases = [("cornell", 300, 300, 100), ("book2", 800, 800, 100), ("book2_bvh", 800, 800, 100), ("volume", 300, 300, 100),
         ("simple_light", 300, 300, 20), ("book1", 1200, 800, 50)]
if len(sys.argv) > 1:
    a = sys.argv[1:]
    cases = [(a[i], int(a[i + 1]), int(a[i + 2]), int(a[i + 3])) for i in range(0, len(a), 4)]
for name, nx, ny, ns in cases:
    sc, cam, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    sc.par_cast(cam, nx, ny, 1)
    img, st = sc.par_cast(cam, nx, ny, ns, stats=True)   # instrumented (for counters)
    import ctypes, numpy as np
    ts = []
    for _ in range(5):   # short launches are bimodal on a cold GPU (clock ramp): report the best of 5
        t0 = time.perf_counter(); img = sc.par_cast(cam, nx, ny, ns); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    import zlib
    print("%-13s %4dx%-4d x%-4d  %8.1f ms wall  %8.1f Msamples/s   rays/sample %.2f  box/ray %.1f prim/ray %.1f  crc %08x" % (
        name, nx, ny, ns, dt * 1e3, nx * ny * ns / dt / 1e6, st["rays"] / st["samples"], st["aabb_tests"] / st["rays"],
        st["prim_tests"] / st["rays"], zlib.crc32(img.tobytes())))
