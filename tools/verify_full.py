#!/usr/bin/env python3
"""Whole-frame, full-size comparison of the HIP path with the CPU oracle (GPU box; minutes of CPU time).
usage: verify_full.py [case nx ny ns]...   default: C2 (book1 1200x800x50), C1, C4 at 100 spp (list and Bvh world)"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); oracle = g.load_oracle()
cases = [("book1", 1200, 800, 50), ("cornell", 300, 300, 100), ("book2", 800, 800, 100), ("book2_bvh", 800, 800, 100)]
if len(sys.argv) > 1:
    a = sys.argv[1:]
    cases = [(a[i], int(a[i + 1]), int(a[i + 2]), int(a[i + 3])) for i in range(0, len(a), 4)]
for name, nx, ny, ns in cases:
    sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)   # the instrumented variant (counters)
    img_p = sg.par_cast(cam_g, nx, ny, ns)                     # the production variant (what bench.py times)
    t0 = time.perf_counter(); img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True); dt = time.perf_counter() - t0
    diff = int((img_g.view(np.uint32) != img_o.view(np.uint32)).sum()) + int((img_p.view(np.uint32) != img_o.view(np.uint32)).sum())
    same_counts = all(st_g[k] == st_o[k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"))
    print("%-10s %dx%dx%d: %d of %d channel values differ (production + instrumented variant); counters N/P/H/rays/draws equal: %s; crc %08x; oracle %.1f s (%.1f Msamples/s)" % (
        name, nx, ny, ns, diff, img_g.size, same_counts, zlib.crc32(img_g.tobytes()), dt, nx * ny * ns / dt / 1e6), flush=True)
