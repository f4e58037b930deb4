#!/usr/bin/env python3
"""Which samples make the end-of-frame tail (GPU box): per-sample trace of the production kernel on a small frame --
bounces, draws, Aabb tests, primitive tests of the deepest paths.  usage: probe_deep.py [case nx ny ns]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load()
case, nx, ny, ns = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("book2", 64, 64, 16)
sc, cam, _, _, _ = build_case(pkg, gpu, case, nx, ny)
xs, ys, ss = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(ns), indexing="ij")
rgb, info = sc.debug_samples(cam, nx, ny, ns, xs.ravel(), ys.ravel(), ss.ravel(), trace_kernel=True)
b = info[:, 0].astype(np.int64)
print("%s %dx%dx%d: %d samples, rays/sample %.2f; bounce histogram (bounces: samples):" % (case, nx, ny, ns, b.size, (b + 1).mean()))
h = np.bincount(b, minlength=51)
print("  " + "  ".join("%d:%d" % (i, h[i]) for i in range(51) if h[i]))
for k in np.argsort(-b)[:8]:
    print("  deepest: pixel (%d, %d) sample %d: %d bounces, %d draws, %d Aabb tests (%.1f per ray), %d primitive tests (%.1f per ray)" % (
        xs.ravel()[k], ys.ravel()[k], ss.ravel()[k], b[k], info[k, 1], info[k, 2], info[k, 2] / (b[k] + 1.0), info[k, 3], info[k, 3] / (b[k] + 1.0)))
print("  all: Aabb tests per ray %.1f, primitive tests per ray %.1f" % (info[:, 2].sum() / (b + 1).sum(), info[:, 3].sum() / (b + 1).sum()))
