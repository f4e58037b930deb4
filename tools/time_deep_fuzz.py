#!/usr/bin/env python3
"""Kernel time of the 24 deep-shape fuzz graphs (tests/test_fuzz.py: FEAT_DEEP programs, rendered by the baseline kernel's general
walk) at a size that fills the chip, with the walk sized for each graph's real depths (option deep_sized = 1, the default) and with the
largest instantiation (0).  usage (GPU box): tools/time_deep_fuzz.py [nx ny ns]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from test_fuzz import _build, N_DEEP_SCENES
pkg = g.load_package(); gpu = pkg.load()
nx, ny, ns = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 240, 8)
tot = {0: 0.0, 1: 0.0}
kinds = {}
for seed in range(N_DEEP_SCENES):
    b, world, cam = _build(pkg, gpu, 9000 + seed, nx, ny, True, True)
    words, feat = b.flatten(world)
    if not feat & 128:
        continue
    sc = b.scene(world)
    ref = None
    ms = {}
    for sized in (1, 0):
        sc.set_option("deep_sized", sized)
        sc.par_cast(cam, nx, ny, 1)
        best = 1e9
        for _ in range(3):
            img, st = sc.par_cast(cam, nx, ny, ns, stats=True)
            best = min(best, st["kernel_ms"])
        ms[sized] = best
        tot[sized] += best
        if ref is None:
            ref = img
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), seed
    key = ("<= 8 wrappers" if feat & 256 else "<= 32 wrappers") + ", " + ("1 level of boundary queries" if feat & 512 else "3 levels")
    kinds.setdefault(key, [0, 0.0, 0.0])
    kinds[key][0] += 1; kinds[key][1] += ms[1]; kinds[key][2] += ms[0]
    print("deep fuzz %2d  features 0x%03x  sized %.2f ms  largest %.2f ms" % (seed, feat, ms[1], ms[0]))
for k, (n, a, c) in kinds.items():
    print("%-55s %2d graphs: sized %.1f ms, largest instantiation %.1f ms" % (k, n, a, c))
print("all: sized %.1f ms, largest %.1f ms (%dx%dx%d, instrumented variant)" % (tot[1], tot[0], nx, ny, ns))
