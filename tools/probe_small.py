#!/usr/bin/env python3
"""Latency probe (GPU box): tiny frames of a scene at two bounce caps -- what ONE deep path costs when the chip is empty.
usage: probe_small.py [case]   (RTG_VERBOSE=1 prints the instrumented variant's schedule / time shares)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load()
case = sys.argv[1] if len(sys.argv) > 1 else "book2"
for (nx, ny, spp) in ((16, 16, 1), (16, 16, 64), (128, 128, 10)):
    sc, cam, _, _, _ = build_case(pkg, gpu, case, nx, ny)
    sc.par_cast(cam, nx, ny, 1)
    for mb in (50, 4):
        ts = []
        for _ in range(3):
            img, st = sc.par_cast(cam, nx, ny, spp, stats=True, max_bounces=mb)
            ts.append(st["kernel_ms"])
        print("%s %dx%dx%d max_bounces %d: kernel %.2f ms (instrumented variant), %d rays" % (case, nx, ny, spp, mb, min(ts), st["rays"]), flush=True)
