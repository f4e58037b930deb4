#!/usr/bin/env python3
"""Determinism stress of the drain-phase work sharing and the small-frame launch geometry (GPU box): many launches of drain-heavy
frames of several sizes on the full-feature pool kernel, every frame's bits against the first render of its configuration
(sharing off), i.e. any race in the hand-over shows as a differing checksum or a hang.  usage: stress_drain.py [rounds]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfgs = [("book2", 96, 96, 24), ("book2", 300, 300, 20), ("book2", 48, 40, 60), ("book2_bvh", 128, 96, 16), ("volume_bvh", 160, 120, 12),
        ("bench", 200, 150, 10), ("bench", 10, 10, 4), ("cornell_smoke", 64, 64, 10)]
bad = 0
for name, nx, ny, ns in cfgs:
    sc, cam, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    sc.set_option("drain_share", 0)
    ref = zlib.crc32(sc.par_cast(cam, nx, ny, ns).tobytes())
    sc.set_option("drain_share", 1)
    n_bad = 0
    for r in range(rounds):
        sc.set_option("small_frames", r & 1)
        if zlib.crc32(sc.par_cast(cam, nx, ny, ns).tobytes()) != ref:
            n_bad += 1
    bad += n_bad
    print("%-14s %4dx%-4d x%-3d: %d launches, %d differ from the frame rendered without sharing" % (name, nx, ny, ns, rounds, n_bad), flush=True)
print("stress_drain:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
