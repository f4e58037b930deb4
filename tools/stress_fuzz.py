#!/usr/bin/env python3
"""More random object graphs than the test-suite holds (GPU box): GPU (instrumented and timed variant) against the oracle, bits and
counters.  usage: stress_fuzz.py <first seed> <last seed>   [FZ_NX / FZ_NY / FZ_NS = frame size and samples; FZ_DEEP=1: every
third graph is drawn with the shapes only the general walk handles (FEAT_DEEP)]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from fuzz_scenes import random_camera, random_world
pkg = g.load_package(); gpu = pkg.load(); ora = g.load_oracle()
bad = 0
import time
t0 = time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    gb = bool(seed % 4 == 0)
    deep = os.environ.get("FZ_DEEP") == "1" and seed % 3 == 0
    imgs = []
    for be in (gpu, ora):
        rs = np.random.RandomState(seed)
        b = be.builder()
        w = random_world(pkg, b, rs, general_boundaries=gb or deep, deep_shapes=deep)
        cam = random_camera(pkg, be, rs, int(os.environ.get("FZ_NX", "64")), int(os.environ.get("FZ_NY", "40")))
        sc = b.scene(w)
        if be is gpu:
            # both schedules of the full-feature path: pool kernel (sync 0) and lock-step kernel (sync 1); lean / FEAT_DEEP programs ignore the option
            runs = []
            for sync in (0, 1):
                sc.set_option("sync", sync)
                img, st = sc.par_cast(cam, int(os.environ.get("FZ_NX", "64")), int(os.environ.get("FZ_NY", "40")), int(os.environ.get("FZ_NS", "12")), stats=True)
                img2 = sc.par_cast(cam, int(os.environ.get("FZ_NX", "64")), int(os.environ.get("FZ_NY", "40")), int(os.environ.get("FZ_NS", "12")))
                runs.append((img, st, img2))
            if not (np.array_equal(runs[0][0].view(np.uint32), runs[1][0].view(np.uint32)) and np.array_equal(runs[0][2].view(np.uint32), runs[1][2].view(np.uint32))
                    and all(runs[0][1][k] == runs[1][1][k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"))):
                bad += 1
                print("MISMATCH between the two schedules, seed", seed)
            imgs.append(runs[0])
        else:
            img, st = sc.par_cast(cam, int(os.environ.get("FZ_NX", "64")), int(os.environ.get("FZ_NY", "40")), int(os.environ.get("FZ_NS", "12")), stats=True)
            imgs.append((img, st))
    (ig, sg, ig2), (io, so) = imgs
    ok = np.array_equal(ig.view(np.uint32), io.view(np.uint32)) and np.array_equal(ig2.view(np.uint32), io.view(np.uint32)) and all(sg[k] == so[k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, gb)
print("seeds %s..%s: %d mismatches, %.0f s" % (sys.argv[1], sys.argv[2], bad, time.time() - t0))
