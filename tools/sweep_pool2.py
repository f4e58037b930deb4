#!/usr/bin/env python3
"""One-at-a-time sweep of the pool-2 kernel's schedule thresholds (GPU box) on book-2 (or a part of it).
usage: sweep_pool2.py [nx ny ns] [--keep 0,1,...] [--set name=v,...] [--params name:v1,v2,..;name:...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
gpu = pkg.load()
S = pkg.scenes


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


pos = []
skip = False
for a in sys.argv[1:]:
    if skip:
        skip = False
        continue
    if a.startswith("--"):
        skip = True
        continue
    pos.append(a)
nx, ny, ns = (int(a) for a in pos[:3]) if len(pos) >= 3 else (800, 800, 200)
keep = [int(k) for k in opt("--keep", "0,1,2,3,4,5,6,7,8,9").split(",")]
base = dict(kv.split("=") for kv in opt("--set", "").split(",") if kv)
DEFAULT = "p2_refill:12,16,20,24,28,36;p2_box_leave:8,16,24,32,64;p2_park:16,24,32,40,64;p2_sphere:4,8,12,16,24;p2_prism:4,8,12,16,24;p2_list:4,8,12,16,24,32;p2_push:1,2,4,8,16"
params = [(p.split(":")[0], [int(v) for v in p.split(":")[1].split(",")]) for p in opt("--params", DEFAULT).split(";")]
b = gpu.builder()
world, cam, _ = S.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF))
sc = b.scene([world[i] for i in keep])


def timed(reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sc.par_cast(cam, nx, ny, ns)
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


sc.par_cast(cam, nx, ny, 1)
sc.set_option("pool2", 0)
print("book-2 %s %dx%dx%d   first kernel: %.2f ms" % (keep, nx, ny, ns, timed()))
sc.set_option("pool2", 2)
for k, v in base.items():
    sc.set_option(k, int(v))
print("pool 2, base %s: %.2f ms" % (base, timed()), flush=True)
DEFAULTS = dict(p2_refill=24, p2_box_leave=48, p2_park=40, p2_sphere=4, p2_prism=8, p2_list=24, p2_push=2,
                lpt=2, lpt_deep=4, lpt_shift=0, lpt_phase1=0, drain_share=1)
DEFAULTS.update({k: int(v) for k, v in base.items()})
for name, vals in params:
    row = []
    for v in vals:
        sc.set_option(name, v)
        row.append("%d: %.2f" % (v, timed()))
    sc.set_option(name, DEFAULTS[name])
    print("%-13s %s" % (name, "   ".join(row)), flush=True)
