#!/usr/bin/env python3
"""Run one scene case GPU-vs-oracle in THIS process (wrap in `timeout` on the GPU box: a hung kernel must not eat the budget).
usage: try_case.py <case|fuzzb:SEED> [nx ny ns] [stats]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from scene_cases import build_case, CASES
pkg = g.load_package(); gpu = pkg.load(); ora = g.load_oracle()
name = sys.argv[1]
stats = "stats" in sys.argv
if name.startswith("fuzzb:"):
    import test_fuzz
    seed = int(name.split(":")[1])
    nx, ny, ns = 40, 24, 5
    bg, wg, cam_g = test_fuzz._build(pkg, gpu, 5000 + seed, nx, ny, True)
    bo, wo, cam_o = test_fuzz._build(pkg, ora, 5000 + seed, nx, ny, True)
    sg, so = bg.scene(wg), bo.scene(wo)
else:
    a = [int(x) for x in sys.argv[2:5] if x.isdigit()]
    sg, cam_g, nx, ny, ns = build_case(pkg, gpu, name, *(a[:2] if len(a) >= 2 else []))
    so, cam_o, _, _, _ = build_case(pkg, ora, name, nx, ny)
    if len(a) >= 3: ns = a[2]
print("rendering", name, nx, ny, ns, "stats" if stats else "", flush=True)
img_g = sg.par_cast(cam_g, nx, ny, ns, stats=stats)
img_o = so.par_cast(cam_o, nx, ny, ns, stats=stats)
if stats:
    print(img_g[1], img_o[1]); img_g, img_o = img_g[0], img_o[0]
same = (img_g.view(np.uint32) == img_o.view(np.uint32)) | (np.isnan(img_g) & np.isnan(img_o))
print(name, "bit-equal:", bool(same.all()), "differing:", int((~same).sum()), flush=True)
