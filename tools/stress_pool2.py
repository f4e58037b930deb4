#!/usr/bin/env python3
"""Random worlds of the second program's shape (tests/test_pool2.py random_p2_world) on BOTH full-feature pool kernels against the oracle:
frames and counters.  usage (GPU box): stress_pool2.py [first_seed n_seeds [nx ny ns]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
from test_pool2 import KEYS, random_p2_world  # noqa: E402

pkg = g.load_package()
gpu, ora = pkg.load(), g.load_oracle()
a = [int(v) for v in sys.argv[1:]]
first, n = (a + [100000, 100])[:2]
nx, ny, ns = (a[2:5] + [64, 40, 6])[:3] if len(a) >= 5 else (64, 40, 6)
bad = 0
for seed in range(first, first + n):
    bo, bg = ora.builder(), gpu.builder()
    wo, co, _ = random_p2_world(pkg, bo, seed, nx, ny)
    wg, cg, _ = random_p2_world(pkg, bg, seed, nx, ny)
    assert len(bg.flatten_pool2(wg)[0]) != 0, seed
    io, so_ = bo.scene(wo).par_cast(co, nx, ny, ns, stats=True)
    sg = bg.scene(wg)
    sg.set_option("sync", 0)
    for pool2 in (2, 0):
        sg.set_option("pool2", pool2)
        ig, st = sg.par_cast(cg, nx, ny, ns, stats=True)
        ok = np.array_equal(ig.view(np.uint32), io.view(np.uint32)) and all(st[k] == so_[k] for k in KEYS)
        ok = ok and np.array_equal(sg.par_cast(cg, nx, ny, ns).view(np.uint32), io.view(np.uint32))
        if not ok:
            bad += 1
            print("seed %d pool2=%d DIFFERS: %d floats; counters %s" % (seed, pool2, int((ig.view(np.uint32) != io.view(np.uint32)).sum()),
                                                                        {k: (st[k], so_[k]) for k in KEYS if st[k] != so_[k]}), flush=True)
print("%d worlds x 2 kernels x 2 variants at %dx%dx%d: %d differ" % (n, nx, ny, ns, bad))
