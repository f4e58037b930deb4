import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
from test_pool2 import PARTS
pkg = g.load_package(); gpu = pkg.load(); ora = g.load_oracle()
fn, nx, ny, ns = PARTS["book2"]
bo = ora.builder(); wo, co, _ = fn(pkg, bo, nx, ny); so = bo.scene(wo)
xs, ys = np.meshgrid(np.arange(0, nx, 1, dtype=np.uint32), np.arange(0, ny, 1, dtype=np.uint32)); xs, ys = xs.ravel(), ys.ravel(); ss = (xs + ys) % ns
rgb_o, info_o = so.debug_samples(co, nx, ny, ns, xs, ys, ss)
OLD = "r5final" in os.environ.get("RTIOW_GPU_LIB", "")
CASES = (("first kernel alone", {}, False), ("first kernel, hoist=0", {"hoist": 0}, False)) if OLD else (("pool2=0 alone", {}, False), ("pool2=0, hoist=0", {"hoist": 0}, False), ("pool2=2 alone", {"pool2": 2}, False))
for label, opts, pre in CASES:
    bad = 0
    for it in range(25):
        bg = gpu.builder(); wg, cg, _ = fn(pkg, bg, nx, ny); sg = bg.scene(wg); sg.set_option("sync", 0)
        if pre:
            sg.set_option("pool2", 2); sg.debug_samples(cg, nx, ny, ns, xs, ys, ss, trace_kernel=True)
        if not OLD: sg.set_option("pool2", 0)
        for k, v in opts.items(): sg.set_option(k, v)
        rgb, info = sg.debug_samples(cg, nx, ny, ns, xs, ys, ss, trace_kernel=True)
        if not np.array_equal(info, info_o):
            bad += 1
            if bad == 1:
                d = np.argwhere(info != info_o)
                print("   first diff:", d[:4].tolist(), info[d[0][0]].tolist(), info_o[d[0][0]].tolist(), "n rows differing", len(set(d[:,0].tolist())))
    print(label, "mismatches", bad, "of 25", flush=True)
