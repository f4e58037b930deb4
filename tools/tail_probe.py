#!/usr/bin/env python3
"""Fixed cost per launch of the book-1 kernel (GPU box): kernel ms over spp for several bounce caps, with the
least-squares intercept (start-up + end-of-frame tail) and slope.  usage: tail_probe.py [case] [nx ny]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); capi = pkg.capi
case = sys.argv[1] if len(sys.argv) > 1 else "book1"
nx, ny = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1200, 800)
sc, cam, _, _, _ = build_case(pkg, gpu, case, nx, ny)
out = np.zeros((ny, nx, 3), dtype=np.float32)


def kernel_ms(ns, mb, reps=4):
    best = 1e9
    for _ in range(reps):
        p = capi.make_params(nx, ny, ns, max_bounces=mb)
        st = capi.Stats(); st.struct_size = C.sizeof(capi.Stats)
        gpu.check(gpu._par_cast(sc.h, C.byref(cam), C.byref(p), out.ctypes.data_as(capi.c_f32p), C.byref(st)))
        best = min(best, st.kernel_ms)
    return best


kernel_ms(10, 50)
spps = [10, 25, 50, 100, 200]
for mb in (50, 12, 4):
    ms = [kernel_ms(ns, mb) for ns in spps]
    slope, icpt = np.polyfit(spps, ms, 1)
    print("%s max_bounces %2d: " % (case, mb) + "  ".join("%d spp %.2f ms" % (a, b) for a, b in zip(spps, ms)) +
          "  | fit: %.3f ms/spp + %.2f ms fixed" % (slope, icpt))
