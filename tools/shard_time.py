"""Kernel time of one rank of an N-rank frame (pixel tiles interleaved, spp = 50 N: the weak-scaling shard bench.py --gpus N
renders per GPU), measured on ONE GPU.  usage (GPU box): python tools/shard_time.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); capi = pkg.capi
nx, ny = 1200, 800
sc, cam, _, _, _ = build_case(pkg, gpu, "book1", nx, ny)
out = np.zeros((ny, nx, 3), dtype=np.float32)
def ms(ns, rank, nranks):
    best = 1e9
    for _ in range(4):
        p = capi.make_params(nx, ny, ns, rank=rank, nranks=nranks)
        st = capi.Stats(); st.struct_size = C.sizeof(capi.Stats)
        gpu.check(gpu._par_cast(sc.h, C.byref(cam), C.byref(p), out.ctypes.data_as(capi.c_f32p), C.byref(st)))
        best = min(best, st.kernel_ms)
    return best
print("N=1 50spp: %.2f ms" % ms(50, 0, 1))
for n in (2, 4, 8):
    print("N=%d rank0 %dspp: %.2f ms   rank%d: %.2f ms" % (n, 50 * n, ms(50 * n, 0, n), n - 1, ms(50 * n, n - 1, n)))
