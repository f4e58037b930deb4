"""Kernel time of the shards of an N-rank frame (pixel tiles interleaved by rank, 16x16 up to 4 ranks and 8x8 from 8 on: what `bench.py --gpus N` and
rtg_par_cast_multi give each GPU), each measured ALONE on ONE GPU -- a projection of the per-GPU time of a multi-GPU run
(the slowest shard + the framebuffer reduce), not a multi-GPU measurement.  usage (GPU box): python tools/shard_time.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); capi = pkg.capi
from rtiow_rust_amd import parallel


def shard_ms(sc, cam, nx, ny, ns, rank, nranks, out):
    best = 1e9
    for _ in range(3):
        tw, th = parallel.shard_tile(nranks)   # the interleave bench.py --gpus N uses
        p = capi.make_params(nx, ny, ns, rank=rank, nranks=nranks, tile_w=tw, tile_h=th)
        st = capi.Stats(); st.struct_size = C.sizeof(capi.Stats)
        gpu.check(gpu._par_cast(sc.h, C.byref(cam), C.byref(p), out.ctypes.data_as(capi.c_f32p), C.byref(st)))
        best = min(best, st.kernel_ms)
    return best


for name, nx, ny, ns in (("book1", 1200, 800, 500), ("book2", 800, 800, 1000)):
    sc, cam, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    out = np.zeros((ny, nx, 3), dtype=np.float32)
    t1 = shard_ms(sc, cam, nx, ny, ns, 0, 1, out)
    print("%s %dx%dx%d (strong scaling: the frame is fixed)   1 GPU: %.2f ms" % (name, nx, ny, ns, t1))
    for n in (2, 4, 8):
        ts = [shard_ms(sc, cam, nx, ny, ns, r, n, out) for r in range(n)]
        print("   %d shards: slowest %.2f ms, fastest %.2f ms  ->  %.2fx of %dx if the reduce were free (%.0f %%)" % (
            n, max(ts), min(ts), t1 / max(ts), n, 100.0 * t1 / max(ts) / n))
