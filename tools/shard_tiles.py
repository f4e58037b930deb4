"""Kernel time of every shard of an N-rank frame, rendered alone on ONE GPU, for several tile shapes of the interleave
(rtg_params.tile_w / tile_h): which shape balances 2 / 4 / 8 ranks best.  usage (GPU box): shard_tiles.py [case nx ny ns]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); capi = pkg.capi
CASE, NX, NY, NS = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("book1", 1200, 800, 500)
sc, cam, _, _, _ = build_case(pkg, gpu, CASE, NX, NY)
out = np.zeros((NY, NX, 3), dtype=np.float32)
def ms(rank, n, ns=None, tw=16, th=None):
    ns = ns or NS; th = th or tw
    best = 1e9
    for _ in range(4):
        p = capi.make_params(NX, NY, ns, rank=rank, nranks=n, tile_w=tw, tile_h=th)
        st = capi.Stats(); st.struct_size = C.sizeof(capi.Stats)
        gpu.check(gpu._par_cast(sc.h, C.byref(cam), C.byref(p), out.ctypes.data_as(capi.c_f32p), C.byref(st)))
        best = min(best, st.kernel_ms)
    return best
for tw, th in ((16, 16), (8, 8), (16, 8), (8, 16)):
    t8 = [ms(r, 8, tw=tw, th=th) for r in range(8)]; t4 = [ms(r, 4, tw=tw, th=th) for r in range(4)]; t2 = [ms(r, 2, tw=tw, th=th) for r in range(2)]
    print("%s tile %2dx%-2d: 8 shards max %.2f mean %.2f | 4 shards max %.2f mean %.2f | 2 shards max %.2f mean %.2f" % (CASE, tw, th, max(t8), sum(t8) / 8, max(t4), sum(t4) / 4, max(t2), sum(t2) / 2))
