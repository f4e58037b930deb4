#!/usr/bin/env python3
"""Drain timeline of a pool kernel (GPU box, a -DRT_TIMELINE build of the library: tools/build_variant.sh tl -DRT_TIMELINE, then
RTIOW_GPU_LIB=.../variants/tl.so): per-wave records of rt_pool.h RT_TL_* -- when each wave sees the work queue empty, when its
pool thins out, when it is done, what it did in between.  usage: timeline.py [case nx ny ns [max_bounces [rank nranks]]]"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from scene_cases import build_case
pkg = g.load_package(); gpu = pkg.load(); capi = pkg.capi
a = sys.argv[1:]
case, nx, ny, ns = (a[0], int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else ("book1", 1200, 800, 50)
mb = int(a[4]) if len(a) > 4 else 50
rank, nranks = (int(a[5]), int(a[6])) if len(a) > 6 else (0, 1)
path = os.path.join(tempfile.gettempdir(), "rtg_timeline.bin")
os.environ["RTG_TIMELINE_OUT"] = path
sc, cam, _, _, _ = build_case(pkg, gpu, case, nx, ny)
out = np.zeros((ny, nx, 3), dtype=np.float32)


def run():
    p = capi.make_params(nx, ny, ns, max_bounces=mb, rank=rank, nranks=nranks)
    st = capi.Stats(); st.struct_size = C.sizeof(capi.Stats)
    gpu.check(gpu._par_cast(sc.h, C.byref(cam), C.byref(p), out.ctypes.data_as(capi.c_f32p), C.byref(st)))
    return st.kernel_ms


for _ in range(3):
    ms = run()
raw = open(path, "rb").read()
rec = np.frombuffer(raw[4:], dtype=np.uint32).reshape(-1, 16)
rec = rec[rec[:, 4] != 0]
if not len(rec):
    raise SystemExit("no timeline records: is RTIOW_GPU_LIB a -DRT_TIMELINE build?")
t0 = rec[:, 0].min()
us = lambda c: (rec[:, c].astype(np.int64) - int(t0)) / 100.0   # 100 MHz
start, exh, done = us(0), us(1), us(4)
exh = np.where(rec[:, 1] == 0, done, exh)
le64 = np.where(rec[:, 3] == 0, done, us(3)); le8 = np.where(rec[:, 10] == 0, done, us(10)); le1 = np.where(rec[:, 11] == 0, done, us(11))
q = lambda v: "min %.0f  p10 %.0f  median %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
    v.min(), np.percentile(v, 10), np.median(v), v.mean(), np.percentile(v, 90), np.percentile(v, 99), v.max())
print("%s %dx%dx%d cap %d: kernel %.2f ms (events), %d waves; microseconds from the first wave's start" % (case, nx, ny, ns, mb, ms, len(rec)))
print("  wave starts           : " + q(start))
print("  sees queue empty      : " + q(exh))
print("  live paths <= 64      : " + q(le64))
print("  live paths <= 8       : " + q(le8))
print("  live paths <= 1       : " + q(le1))
print("  done                  : " + q(done))
drain = done - exh
print("  drain (done - empty)  : " + q(drain))
print("  live paths at empty   : " + q(rec[:, 2].astype(np.float64)) + "   (T %.0f  S %.0f  E|X %.0f on the lists)" % (rec[:, 12].mean(), rec[:, 13].mean(), rec[:, 14].mean()))
print("  idle wave-time after 'done' until the last wave: %.1f %% of waves x kernel;  between 'empty' and 'done': %.1f %%" % (
    100 * (done.max() - done).mean() / done.max(), 100 * drain.mean() / done.max()))
order = np.argsort(-done)
print("  the 12 last waves (wave: empty -> <=64 -> <=8 -> <=1 -> done | live at empty, shade passes / their lanes / with a >=16-bounce ray, services, rays after empty):")
for k in order[:12]:
    print("   %5d: %7.0f %7.0f %7.0f %7.0f %7.0f | %3d live, %4d passes / %5d lanes / %4d deep, %5d services, %5d rays" % (
        k, exh[k], le64[k], le8[k], le1[k], done[k], rec[k, 2], rec[k, 5], rec[k, 8], rec[k, 7], rec[k, 6], rec[k, 9]))
# how the chip empties: waves still running over time
for t in np.linspace(np.percentile(exh, 1), done.max(), 12):
    print("  t = %7.0f us: %5d waves before 'empty', %5d draining, %5d done" % (t, (exh > t).sum(), ((exh <= t) & (done > t)).sum(), (done <= t).sum()))
late = drain > np.percentile(drain, 99)
print("  slowest 1 %% of drains: live at empty %.0f, passes %.0f (%.1f lanes each), deep-pass share %.2f, rays %.0f, %.1f us per pass" % (
    rec[late, 2].mean(), rec[late, 5].mean(), rec[late, 8].sum() / max(rec[late, 5].sum(), 1), rec[late, 7].sum() / max(rec[late, 5].sum(), 1),
    rec[late, 9].mean(), drain[late].sum() / max(rec[late, 5].sum(), 1)))
