#!/usr/bin/env python3
"""GPU box: the device copies of the libm restatements (csrc/rt_libm.h: rt_logf / rt_pow5f / rt_sinf) against THIS host's
platform libm (glibc logf / powf(x, 5) / sinf, called through the oracle's rto_debug_glibc -- test infrastructure) on ALL 2^32
float bit patterns.  Prints one line per function; exit code 1 on any difference (NaN == NaN)."""
import ctypes as C
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package(); gpu = pkg.load(); ora = g.load_oracle()
fp = C.POINTER(C.c_float)
CH = 1 << 26
bad_total = 0
for op, name in ((0, "logf(x)"), (1, "powf(x, 5.0f)"), (2, "sinf(x)")):
    bad = 0
    for c in range((1 << 32) // CH):
        x = (np.arange(CH, dtype=np.uint64) + c * CH).astype(np.uint32).view(np.float32)
        got = gpu.debug_math(op, x)
        want = np.empty_like(x)
        parts = 16
        def work(i):
            sl = slice(i * CH // parts, (i + 1) * CH // parts)
            xs = np.ascontiguousarray(x[sl]); ws = np.empty_like(xs)
            ora.lib.rto_debug_glibc(C.c_int(op), C.c_size_t(xs.size), xs.ctypes.data_as(fp), ws.ctypes.data_as(fp))
            want[sl] = ws
        with ThreadPoolExecutor(parts) as ex:
            list(ex.map(work, range(parts)))
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        bad += int((~same).sum())
    print("GPU rt_%s vs host glibc %s: 4294967296 tested, %d differ" % (("logf", "pow5f", "sinf")[op], name, bad), flush=True)
    bad_total += bad
sys.exit(1 if bad_total else 0)
