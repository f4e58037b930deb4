"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Loader for oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTIOW_ORACLE_LIB: the sanitizer leg (tests/test_oracle_sanitized.py) loads oracle/_san/liboracle_san.so instead
LIB_PATH = os.environ.get("RTIOW_ORACLE_LIB", os.path.join(_HERE, "liboracle.so"))
_backend = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load(pkg):
    """`pkg` is the loaded rtiow_rust_amd package: liboracle.so exports the product's entry points name
    for name (rtg_* -> rto_*), so the product's ctypes binder drives it with the same calls.  The
    oracle-only entry points (thread count for par_cast, sequential cast()) are declared here."""
    global _backend
    if _backend is not None:
        return _backend
    if not os.path.exists(LIB_PATH):
        build()
    import ctypes as C
    import numpy as np
    capi = pkg.capi

    class OracleScene(capi.Scene):
        def _par_cast_args(self, args, threads):
            return args + [threads]

        def cast(self, camera, nx, ny, ns, small_rng_seed=0xDEADBEEF, max_bounces=50):
            """cast, lib.rs:378: sequential, ONE SmallRng stream through the whole image."""
            out = np.zeros((ny, nx, 3), dtype=np.float32)
            self.be.check(self.be._cast(self.h, C.byref(camera), nx, ny, ns, max_bounces, small_rng_seed,
                                        out.ctypes.data_as(capi.c_f32p)))
            return out

    class OracleBackend(capi.Backend):
        scene_class = OracleScene

        def _declare_render(self):
            f = self._fn
            f("par_cast", C.c_int, [C.c_void_p, C.POINTER(capi.Camera), C.POINTER(capi.Params), capi.c_f32p,
                                    C.POINTER(capi.Stats), C.c_int])
            f("par_cast_multi", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(capi.Camera), C.POINTER(capi.Params),
                                          capi.c_f32p, C.POINTER(capi.Stats)])
            f("cast", C.c_int, [C.c_void_p, C.POINTER(capi.Camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_uint64, capi.c_f32p])

    _backend = OracleBackend(LIB_PATH, "rto_")
    return _backend
