"""rtiow-rust_amd: MI355X-native path-tracing hot path behind rtiow-rust's `par_cast` seam.

Layout:
  csrc/        HIP kernels (gfx950) + host flattener + the C ABI of include/rtiow_gpu.h
               -> csrc/librtiow_gpu.so (built in-tree by __graft_entry__.build())
  capi.py      ctypes binding of the C ABI (method names mirror the crate's constructors)
  scenes.py    the reference's scene builders (src/lib.rs, src/main.rs, benches/scene.rs)
  small_rng.py SmallRng (Pcg64Mcg) emulation used ONLY for host-side scene construction
  host/        C++ mirror of the crate surface over the C ABI (rtiow.hpp)

The directory name contains a '-', so import it through `__graft_entry__.load_package()` (which
registers it as module `rtiow_rust_amd`).

There is NO CPU fallback: load() raises if the HIP library is missing.
"""
import os

from . import capi, ppm, scenes, small_rng  # noqa: F401  (parallel imports torch: import it explicitly)
from .capi import Backend, Camera, Params, Stats, RtError, make_params  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librtiow_gpu.so")

_backend = None


def load():
    """Load librtiow_gpu.so (the HIP product).  Fails loudly when it has not been built."""
    global _backend
    if _backend is None:
        # RTIOW_GPU_LIB: measurement hook (tools/sweep*.sh) -- another BUILD of the same HIP library, e.g. one compiled
        # with different -DRT_* experiment defines; never anything but librtiow_gpu
        _backend = Backend(os.environ.get("RTIOW_GPU_LIB", LIB_PATH), "rtg_")
    return _backend
