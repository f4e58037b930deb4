"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Loader for oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_backend = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load(pkg):
    """`pkg` is the loaded rtiow_rust_amd package (its capi.Backend drives both libraries)."""
    global _backend
    if _backend is None:
        if not os.path.exists(LIB_PATH):
            build()
        _backend = pkg.capi.Backend(LIB_PATH, "rto_")
    return _backend
