import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive scans, excluded from the default CPU run")


@pytest.fixture(scope="session")
def pkg():
    p = graft.load_package()
    if not os.path.exists(p.LIB_PATH):   # fresh checkout: compile the HIP library (hipcc cross-compiles anywhere)
        graft.build()
    return p


@pytest.fixture(scope="session")
def oracle():
    """TEST INFRASTRUCTURE: the CPU restatement (oracle/liboracle.so)."""
    return graft.load_oracle()


@pytest.fixture(scope="session")
def gpu(pkg):
    """The HIP product through its C ABI.  Fails loudly if the library is missing or no GPU."""
    be = pkg.load()
    n = be.device_count()
    assert n >= 1, "no HIP device visible"
    return be


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    """Bit-exact float32 equality; NaNs compare equal to NaNs (x86 and gfx950 differ in NaN sign/payload)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)
        first = tuple(idx[0])
        raise AssertionError("%s: %d of %d floats differ; first at %s: %r vs %r" % (
            what, idx.shape[0], a.size, first, a[first], b[first]))


def free_port():
    """A rendezvous port nobody holds right now (bind port 0, as bench.py self_launch does): two suites on one host
    must not meet on a fixed number."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def retry_on_busy_port(run, attempts=3):
    """free_port() closes its socket before the rendezvous binds the number: another process may take it in between.
    `run(port)` starts the ranks and returns (returncode, stderr text); a failed start whose stderr names the address is
    repeated on a fresh port (anything else is returned as it is)."""
    last = None
    for _ in range(attempts):
        last = run(free_port())
        rc, err = last
        if rc == 0 or not any(k in (err or "") for k in ("Address already in use", "EADDRINUSE", "address already in use", "errno: 98")):
            return last
    return last
