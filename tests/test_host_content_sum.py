"""The session's host-side content sums (amhip_content_sum.h, amhip_hostsum.cc): the AVX-512 loop
the library picks on CPUs with avx512dq gives the sums of the plain per-cell loop -- the same
arithmetic the device kernel k_layer_hash runs -- for every length, alignment and position offset
(a mismatch would make the session upload or download matrices it already holds, or worse, skip
one it does not)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, env=None):
    out = str(tmp / "libhostsum_host.so")
    csrc = os.path.join(ROOT, "aerial_mapper_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + csrc,
                           os.path.join(ROOT, "tests", "cpp", "hostsum_host.cc"),
                           os.path.join(csrc, "amhip_hostsum.cc"), os.path.join(csrc, "amhip_tuning.cc"), "-o", out])
    h = C.CDLL(out)
    for f in (h.amt_sum_dispatch, h.amt_sum_plain):
        f.argtypes = [C.c_void_p, C.c_long, C.c_ulonglong, C.c_void_p]
    return h


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("hostsum"))


def _both(lib, cells, g0):
    a = np.zeros(2, np.uint64)
    b = np.zeros(2, np.uint64)
    lib.amt_sum_dispatch(cells.ctypes.data, cells.size, g0, a.ctypes.data)
    lib.amt_sum_plain(cells.ctypes.data, cells.size, g0, b.ctypes.data)
    return a, b


def test_dispatching_sum_equals_the_plain_loop(lib):
    rng = np.random.default_rng(5)
    buf = rng.integers(0, 2 ** 32, 70000, dtype=np.uint64).astype(np.uint32)
    buf[::7] = np.float32(np.nan).view(np.uint32)          # the layers' usual content
    buf[3::11] = 0
    for n in list(range(0, 40)) + [63, 64, 65, 1000, 4097, 65536, 69999]:
        for off in (0, 1, 3, 5):                            # (unaligned starts: columns of a window)
            cells = buf[off:off + n]
            for g0 in (0, 1, 12345678901, 2 ** 40 + 77, 2 ** 64 - 5):   # (positions wrap modulo 2^64)
                a, b = _both(lib, cells, g0)
                assert (a == b).all(), (n, off, g0)


def test_sums_accumulate_into_their_outputs(lib):
    cells = np.arange(1000, dtype=np.uint32)
    a = np.array([5, 7], np.uint64)
    b = np.array([5, 7], np.uint64)
    lib.amt_sum_dispatch(cells.ctypes.data, cells.size, 9, a.ctypes.data)
    lib.amt_sum_plain(cells.ctypes.data, cells.size, 9, b.ctypes.data)
    assert (a == b).all() and a[0] != 5


def test_which_loop_runs_here(lib):
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    has = " avx512dq" in flags and " avx512f" in flags
    assert bool(lib.amt_sum_vectorized()) == has
