"""amhip_atan_cr.h (the device's atan for aslam's equidistant distortion) compiled on the host:
correctly rounded on every sampled input (mpmath, 200 bits), edge cases included; and how often
the host's own libm agrees with the correctly rounded value (what the GPU-vs-reference parity of
equidistant cameras rests on: tests/test_gpu_reference_loops.py)."""
import ctypes as C
import math
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("atan") / "libatan_cr_host.so")
    src = os.path.join(ROOT, "tests", "cpp", "atan_cr_host.cc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "aerial_mapper_amd", "csrc"), src, "-o", out])
    h = C.CDLL(out)
    h.amt_atan_cr.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    h.amt_atan_cr_fast.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    h.amt_atan_device.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    return h


def _run(lib, x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib.amt_atan_cr(x.ctypes.data, x.size, out.ctypes.data)
    return out


def _cr(x):
    import mpmath
    mpmath.mp.prec = 200
    return np.array([float(mpmath.atan(mpmath.mpf(float(v)))) for v in x])


def test_atan_cr_is_correctly_rounded_on_random_inputs(lib):
    pytest.importorskip("mpmath")
    rng = random.Random(7)
    xs = [rng.uniform(0.0, 2.5) for _ in range(20000)]              # the distortion's usual radii
    xs += [math.exp(rng.uniform(-20.0, 45.0)) for _ in range(20000)]  # every magnitude
    xs += [j / 32.0 + d for j in range(33) for d in (0.0, 1e-17, -1e-17, 1e-9, -1e-9, 1 / 64.0 - 1e-12)
           if j / 32.0 + d >= 0.0]                                   # the reduction's seams
    xs = np.array(xs)
    got, want = _run(lib, xs), _cr(xs)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(float(xs[k]), float(got[k]), float(want[k])) for k in bad[:5]]


def test_atan_cr_edge_cases(lib):
    x = np.array([0.0, 5e-324, 1e-300, 2.0 ** -27, 2.0 ** -27 * (1 - 2 ** -53), 1.0, 1.0 + 2 ** -52, 2.0 ** 60,
                  2.0 ** 61, 1e300, np.inf, np.nan, -0.5])
    got = _run(lib, x)
    assert got[0] == 0.0 and got[1] == 5e-324 and got[2] == 1e-300
    assert got[5] == math.pi / 4 and got[10] == math.pi / 2 and got[9] == math.pi / 2
    assert math.isnan(got[11]) and got[12] == -_run(lib, np.array([0.5]))[0]
    pytest.importorskip("mpmath")
    fin = np.isfinite(x) & (x >= 0)
    assert np.array_equal(got[fin], _cr(x[fin]))


def test_the_hosts_libm_is_correctly_rounded_almost_always(lib):
    """Not a property of this repository: it documents why equidistant cameras are held to the
    reference bit for bit in the GPU tests although the two sides call different atan routines."""
    rng = random.Random(11)
    xs = np.array([rng.uniform(0.0, 2.5) for _ in range(100000)])
    mine = _run(lib, xs)
    host = np.array([math.atan(v) for v in xs])
    differ = int((mine != host).sum())
    assert differ <= 0.005 * xs.size, differ          # glibc 2.35: ~0.1 %
    assert np.abs(mine - host).max() <= 2.3e-16      # and then by one ulp


def test_fast_path_is_right_whenever_it_says_so_and_says_so_almost_always(lib):
    """atan_cr_fast (plain doubles + carried errors + a rounding test) against the double-double
    routine on 2 M inputs, against mpmath on a sample; its refusals are ~1 in 65 000."""
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(2.0 ** -27, 2.5, 1_000_000),
                         np.exp(rng.uniform(np.log(2.0 ** -27), np.log(2.0 ** 60), 1_000_000)),
                         np.array([j / 32.0 + d for j in range(33) for d in (0.0, 1e-17, 1e-9, 1 / 64.0 - 1e-12)
                                   if j / 32.0 + d >= 2.0 ** -27])])
    out = np.empty_like(xs)
    ok = np.empty(xs.size, np.uint8)
    lib.amt_atan_cr_fast(xs.ctypes.data, xs.size, out.ctypes.data, ok.ctypes.data)
    full = _run(lib, xs)
    good = ok.astype(bool)
    assert np.array_equal(out[good], full[good])
    assert (~good).sum() <= 2e-4 * xs.size, int((~good).sum())
    pytest.importorskip("mpmath")
    pick = rng.choice(xs.size, 20000, replace=False)
    pick = pick[good[pick]]
    assert np.array_equal(out[pick], _cr(xs[pick]))


def test_the_device_routine_is_total_and_equals_the_rigorous_one_on_millions_of_inputs(lib):
    """atan_device = the fast path's hi + lo rounded once, no second step (76 good bits: a misrounding
    needs the exact value within 2^-76 of a boundary -- none in 2 M samples), plus the shortcuts."""
    rng = np.random.default_rng(6)
    xs = np.concatenate([rng.uniform(0.0, 2.5, 1_000_000),
                         np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 1_000_000)),
                         np.array([0.0, 5e-324, 2.0 ** -27, 2.0 ** 60, 2.0 ** 61, np.inf])])
    out = np.empty_like(xs)
    lib.amt_atan_device(xs.ctypes.data, xs.size, out.ctypes.data)
    assert np.array_equal(out, _run(lib, xs))
    nan = np.array([np.nan])
    lib.amt_atan_device(nan.ctypes.data, 1, out.ctypes.data)
    assert np.isnan(out[0])
