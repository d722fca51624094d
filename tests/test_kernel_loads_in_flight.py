"""Loads in flight: the unrolled per-thread loops of the sort's
scatter / placement passes and of the count pass must ISSUE all of a thread's loads before they
wait for any.  The compiler once put every load behind the previous one's `s_waitcnt vmcnt(0)`
(key arithmetic with branches / LDS atomics sat between them): 6 - 8 dependent memory round
trips per workgroup, invisible in the source and in the register remarks.  This test reads the
ISA of amhip_sort.hip (hipcc -S, cross-compiled: no GPU) and counts, per kernel, the global loads
issued before the first wait that needs one of them."""
import os
import re
import subprocess

import pytest

from aerial_mapper_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aerial_mapper_amd", "csrc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = build._hipcc()
    if not hipcc:
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "sort.s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                                          "-I" + CSRC, os.path.join(CSRC, "amhip_sort.hip"), "-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return open(out).read().splitlines()


def _body(lines, needle):
    start = [k for k, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(needle), l)]
    assert len(start) == 1, (needle, len(start))
    k = start[0]
    end = next(j for j in range(k, len(lines)) if "s_endpgm" in lines[j])
    return lines[k:end]


def _longest_burst(body):
    """most global loads issued back to back without a full wait (s_waitcnt vmcnt(0)) between"""
    best = run = 0
    for l in body:
        if re.search(r"\bglobal_load_", l):
            run += 1
            best = max(best, run)
        elif re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", l):
            run = 0
    return best


# (kernel, loads a thread must have in flight at once: records are 16 + 4 bytes = 2 loads,
# points 16 + 8 bytes = 2 loads)
CASES = [
    ("k_dsm_p3_scatter_recILb0E", 2 * 6),   # six 20-byte records per thread
    ("k_dsm_p3_scatter_recILb1E", 2 * 6),   # six points of the cloud
    ("16k_dsm_p3_scatterILb0E", 2 * 5),     # five 24-byte points
    ("16k_dsm_p3_scatterILb1E", 2 * 5),
    ("18k_dsm_p3_place_recE", 2 * 8),       # eight records of the sub-partition
    ("14k_dsm_p3_placeE", 2 * 8),
    ("k_dsm_p3_countILb0E", 2 * 4),         # four points (x, y | z)
]


@pytest.mark.parametrize("needle,want", CASES)
def test_all_loads_of_a_thread_leave_before_the_first_wait(isa, needle, want):
    got = _longest_burst(_body(isa, needle))
    assert got >= want, "%s: %d loads in flight at most, %d expected" % (needle, got, want)
