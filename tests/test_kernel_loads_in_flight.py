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


def _isa_of(source, tmp_path_factory):
    hipcc = build._hipcc()
    if not hipcc:
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / (source + ".s"))
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                                          "-I" + CSRC, os.path.join(CSRC, source), "-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return open(out).read().splitlines()


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    return _isa_of("amhip_sort.hip", tmp_path_factory)


@pytest.fixture(scope="module")
def isa_gather(tmp_path_factory):
    return _isa_of("amhip_dsm.hip", tmp_path_factory)


def _body(lines, needle):
    start = [k for k, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(needle), l)]
    assert len(start) == 1, (needle, len(start))
    k = start[0]
    end = next(j for j in range(k, len(lines)) if lines[j].startswith(".Lfunc_end"))  # (not s_endpgm: early exits)
    return lines[k:end]


def _longest_burst(body, what=r"\bglobal_load_"):
    """most global loads issued back to back without a full wait (s_waitcnt vmcnt(0)) between"""
    best = run = 0
    for l in body:
        if re.search(what, l):
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


# The gather's staging: a thread's records (one 16-byte load each) / points (16 + 8 bytes) of the
# tile's region, all asked for before the first is waited for (kCap / 512 per thread).
GATHER_CASES = [
    ("21k_dsm_gather_f32_wideILi512ELi16ELi7680E", r"global_load_dwordx4", 15),
    ("21k_dsm_gather_f32_wideILi512ELi16ELi4096E", r"global_load_dwordx4", 8),
    ("16k_dsm_gather_f32ILi512ELi16ELi2048ELi0E", r"global_load_dwordx4", 4),
    ("16k_dsm_gather_f32ILi512ELi16ELi1024ELi0E", r"global_load_dwordx4", 2),
    ("18k_dsm_gather_tiledILi512ELi16ELi2048E", r"global_load_", 8),   # FP64 mode: 4 points x 2 loads
    ("18k_dsm_gather_tiledILi512ELi16ELi1024E", r"global_load_", 4),
]


@pytest.mark.parametrize("needle,what,want", GATHER_CASES)
def test_the_gathers_staging_asks_for_all_its_records_at_once(isa_gather, needle, what, want):
    got = _longest_burst(_body(isa_gather, needle), what)
    assert got >= want, "%s: %d loads in flight at most, %d expected" % (needle, got, want)


@pytest.fixture(scope="module")
def isa_ortho(tmp_path_factory):
    return _isa_of("amhip_ortho.hip", tmp_path_factory)


def test_the_mosaics_pixel_reads_of_a_lane_leave_together(isa_ortho):
    # two cells per lane in the default kernel: both winners' gray pixels (one byte each) asked for
    # before the first is waited for (the colour / gray branch used to sit between them)
    body = _body(isa_ortho, "22k_ortho_backward_fast4E")
    assert _longest_burst(body, r"global_load_ubyte") >= 2
