"""BASELINE.json configs[0] / SURVEY.md 8d cfg1: the input is defined through std::mt19937_64.
The numpy engine of aerial_mapper_amd/synth.py is held to libstdc++'s own (a small program compiled
here with g++) draw by draw, and the oracle's DSM of the exact configuration to the figures the
survey quotes for it (~3.1 neighbours per cell, ~4 % of the cells on the fallback ladder,
dsm.cc:133-144)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
from aerial_mapper_amd import synth

SRC = r"""
#include <cstdio>
#include <random>
int main() {
  std::mt19937_64 e(42);
  for (int i = 0; i < 5; ++i) std::printf("%llu\n", (unsigned long long)e());
  std::mt19937_64 f(42);
  std::uniform_real_distribution<double> d(-500.0, 500.0), n(-0.05, 0.05);
  for (int i = 0; i < 3; ++i) { double x = d(f), y = d(f), z = n(f); std::printf("%.17g %.17g %.17g\n", x, y, z); }
  std::mt19937_64 g(42);
  unsigned long long x = 0;
  for (int i = 0; i < 3000000; ++i) x ^= g() * (2ull * i + 1ull);
  std::printf("%llu\n", x);
}
"""


def test_numpy_engine_is_libstdcxx_mt19937_64(tmp_path):
    c = tmp_path / "mt.cc"
    c.write_text(SRC)
    exe = str(tmp_path / "mt")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-o", exe, str(c)])
    lines = subprocess.run([exe], stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout.split("\n")
    e = synth.Mt19937_64(42)
    assert [int(v) for v in e.draws(5)] == [int(v) for v in lines[:5]]
    f = synth.Mt19937_64(42)
    for k in range(3):
        want = [float(v) for v in lines[5 + k].split()]
        got = [f.uniform(-500.0, 500.0, 1)[0], f.uniform(-500.0, 500.0, 1)[0], f.uniform(-0.05, 0.05, 1)[0]]
        assert got == want          # bit for bit: std::uniform_real_distribution<double>
    g = synth.Mt19937_64(42)
    w = g.draws(3_000_000)
    with np.errstate(over="ignore"):
        mix = np.bitwise_xor.reduce(w * (np.arange(3_000_000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)))
    assert int(mix) == int(lines[8])
    # chunked draws continue the same stream
    h = synth.Mt19937_64(42)
    parts = np.concatenate([h.draws(k) for k in (1, 311, 312, 313, 1000)])
    assert np.array_equal(parts, w[:parts.size])


def test_cfg1_points_are_the_survey_s_configuration():
    pts = synth.make_points_cfg1()
    assert pts.shape == (1_000_000, 3)
    assert pts[:, :2].min() >= -500.0 and pts[:, :2].max() < 500.0
    f = synth.Mt19937_64(42)
    u = f.canonical(6)
    assert pts[0, 0] == u[0] * 1000.0 - 500.0 and pts[0, 1] == u[1] * 1000.0 - 500.0
    assert pts[1, 0] == u[3] * 1000.0 - 500.0
    z = 400.0 + 10.0 * np.sin(0.01 * pts[:, 0]) * np.cos(0.01 * pts[:, 1])
    assert np.abs(pts[:, 2] - z).max() <= 0.05


@pytest.mark.timeout(600)
def test_cfg1_oracle_ladder_share():
    """the exact configuration through the oracle: every cell filled, ~4 % of them by the ladder"""
    pts = synth.make_points_cfg1()
    g = O.make_grid(1000.0, 1000.0, 1.0)
    assert (g.rows, g.cols) == (1000, 1000)
    rc, elev, _ = O.dsm_process(pts, g, 1)
    assert rc == O.OK
    # cells whose first search (d2 < 1) is empty: count them independently with a 2-D histogram
    # bound -- a cell with no point within the 3 x 3 cells around it certainly takes the ladder,
    # a cell with a point in its own cell's inscribed disc certainly does not
    assert not np.isnan(elev).any()           # the last ladder radius (2.59 m) reaches a point everywhere
    # exact share of ladder cells by brute force on a 200 x 200 corner
    s = 200
    cx = (g.pos_x + g.length_x / 2 - g.resolution / 2) - g.resolution * np.arange(s)
    cy = (g.pos_y + g.length_y / 2 - g.resolution / 2) - g.resolution * np.arange(s)
    keep = (pts[:, 0] > cx.min() - 1.5) & (pts[:, 1] > cy.min() - 1.5)
    p = pts[keep]
    empty = 0
    for i in range(s):
        near = p[np.abs(p[:, 0] - cx[i]) < 1.0]
        d2 = (near[:, 0][None, :] - cx[i]) ** 2 + (near[:, 1][None, :] - cy[:, None]) ** 2
        empty += int((~(d2 < 1.0).any(axis=1)).sum())
    share = empty / float(s * s)
    assert 0.03 < share < 0.055, share        # e^-pi = 0.0432
