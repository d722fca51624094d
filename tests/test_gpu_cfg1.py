"""BASELINE.json configs[0] exactly: 1 M points from std::mt19937_64(42) ->
1000 x 1000 cells @ 1.0 m, interpolation_radius 1 -- ~3.1 neighbours per cell, ~4 % of the cells
on the expanding-radius ladder (dsm.cc:133-144).  The whole map against the reference's own
compiled dsm.cc where it was built (oracle/_ref), else the restated oracle; both gather modes."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg1():
    pts = synth.make_points_cfg1()
    which = "loops" if O.have_loops() else "port"
    g = O.make_grid(1000.0, 1000.0, 1.0, which="port")
    rc, want, _ = O.dsm_process(pts, g, 1, which=which)
    assert rc == O.OK
    return pts, g, want


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("device_cloud", [False, True])
def test_cfg1_whole_map(cfg1, mode, device_cloud):
    import torch
    import aerial_mapper_amd as A
    pts, g, want = cfg1
    st = A.GridMapSettings(0.0, 0.0, 1000.0, 1000.0, 1.0)
    with A.AerialGridMap(st) as m:
        assert (m.rows, m.cols) == (1000, 1000)
        m.set_dsm_precision(mode == "exact")
        cloud = torch.from_numpy(pts).to("cuda:0") if device_cloud else pts
        A.Dsm(A.DsmSettings(interpolation_radius=1), m).process(cloud, m)
        got = m.get("elevation")
        binned = m.dsm_stats()["points_binned"]
    assert binned == 1_000_000
    assert not np.isnan(want).any()
    frac = S.assert_dsm_close(got, want, tol=1e-6 if mode == "exact" else 1e-4)
    assert frac > (0.99999 if mode == "exact" else 0.99)


def test_cfg1_bench_line(tmp_path):
    """`python bench.py --workload cfg1` prints the contract's line with whole-map parity"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg1", "--steps", "5",
                        "--warmup", "2", "--no-host-path"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["workload"].startswith("cfg1")
    ps = line["parity_sample"]
    assert ps["cells"] == 1_000_000 and ps["dsm_nan_pattern_equal"]
    assert ps["dsm_max_abs_err_m"] <= 1e-6 and ps["dsm_bit_identical_frac"] > 0.99999
    assert line["cpu_baseline"]["kind"] in ("reference", "port")
