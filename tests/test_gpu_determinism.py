"""GPU: the default (FP64) mode stores the same floats in every run, whatever order the
binning's atomics leave the points in (the reference's kd-tree order is
fixed: nanoflann.hpp:929-946).

How (amhip_dsm.hip, round_is_certain / canonical_search): a routine stores (float)h only when
every value within its own error bound of h rounds to that float; the few cells per 1e8 that
sit on a float rounding boundary are redone in double-double sums of the reference's terms,
where the order of the additions cannot reach the 24th bit.  Two consequences are tested:
  * N runs on one cloud -- and a run on a PERMUTED cloud -- give equal bits in all 1e8 cells;
  * forcing EVERY cell through canonical_search (tuning knob dsm_canon_all) gives the same bits as
    the normal run: the error bound really covers the distance between the two arithmetics."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_mode_is_bit_reproducible_at_full_size():
    """BASELINE configs[1]: 50 M points -> 10 000 x 10 000 cells; ten runs and a permuted cloud."""
    import torch
    import aerial_mapper_amd as A
    side, res, npts = 10000, 0.25, 50_000_000
    L = side * res
    dev = torch.device("cuda", 0)
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        m.set_dsm_precision(True)
        pts = synth.make_points_torch(npts, L / 2.0 + 4.0, 243, dev)
        dsm = A.Dsm(A.DsmSettings(), m)
        dsm.process(pts, m)
        first = m.as_torch("elevation").clone()
        assert not torch.isnan(first).any()
        for run in range(9):
            m.reset()
            dsm.process(pts, m)
            assert torch.equal(first.view(torch.int32), m.as_torch("elevation").view(torch.int32)), run
        perm = torch.randperm(npts, device=dev)
        shuffled = pts[perm].contiguous()
        del perm
        m.reset()
        dsm.process(shuffled, m)
        again = m.as_torch("elevation")
        diff = int((first.view(torch.int32) != again.view(torch.int32)).sum().item())
        assert diff == 0, "%d of 1e8 cells depend on the order of the cloud" % diff


_CHILD = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, aerial_mapper_amd as A, scenarios as S
from aerial_mapper_amd import synth
out = {}
def run(name, sc, radius=1):
    g = sc.grid
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)) as m:
        m.set_dsm_precision(True)
        A.Dsm(A.DsmSettings(radius), m).process(sc.points, m)
        out[name] = m.get("elevation")
%(scenes)s
np.savez(sys.argv[1], **out)
"""

_SCENES = r"""
run("sparse_ladder", S.Scene(300.0, 200.0, 1.0, 66000, seed=42))
run("dense_quarter", S.Scene(60.0, 45.0, 0.25, int(8 * 68 * 68), seed=43, point_extent=34.0))
run("very_sparse", S.Scene(200.0, 160.0, 1.0, 1800, seed=45))
rng = np.random.default_rng(9)
sc = S.Scene(72.0, 44.0, 0.25, 10, seed=5)
n = int(9.0 * sc.grid.rows * sc.grid.cols)
pts = np.empty((n, 3))
pts[:, 0] = rng.uniform(-37.5, 37.5, n); pts[:, 1] = rng.uniform(-23.5, 23.5, n)
pts[:, 2] = synth.terrain_height(pts[:, 0], pts[:, 1]) + rng.uniform(-2.0, 2.0, n)
sc.points = pts
run("wave_per_block", sc)
sc2 = S.Scene(90.0, 70.0, 0.5, 30000, seed=46)
sc2.points[:, 2] -= 400.0        # heights of both signs: the sums cancel
run("mixed_signs_r2", sc2, radius=2)
# OrthoFromPcl: the same gather on intensities (values mode), written to the ortho layer
sc3 = S.Scene(120.0, 90.0, 0.5, 60000, seed=47)
inten = (np.arange(sc3.points.shape[0]) * 7919 % 251).astype(np.int32)
g = sc3.grid
with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)) as m:
    A.OrthoFromPcl(A.OrthoFromPclSettings()).process(sc3.points, inten, m)
    out["from_pcl_intensities"] = m.get("ortho")
"""


def _child(env_extra, tmp_path, tag):
    path = str(tmp_path / ("elev_%s.npz" % tag))
    code = _CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "scenes": _SCENES}
    from conftest import tuning_env
    env = tuning_env(**env_extra)
    r = subprocess.run([sys.executable, "-c", code, path], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return dict(np.load(path))


def test_canonical_arithmetic_in_every_cell_gives_the_same_floats(tmp_path):
    normal = _child({}, tmp_path, "normal")
    canon = _child({"dsm_canon_all": 1}, tmp_path, "canon")
    assert sorted(normal) == sorted(canon) and len(normal) == 6
    for name in normal:
        a, b = normal[name], canon[name]
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        ok = ~np.isnan(a)
        assert ok.any(), name
        bad = int((a[ok].view(np.uint32) != b[ok].view(np.uint32)).sum())
        assert bad == 0, "%s: %d of %d cells differ between the product-form quotient that passed " \
                         "round_is_certain and the double-double sums" % (name, bad, int(ok.sum()))


def test_canonical_arithmetic_matches_the_oracle():
    """canonical_search sums the reference's own terms (z / d2, 1 / d2 with true divisions,
    dsm.cc:166-168): against the oracle it may differ only where the kd-tree's summation order
    itself decided a rounding."""
    code = (_CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"),
                      "scenes": 'run("sparse_ladder", S.Scene(300.0, 200.0, 1.0, 66000, seed=42))\n'
                                'run("dense_quarter", S.Scene(60.0, 45.0, 0.25, int(8 * 68 * 68), seed=43, '
                                'point_extent=34.0))\n'})
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "e.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, AMHIP_TUNING="dsm_canon_all"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        got = dict(np.load(path))
    for name, sc in (("sparse_ladder", S.Scene(300.0, 200.0, 1.0, 66000, seed=42)),
                     ("dense_quarter", S.Scene(60.0, 45.0, 0.25, int(8 * 68 * 68), seed=43, point_extent=34.0))):
        rc, want, _ = O.dsm_process(sc.points, sc.grid)
        assert rc == O.OK
        frac = S.assert_dsm_close(got[name], want, tol=1e-6)
        assert frac >= 0.9999, (name, frac)


@pytest.mark.parametrize("seed", range(12))
def test_two_runs_and_a_permuted_cloud_give_the_same_bits(seed):
    """Random scenes across the gather's routines (LDS tiles of every capacity class, the
    wave-per-block kernel of dense clouds, the ladder of sparse ones, heights of both signs, radii
    1 .. 9): the default mode's map is a function of the point SET."""
    import torch
    import aerial_mapper_amd as A
    rng = np.random.default_rng(4100 + seed)
    res = float(rng.choice([0.2, 0.25, 0.5, 1.0]))
    rows, cols = int(rng.integers(200, 700)), int(rng.integers(150, 500))
    lx, ly = rows * res, cols * res
    dens = float(rng.choice([0.03, 0.3, 0.6, 2.5, 7.0, 14.0]))       # points per cell
    n = max(1000, min(int(dens * rows * cols), 2_500_000))
    radius = int(rng.choice([1, 1, 1, 2, 4, 9]))
    pts = np.empty((n, 3))
    pts[:, 0] = rng.uniform(-lx / 2 - 2.0, lx / 2 + 2.0, n)
    pts[:, 1] = rng.uniform(-ly / 2 - 2.0, ly / 2 + 2.0, n)
    pts[:, 2] = synth.terrain_height(pts[:, 0], pts[:, 1]) + rng.uniform(-1.0, 1.0, n)
    if seed % 3 == 1:
        pts[:, 2] -= 400.0                                            # both signs
    if seed % 4 == 2:
        pts = pts[np.abs(pts[:, 0]) > 0.05 * lx]                      # a strip without points
    dev = torch.device("cuda", 0)
    cloud = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, lx, ly, res)) as m:
        m.set_dsm_precision(True)
        dsm = A.Dsm(A.DsmSettings(radius), m)
        maps = []
        for k in range(3):
            src = cloud if k < 2 else cloud[torch.randperm(cloud.shape[0], device=dev)].contiguous()
            m.reset()
            dsm.process(src, m)
            maps.append(m.get("elevation").view(np.uint32).copy())
    assert (~np.isnan(maps[0].view(np.float32))).any()
    assert np.array_equal(maps[0], maps[1]), int((maps[0] != maps[1]).sum())
    assert np.array_equal(maps[0], maps[2]), int((maps[0] != maps[2]).sum())


def test_async_call_orders_torchs_stream_behind_the_maps_own():
    """Dsm.process(sync=False) on a map that runs on its OWN stream: torch's current stream is made to
    wait for the call on the device (amhip_ctx_order_after: an event, no host wait), so a torch read
    of the layer enqueued right behind the call sees the finished DSM -- and refilling the cloud
    tensor there cannot race with the gather."""
    import torch
    import aerial_mapper_amd as A
    sc = S.Scene(400.0, 300.0, 0.25, int(1.3e6), seed=77)
    rc, want, _ = O.dsm_process(sc.points, sc.grid)
    assert rc == O.OK
    g = sc.grid
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)) as m:
        m.set_dsm_precision(True)
        pts = torch.from_numpy(sc.points).to("cuda:0")
        layer = m.as_torch("elevation")          # (a view of the device layer; torch reads it on ITS stream)
        torch.cuda.synchronize()
        A.Dsm(A.DsmSettings(), m).process(pts, m, sync=False)
        got = layer.clone()                      # torch's stream: ordered behind the call by the event
        pts.fill_(float("nan"))                  # ... and so is this overwrite of the input
        torch.cuda.synchronize()
        m.synchronize()
        S.assert_dsm_close(got.cpu().numpy(), want, tol=1e-6)
        S.assert_dsm_close(m.get("elevation"), want, tol=1e-6)
