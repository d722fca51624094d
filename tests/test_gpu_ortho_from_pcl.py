"""GPU: ortho::OrthoFromPcl against the oracle."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu


def _run(pts, inten, g, radius, adaptive, device=False):
    import aerial_mapper_amd as A
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    with A.AerialGridMap(st) as m:
        mosaic = A.OrthoFromPcl(A.OrthoFromPclSettings(interpolation_radius=radius,
                                                      use_adaptive_interpolation=adaptive))
        if device:
            import torch
            mosaic.process(torch.from_numpy(pts).cuda(), torch.from_numpy(inten).cuda(), m)
        else:
            mosaic.process(pts, inten, m)
        return m.get("ortho")


def _check(got, want):
    # untouched cells keep the layer's init value (255) on both sides
    assert got.shape == want.shape
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    assert err <= 1e-4, err
    return float((got.view(np.uint32) == want.view(np.uint32)).mean())


@pytest.mark.parametrize("res,n,radius", [(1.0, 6000, 2), (0.5, 30000, 2), (0.25, 40000, 1),
                                          (1.0, 2500, 10)])
def test_from_pcl_matches_oracle(res, n, radius):
    g = O.make_grid(90.0, 70.0, res, 4.0, -3.0)
    pts = synth.make_points(n, 52.0, 90 + n % 7, center=(4.0, -3.0))
    inten = ((np.arange(n) * 37) % 256).astype(np.int32)
    rc, want = O.ortho_from_pcl(pts, inten, g, radius, False)
    assert rc == O.OK
    got = _run(pts, inten, g, radius, False)
    assert _check(got, want) > 0.99
    dev = _run(pts, inten, g, radius, False, device=True)
    _check(dev, want)


def test_from_pcl_exact_hit_takes_the_points_value():
    g = O.make_grid(30.0, 20.0, 1.0)
    pts = synth.make_points(900, 18.0, 95)
    inten = ((np.arange(900) * 11) % 200).astype(np.int32)
    x, y = O.cell_position(g, 7, 5)
    pts[13, :2] = (x, y)
    inten[13] = 251
    rc, want = O.ortho_from_pcl(pts, inten, g, 2, False)
    assert rc == O.OK and want[5, 7] == 251.0
    got = _run(pts, inten, g, 2, False)
    assert got[5, 7] == 251.0
    _check(got, want)


def test_from_pcl_adaptive_fills_every_cell():
    # sparse cloud in one corner: the retries x10, x100, ... reach every cell
    g = O.make_grid(64.0, 48.0, 1.0)
    pts = synth.make_points(400, 12.0, 96, center=(-20.0, -12.0))
    inten = ((np.arange(400) * 53) % 256).astype(np.int32)
    rc, want = O.ortho_from_pcl(pts, inten, g, 2, True)
    assert rc == O.OK
    rc, plain = O.ortho_from_pcl(pts, inten, g, 2, False)
    assert (plain == 255.0).sum() > 1000          # without the retries most cells stay untouched
    got = _run(pts, inten, g, 2, True)
    assert _check(got, want) > 0.99


@pytest.mark.parametrize("seed", list(range(12)))
def test_from_pcl_random_configurations(seed):
    """Random grids / radii / densities (incl. patches dense enough for the gather's higher
    capacity classes and the wave-per-cell kernel), plain and adaptive."""
    rng = np.random.default_rng(8800 + seed)
    res = float(rng.choice([0.25, 0.5, 1.0, 2.0]))
    cx, cy = int(rng.integers(40, 260)), int(rng.integers(40, 200))
    lx, ly = cx * res, cy * res
    g = O.make_grid(lx, ly, res, float(rng.uniform(-500, 500)), float(rng.uniform(-500, 500)))
    radius = int(rng.choice([1, 2, 2, 4, 10]))
    adaptive = bool(rng.integers(0, 2))
    n0 = int(rng.integers(2000, 60000))
    parts = [np.c_[rng.uniform(g.pos_x - lx / 2 - 2, g.pos_x + lx / 2 + 2, n0),
                   rng.uniform(g.pos_y - ly / 2 - 2, g.pos_y + ly / 2 + 2, n0)]]
    if seed % 2 == 0:        # a dense patch
        w, h = rng.uniform(0.05, 0.2) * lx, rng.uniform(0.05, 0.2) * ly
        x0 = rng.uniform(g.pos_x - lx / 2, g.pos_x + lx / 2 - w)
        y0 = rng.uniform(g.pos_y - ly / 2, g.pos_y + ly / 2 - h)
        nk = int(rng.integers(20000, 120000))
        parts.append(np.c_[rng.uniform(x0, x0 + w, nk), rng.uniform(y0, y0 + h, nk)])
    xy = np.concatenate(parts)
    if seed % 3 == 0:        # a strip without points
        xy = xy[np.abs(xy[:, 1] - g.pos_y) > 0.08 * ly]
    n = xy.shape[0]
    pts = np.empty((n, 3))
    pts[:, :2] = xy
    pts[:, 2] = synth.terrain_height(xy[:, 0], xy[:, 1])
    inten = rng.integers(0, 256, n).astype(np.int32)
    rc, want = O.ortho_from_pcl(pts, inten, g, radius, adaptive)
    assert rc == O.OK
    got = _run(pts, inten, g, radius, adaptive, device=bool(seed % 2))
    assert _check(got, want) > 0.98
