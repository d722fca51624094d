"""GPU, BASELINE.json's full single-GPU size (50 M points -> 10 000 x 10 000
cells @ 0.25 m, 249 frames 1920x1080): size-independent properties the domain
offers, plus an oracle spot check on a corner tile.  ~20 s on an MI355X."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

SIDE, RES, NPTS, F, W, H = 10000, 0.25, 50_000_000, 249, 1920, 1080


# both arithmetic modes of the gather: "exact" = the library's default (FP64, bench.py's headline),
# "fast" = the opt-in single-precision mode
@pytest.fixture(scope="module", params=["exact", "fast"])
def world(request):
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth
    dev = torch.device("cuda", 0)
    L = SIDE * RES
    m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, RES))
    m.set_dsm_precision(request.param == "exact")
    m.mode_name = request.param
    pts = synth.make_points_torch(NPTS, L / 2.0 + 4.0, 143, dev)
    frames = synth.make_frames_torch(F, H, W, 1, 144, dev)
    poses = synth.make_lawnmower_poses(F, L / 2.0, 700.0, 144, tilt_deg=5.0)
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    yield A, m, pts, frames, poses, ncam
    m.close()


def test_full_size_properties(world):
    import torch
    A, m, pts, frames, poses, ncam = world
    dsm = A.Dsm(A.DsmSettings(), m)
    mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)

    dsm.process(pts, m)
    e1 = m.as_torch("elevation").clone()
    assert not torch.isnan(e1).any()                      # 8 pts/m^2: every cell has neighbours
    assert float(e1.min()) > 389.0 and float(e1.max()) < 411.0   # terrain 400 +- 10 (+ noise)

    # idempotence: a second DSM pass over the same cloud rewrites the same heights
    dsm.process(pts, m)
    e2 = m.as_torch("elevation")
    # single-precision mode: the f32 sums move with the order the binning's atomics leave the
    # points in (a float spacing = 3e-5 m at 400 m in a few cells); default mode: the same bits
    # (round_is_certain / canonical_search, tests/test_gpu_determinism.py)
    assert float((e1 - e2).abs().max()) <= 1e-4
    assert float((e1 == e2).float().mean()) > 0.9999
    if m.mode_name == "exact":
        assert torch.equal(e1.view(torch.int32), e2.view(torch.int32))

    # permutation invariance: the DSM is a function of the point SET
    perm = torch.randperm(pts.shape[0], device=pts.device)
    m.reset()
    dsm.process(pts[perm].contiguous(), m)
    e3 = m.as_torch("elevation")
    assert float((e1 - e3).abs().max()) <= 1e-4
    assert float((e1 == e3).float().mean()) > 0.9999
    if m.mode_name == "exact":
        assert torch.equal(e1.view(torch.int32), e3.view(torch.int32))
    del perm

    # mosaic: ranges, coverage, and idempotence of the fold (a second pass over
    # the same frames can not beat the stored angles -> nothing changes)
    mosaic.process(poses, frames, m)
    ang = m.as_torch("elevation_angle").clone()
    idx = m.as_torch("observation_index").clone()
    ort = m.as_torch("ortho").clone()
    seen = ~torch.isnan(idx)
    assert float(seen.float().mean()) > 0.9
    assert float(idx[seen].min()) >= 0 and float(idx[seen].max()) <= F - 1
    assert float(ang.max()) <= 1.5707964 and float(ang[seen].min()) > 0.5
    assert float(ort.min()) >= 0 and float(ort.max()) <= 255
    assert bool((ort[~seen] == 255).all()) and bool((ang[~seen] == 0).all())
    mosaic.process(poses, frames, m)
    assert torch.equal(ang, m.as_torch("elevation_angle"))
    assert torch.equal(ort, m.as_torch("ortho"))
    same_idx = (idx == m.as_torch("observation_index")) | (~seen)
    assert bool(same_idx.all())

    # incremental == batch: folding the frames in two batches gives the same
    # angles and pixels (indices are per batch, ortho-backward-grid.cc:182)
    m.set("elevation_angle", np.zeros((SIDE, SIDE), np.float32))
    m.set("ortho", np.full((SIDE, SIDE), 255.0, np.float32))
    mosaic.process(poses[:100], frames[:100], m)
    mosaic.process(poses[100:], frames[100:], m)
    assert torch.equal(ang, m.as_torch("elevation_angle"))
    assert torch.equal(ort, m.as_torch("ortho"))


def test_full_size_corner_matches_oracle(world):
    A, m, pts, frames, poses, ncam = world
    m.reset()
    A.Dsm(A.DsmSettings(), m).process(pts, m)
    A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m).process(poses, frames, m)
    s = 600
    L = SIDE * RES
    sub_len = s * RES
    c = L / 2.0 - sub_len / 2.0
    g = O.make_grid(sub_len, sub_len, RES, c, c)
    x, y = pts[:, 0], pts[:, 1]
    keep = (x > c - sub_len / 2 - 3) & (y > c - sub_len / 2 - 3)
    sub = pts[keep].cpu().numpy()
    rc, elev, _ = O.dsm_process(sub, g)
    assert rc == O.OK
    got = m.get("elevation")[:s, :s]
    assert np.array_equal(np.isnan(got), np.isnan(elev))
    assert np.abs(got.astype(np.float64) - elev).max() <= (1e-6 if m.mode_name == "exact" else 1e-4)
    same = float((got.view(np.uint32) == elev.view(np.uint32)).mean())
    # default mode: the reference's floats (the order of the double sums may flip a cell on a float
    # rounding boundary: 1 in 1e8); single-precision mode: within a float spacing
    assert same >= (0.999999 if m.mode_name == "exact" else 0.99), same
    layers = O.new_layers(g)
    layers["elevation"] = got.copy()
    cam = O.Camera()
    cam.fu = cam.fv = 1400.0
    cam.cu, cam.cv, cam.width, cam.height = (W - 1) / 2.0, (H - 1) / 2.0, W, H
    rc = O.ortho_process(g, cam, poses, np.array([0, 0, 0, 1, 0, 0, 0.0]),
                         [f for f in frames.cpu().numpy()], layers)
    assert rc == O.OK
    for name in ("elevation_angle", "observation_index", "ortho"):
        a, b = m.get(name)[:s, :s], layers[name]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.all(), (name, int((~eq).sum()))
