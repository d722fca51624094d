"""GPU: the three builds of the backward-grid kernel -- k_ortho_backward (every pair in
the reference's arithmetic), k_ortho_backward_fast and k_ortho_backward_fast4 (the
margin-guarded fold of amhip_ortho_fold.h at 3 / 4 waves per SIMD) -- must all give the
oracle's layers, bit for bit, on the scenes that stress the margins: engineered exact
ties, image-border hits, grazing / backward views, UTM magnitudes, incremental batches
replayed onto their own result, NaN / infinite elevations, non-zero num_observations.
(The default build is exercised by every other ortho test of the suite.)"""
import math

import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu

LAYERS = ["elevation_angle", "observation_index", "num_observations", "ortho", "colored_ortho"]
VARIANTS = [{"ortho_exact_fold": "1"}, {}]
IDS = ["exact", "fast4"]


def run_gpu(g, cam, batches, elevation, nobs0=None, angle0=None):
    """batches: list of (poses, frames).  -> (GPU layers, oracle layers)"""
    import aerial_mapper_amd as A
    want = O.new_layers(g)
    want["elevation"][...] = elevation
    if nobs0 is not None:
        want["num_observations"][...] = nobs0
    if angle0 is not None:
        want["elevation_angle"][...] = angle0
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    with A.AerialGridMap(st) as m:
        assert (m.rows, m.cols) == (g.rows, g.cols)
        m.set("elevation", want["elevation"])
        if nobs0 is not None:
            m.set("num_observations", want["num_observations"])
        if angle0 is not None:
            m.set("elevation_angle", want["elevation_angle"])
        nc = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height)
        mosaic = A.OrthoBackwardGrid(nc, A.OrthoSettings(), m)
        for poses, frames in batches:
            assert O.ortho_process(g, cam, poses, synth.IDENTITY_POSE, frames, want) == O.OK
            mosaic.process(poses, frames, m)
        got = {n: m.get(n) for n in LAYERS}
    return got, want


def terrain(g, seed, nan_frac=0.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    j, i = np.meshgrid(np.arange(g.cols), np.arange(g.rows), indexing="ij")
    x = g.pos_x + g.length_x / 2 - g.resolution * (i + 0.5)
    y = g.pos_y + g.length_y / 2 - g.resolution * (j + 0.5)
    z = (synth.terrain_height(x, y) + rng.uniform(-0.5, 0.5, size=x.shape)).astype(np.float32)
    if nan_frac:
        z[rng.uniform(size=z.shape) < nan_frac] = np.nan
    return z


def frames_for(F, cam, salt):
    return [np.ascontiguousarray(f) for f in synth.make_frames(F, cam.height, cam.width, 1, salt=salt)]


@pytest.mark.parametrize("env", VARIANTS, ids=IDS)
def test_ties_borders_and_odd_elevations(tuning, env):
    for k, v in env.items():
        tuning(**{k: v})

    # (1) duplicate and minutely perturbed poses: exact and near ties
    g = O.make_grid(100.0, 70.0, 0.5)
    cam = S.camera()
    base = synth.make_lawnmower_poses(3, 10.0, 500.0, 77, tilt_deg=6.0)
    poses = [base[0], base[0].copy(), base[1], base[1].copy()]
    for k, eps in enumerate([1e-9, 3e-8, 1e-7, 4e-7, 1e-6, 3e-6]):
        p = base[2].copy()
        p[2] += eps * 500.0 * (1 if k % 2 else -1)
        poses.append(p)
    poses.append(base[0].copy())
    poses = np.array(poses)
    z = terrain(g, 5, nan_frac=0.03)
    z[3, 5] = np.inf
    z[7, 11] = -np.inf
    z[9, 2] = 3.0e38
    got, want = run_gpu(g, cam, [(poses, frames_for(len(poses), cam, 3))], z,
                        nobs0=np.full((g.cols, g.rows), 0.75, np.float32))
    S.assert_layers_equal(got, want, LAYERS)
    assert (want["num_observations"] > 0.75).any()

    # (2) flat ground, level camera whose pixel grid puts whole rows / columns of cell
    # centres exactly on u = 0, u = W, v = 0, v = H; sideways and upward looking cameras
    res = 0.5
    g = O.make_grid(160.0 * res, 96.0 * res, res)
    cam = O.Camera()
    cam.fu = cam.fv = 100.0
    cam.cu, cam.cv = 32.0, 24.0
    cam.width, cam.height = 64, 48
    cam.distortion = O.DIST_NONE
    q_down = synth._qmul(synth._axis_angle((0, 0, 1.0), 0.0), synth._axis_angle((1.0, 0, 0), math.pi))
    q_side = synth._qmul(q_down, synth._axis_angle((1.0, 0, 0), math.pi / 2))
    q_up = synth._qmul(q_down, synth._axis_angle((1.0, 0, 0), math.pi))
    poses = [[x, y, 50.0] + list(q_down) for (x, y) in [(0.25, 0.25), (0.0, 0.0), (3.25, -2.25), (0.25, 0.25)]]
    poses += [[0.0, 0.0, 0.0] + list(q_side), [5.0, 5.0, 50.0] + list(q_side), [0.0, 0.0, 50.0] + list(q_up)]
    poses = np.array(poses, np.float64)
    got, want = run_gpu(g, cam, [(poses, frames_for(len(poses), cam, 1))],
                        np.zeros((g.cols, g.rows), np.float32))
    S.assert_layers_equal(got, want, LAYERS)
    assert (~np.isnan(want["observation_index"])).mean() > 0.05


@pytest.mark.parametrize("env", VARIANTS, ids=IDS)
def test_utm_flight_in_batches_and_replay(tuning, env):
    for k, v in env.items():
        tuning(**{k: v})
    c = (464980.25, 5272690.5)
    g = O.make_grid(150.0, 110.0, 0.5, c[0], c[1])
    cam = S.camera()
    F = 15
    poses = synth.make_lawnmower_poses(F, 50.0, 640.0, 21, tilt_deg=12.0, center=c)
    fr = frames_for(F, cam, 4)
    z = terrain(g, 8, nan_frac=0.01)
    cuts = [(0, 4), (4, 5), (5, 11), (11, 15), (5, 11)]       # the last batch replays an earlier one
    got, want = run_gpu(g, cam, [(poses[a:b], fr[a:b]) for a, b in cuts], z)
    S.assert_layers_equal(got, want, LAYERS)
    assert (~np.isnan(want["observation_index"])).mean() > 0.5
    # layer angles no asin can beat, NaN in the layer
    weird = want["elevation_angle"].copy()
    weird[::3, ::2] = np.float32(1.5707964)
    weird[1::3, ::2] = np.float32(2.0)
    weird[2::3, 1::2] = np.nan
    got, want = run_gpu(g, cam, [(poses[0:6], fr[0:6])], z, angle0=weird)
    S.assert_layers_equal(got, want, LAYERS)


@pytest.mark.parametrize("origin", [(0.0, 0.0), (464980.25, 5272690.5), (-2.1e7, 3.3e7)],
                         ids=["local", "utm", "3e7"])
def test_fast_and_exact_kernels_agree_on_a_large_map(tuning, origin):
    """Differential test at a scale the CPU oracle cannot reach in a unit test: 16 M cells x
    249 frames (the bench geometry), random terrain with holes, three coordinate magnitudes;
    the margin-guarded kernel (with and without frame-list pruning) against the kernel that
    does every pair in the reference's arithmetic -- itself oracle-checked above and in
    tests/test_gpu_parity.py."""
    import torch
    import aerial_mapper_amd as A
    side, res, F, W, H = 4000, 0.25, 249, 1920, 1080
    L = side * res
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    elev = 400.0 + 10.0 * torch.rand((side, side), device=dev, generator=g)
    elev += 30.0 * torch.sin(torch.arange(side, device=dev) * 0.01)[None, :]
    elev[torch.rand((side, side), device=dev, generator=g) < 0.01] = float("nan")
    elev = elev.float().cpu().numpy()
    frames = synth.make_frames_torch(F, H, W, 1, 78, dev)
    poses = synth.make_lawnmower_poses(F, L / 2.0 * 1.3, 700.0, 79, tilt_deg=7.0, center=origin)
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    results = {}
    for name, env in (("exact", {"ortho_exact_fold": "1", "ortho_no_prune": "1"}),
                      ("fast", {}), ("fast_noprune", {"ortho_no_prune": "1"})):
        for k in ("ortho_exact_fold", "ortho_no_prune"):
            tuning(**{k: None})
        for k, v in env.items():
            tuning(**{k: v})
        with A.AerialGridMap(A.GridMapSettings(origin[0], origin[1], L, L, res)) as m:
            m.set("elevation", elev)
            mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
            mosaic.process(poses[:100], frames[:100], m)        # two batches: the second one
            mosaic.process(poses[100:], frames[100:], m)        # continues from the layers
            results[name] = {n: m.get(n) for n in LAYERS}
    want = results["exact"]
    assert (~np.isnan(want["observation_index"])).mean() > 0.9
    for name in ("fast", "fast_noprune"):
        S.assert_layers_equal(results[name], want, LAYERS)


@pytest.mark.parametrize("kind,dist", [
    ("radtan", (-0.28, 0.07, 2e-4, -1e-4)),
    ("radtan", (0.12, -0.02, -8e-4, 6e-4)),            # pincushion
    ("radtan", (-0.45, 0.0, 0.0, 0.0)),                # folds back inside the cone
    ("equidistant", (-0.01, 0.02, -0.005, 0.001)),
    ("equidistant", (0.08, -0.03, 0.0, 0.0)),
])
def test_distorted_cameras_prune_and_rectangle_cull_change_nothing(tuning, kind, dist):
    """Cameras with a distortion model on a 9 M-cell rough map, 120 frames: the frame-list
    pruning (inner cone = 'fully visible') and the rectangular outer cull against the plain
    circumscribed-square cull without pruning -- which tests/test_gpu_parity.py checks
    against the oracle.  All layers must be identical."""
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import hip_lib as L
    side, res, F, W, H = 3000, 0.25, 120, 960, 540
    Lm = side * res
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    elev = 400.0 + 8.0 * torch.rand((side, side), device=dev, generator=g)
    elev += 25.0 * torch.sin(torch.arange(side, device=dev) * 0.013)[:, None]
    elev[torch.rand((side, side), device=dev, generator=g) < 0.005] = float("nan")
    elev = elev.float().cpu().numpy()
    frames = synth.make_frames_torch(F, H, W, 1, 6, dev)
    poses = synth.make_lawnmower_poses(F, Lm / 2.0 * 1.2, 650.0, 7, tilt_deg=9.0)
    model = L.DIST_RADTAN if kind == "radtan" else L.DIST_EQUIDISTANT
    ncam = A.NCamera(700.0, 690.0, 470.0, 280.0, W, H, model, dist)
    results = {}
    for name, env in (("plain", {"no_distorted_prune": "1", "distorted_square_cull": "1"}),
                      ("rect", {"no_distorted_prune": "1"}), ("pruned", {})):
        for k in ("no_distorted_prune", "distorted_square_cull"):
            tuning(**{k: None})
        for k, v in env.items():
            tuning(**{k: v})
        with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, Lm, Lm, res)) as m:
            m.set("elevation", elev)
            mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
            mosaic.process(poses[:50], frames[:50], m)
            mosaic.process(poses[50:], frames[50:], m)
            results[name] = {n: m.get(n) for n in LAYERS}
    want = results["plain"]
    assert (~np.isnan(want["observation_index"])).mean() > 0.5
    for name in ("rect", "pruned"):
        S.assert_layers_equal(results[name], want, LAYERS)


def test_small_batches_onto_a_large_map_walk_a_tile_list(tuning):
    """Round 4: a small batch (<= 64 frames) onto a map of >= 16 384 mosaic tiles with materialized
    layers does not dispatch a workgroup per tile: one lane per tile asks the dense launch's own first
    question (can any frame see the tile's bounding sphere?) and a fixed grid walks the list
    (k_ortho_tile_list / k_ortho_backward_fast4_list).  Same layers, bit for bit, as the dense launch
    (tuning knob ortho_no_tile_list), which the oracle tests hold to the reference."""
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth
    side, res = 8320, 1.0                       # 130 x 130 tiles of 64 x 64 cells
    L = side * res
    dev = torch.device("cuda", 0)
    pts = synth.make_points_torch(24_000_000, L / 2.0 + 3.0, 91, dev)
    W, H, F = 320, 240, 40
    frames = synth.make_frames_torch(F, H, W, 1, 92, dev)
    poses = synth.make_lawnmower_poses(F, L / 5.0, 400.0 + 600.0, 92, tilt_deg=6.0)
    ncam = A.NCamera(300.0, 300.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    names = ("elevation_angle", "observation_index", "ortho", "num_observations")

    def run(dense):
        if dense:
            tuning(ortho_no_tile_list=1)
        else:
            tuning(ortho_no_tile_list=None)
        with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
            A.Dsm(A.DsmSettings(), m).process(pts, m)
            mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
            mosaic.process(poses[:8], frames[:8], m)        # (lazily reset layers: the dense launch either way)
            mosaic.process(poses[8:24], frames[8:24], m)    # a batch
            for k in range(24, F):                          # single frames
                mosaic.process(poses[k:k + 1], frames[k:k + 1], m)
            times = None
            m.enable_timing(True)
            m.timing_reset()
            mosaic.process(poses[30:31], frames[30:31], m)
            times = m.kernel_times()["k_ortho_backward"][0]
            return {n: m.get(n) for n in names}, times

    listed, t_list = run(False)
    dense, t_dense = run(True)
    seen = ~np.isnan(listed["observation_index"])
    assert 0.002 < seen.mean() < 0.9
    for n in names:
        a, b = listed[n], dense[n]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.all(), (n, int((~eq).sum()))
    assert t_list < t_dense                                   # (one frame: a few tiles instead of 16 900 workgroups)
