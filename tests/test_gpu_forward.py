"""GPU parity of ortho::OrthoForwardHomography (HIP kernels behind the C ABI,
amhip_mosaic_*) against the CPU oracle (oracle/amo_forward.cc) and the
committed golden fixtures.  The mosaic is integer data (CV_16SC3 + CV_8U mask):
the bar is bit-exact."""
import numpy as np
import pytest

import golden_io as G
import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu


def _A():
    import aerial_mapper_amd as A
    return A


def _ncam(cam, T_C_B=(0, 0, 0, 1, 0, 0, 0)):
    A = _A()
    return A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height, cam.distortion,
                     tuple(cam.dist), T_C_B)


def _settings(desc):
    A = _A()
    return A.OrthoForwardHomographySettings(
        ground_plane_elevation_m=desc.ground_plane_elevation_m,
        width_mosaic_pixels=desc.width_mosaic_pixels,
        height_mosaic_pixels=desc.height_mosaic_pixels, origin=tuple(desc.origin))


def _mosaic_of(d):
    m = d["mosaic"]
    return O.mosaic_desc(int(m[0]), int(m[1]), float(m[2]), [float(v) for v in m[3:6]])


def _assert_same(got, want, what):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%s: %d differing entries, first at %s: %s != %s" % (
            what, bad.shape[0], bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))


@pytest.mark.parametrize("name", G.names("fwd"))
def test_forward_golden(name):
    A = _A()
    d = G.load(name)
    desc = _mosaic_of(d)
    with A.OrthoForwardHomography(_ncam(G.camera_of(d), d["T_C_B"]), _settings(desc)) as mosaic:
        frames = d["frames"]
        if bool(d["incremental"]):
            sums = []
            for k in range(frames.shape[0]):
                mosaic.updateOrthomosaic(d["T_G_B"][k], frames[k])
                sums.append(int(mosaic.result()[0].astype(np.int64).sum()))
            assert sums == [int(v) for v in d["step_checksums"]]
        else:
            mosaic.batch(d["T_G_B"], [f for f in frames])
        res, mask = mosaic.result()
    _assert_same(res, d["result"], name + " result")
    _assert_same(mask, d["mask"], name + " mask")


def test_homography_matches_oracle_bitwise():
    A = _A()
    cam = S.camera(192, 108, 140.0)
    desc = O.mosaic_desc(600, 500, 402.5, (4.0, -3.0, 1.0))
    T_C_B = np.array([0.1, -0.05, 0.02, 0.9998, 0.01, -0.012, 0.008])
    T_C_B[3:] /= np.linalg.norm(T_C_B[3:])
    poses = synth.make_lawnmower_poses(9, 80.0, 520.0, 11, tilt_deg=6.0)
    with A.OrthoForwardHomography(_ncam(cam, T_C_B), _settings(desc)) as mosaic:
        for quirk in (True, False):
            for p in poses:
                T_G_C = O.compose_T_G_C(p[None], T_C_B)[0]
                rc, want = O.fwd_homography(cam, desc, T_G_C, quirk)
                assert rc == O.OK
                got = mosaic.homography(p, batch=quirk)
                assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("colored", [False, True])
def test_batch_larger_scene(colored):
    # 70 frames > the 64-frame chunk the warp/feed kernels take at once
    A = _A()
    cam = S.camera(160, 120, 120.0)
    desc = O.mosaic_desc(420, 380, 400.0)
    F = 70 if not colored else 9
    poses = synth.make_lawnmower_poses(F, 120.0, 520.0, 21, tilt_deg=5.0)
    frames = synth.make_frames(F, 120, 160, 3 if colored else 1, salt=5)
    frames = np.where(frames < 4, 0, frames).astype(np.uint8)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(poses, [f for f in frames]) == O.OK
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        mosaic.batch(poses, [f for f in frames])
        res, mask = mosaic.result()
    assert (fm.mask > 0).mean() > 0.3
    _assert_same(res, fm.result, "result")
    _assert_same(mask, fm.mask, "mask")


def test_device_frames_match_host_frames_and_reset():
    import torch
    A = _A()
    cam = S.camera(128, 96, 100.0)
    desc = O.mosaic_desc(300, 260, 400.0)
    poses = synth.make_lawnmower_poses(10, 60.0, 500.0, 23, tilt_deg=4.0)
    frames = synth.make_frames(10, 96, 128, 1, salt=9)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(poses, [f for f in frames]) == O.OK
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
        mosaic.batch(poses, dev)
        res, mask = mosaic.result()
        _assert_same(res, fm.result, "device frames")
        _assert_same(mask, fm.mask, "device frames mask")
        # a second batch() on the same object keeps blending on top, like the
        # reference's blender_ (never re-prepared by batch()); reset() = new object
        mosaic.reset()
        mosaic.batch(poses[:4], [f for f in frames[:4]])
        res2, _ = mosaic.result()
    fm2 = O.ForwardMosaic(cam, desc)
    assert fm2.batch(poses[:4], [f for f in frames[:4]]) == O.OK
    _assert_same(res2, fm2.result, "after reset")


def test_incremental_sequence_matches_oracle_every_step():
    import torch
    A = _A()
    cam = S.camera(128, 96, 100.0)
    desc = O.mosaic_desc(280, 300, 400.0, (2.0, 1.0, 0.0))
    poses = synth.make_lawnmower_poses(7, 50.0, 500.0, 29, tilt_deg=5.0, center=(2.0, 1.0))
    frames = synth.make_frames(7, 96, 128, 3, salt=13)
    fm = O.ForwardMosaic(cam, desc)
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        for k in range(7):
            assert fm.update(poses[k], frames[k]) == O.OK
            if k % 2:
                mosaic.updateOrthomosaic(poses[k], torch.from_numpy(frames[k].copy()).cuda())
            else:
                mosaic.updateOrthomosaic(poses[k], frames[k])
            res, mask = mosaic.result()
            _assert_same(res, fm.result, "step %d" % k)
            _assert_same(mask, fm.mask, "step %d mask" % k)


@pytest.mark.parametrize("kind", ["radtan", "equidistant"])
def test_distorted_cameras_undistort_then_warp(kind):
    A = _A()
    if kind == "radtan":
        cam = S.camera(128, 96, 95.0, O.DIST_RADTAN, (-0.25, 0.06, 3e-4, -2e-4))
    else:
        cam = S.camera(128, 96, 95.0, O.DIST_EQUIDISTANT, (-0.02, 0.004, -0.001, 0.0002))
    desc = O.mosaic_desc(300, 260, 400.0)
    poses = synth.make_lawnmower_poses(6, 50.0, 500.0, 31, tilt_deg=4.0)
    frames = synth.make_frames(6, 96, 128, 1, salt=17)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(poses, [f for f in frames]) == O.OK
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        mosaic.batch(poses, [f for f in frames])
        res, mask = mosaic.result()
    # libm (sqrt/atan) of the device vs glibc can move a 1/32-pixel remap
    # coordinate by one step in rare pixels: allow a handful of 1-LSB cells
    diff = np.abs(res.astype(np.int32) - fm.result.astype(np.int32))
    assert (diff > 0).mean() < 1e-3 and diff.max() <= 255
    assert (mask != fm.mask).mean() < 1e-3


def test_wide_mosaic_rows_cross_several_wave_chunks():
    # 1500 columns: the row pass of the distance transform carries its state
    # over 24 chunks of 64; tall thin footprints exercise the column pass
    A = _A()
    cam = S.camera(160, 90, 110.0)
    desc = O.mosaic_desc(1500, 200, 400.0)
    poses = synth.make_lawnmower_poses(12, 500.0, 600.0, 37, tilt_deg=5.0, lines=1)
    # batch() puts ground y at mosaic x + W/2 and ground x at mosaic y + W/2 (quirk):
    # fly along northing at easting -650 so that the strip lands in rows ~100
    poses[:, 1] = poses[:, 0].copy()
    poses[:, 0] = -650.0
    frames = synth.make_frames(12, 90, 160, 1, salt=19)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(poses, [f for f in frames]) == O.OK
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        mosaic.batch(poses, [f for f in frames])
        res, mask = mosaic.result()
    assert (fm.mask > 0).mean() > 0.05
    _assert_same(res, fm.result, "result")
    _assert_same(mask, fm.mask, "mask")


def test_argument_errors():
    A = _A()
    cam = S.camera(64, 48, 50.0)
    desc = O.mosaic_desc(100, 80, 400.0)
    with pytest.raises(A.AmhipError):
        A.OrthoForwardHomography(None, _settings(desc))
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        with pytest.raises(A.AmhipError):
            mosaic.batch(np.zeros((2, 7)), [np.zeros((48, 64), np.uint8)])


def test_frames_outside_partially_inside_and_strongly_tilted():
    # frame 0 misses the mosaic completely, frame 1 is cut by its edge, the
    # others are tilted up to ~55 deg (long, thin footprints; a horizon-crossing
    # one makes the kernels fall back to the whole mosaic as their region)
    A = _A()
    cam = S.camera(128, 96, 100.0)
    desc = O.mosaic_desc(320, 320, 400.0)
    poses = synth.make_lawnmower_poses(8, 60.0, 500.0, 41, tilt_deg=55.0)
    poses[0, 0:2] = (900.0, 900.0)
    poses[1, 0:2] = (150.0, -155.0)
    frames = synth.make_frames(8, 96, 128, 1, salt=23)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(poses, [f for f in frames]) == O.OK
    with A.OrthoForwardHomography(_ncam(cam), _settings(desc)) as mosaic:
        mosaic.batch(poses, [f for f in frames])
        res, mask = mosaic.result()
        _assert_same(res, fm.result, "result")
        _assert_same(mask, fm.mask, "mask")
        # the same frames one at a time
        mosaic.reset()
        fi = O.ForwardMosaic(cam, desc)
        for k in range(8):
            assert fi.update(poses[k], frames[k]) == O.OK
            mosaic.updateOrthomosaic(poses[k], frames[k])
        res, mask = mosaic.result()
    assert (fm.mask > 0).mean() > 0.2
    _assert_same(res, fi.result, "incremental result")
    _assert_same(mask, fi.mask, "incremental mask")


@pytest.mark.parametrize("seed", range(10))
def test_forward_random_configurations(seed):
    # randomized sweep: image / mosaic sizes (incl. non-square, smaller than a
    # warp block), focal length, altitude, tilt, origin, T_C_B, colour, sparse
    # frames (many zero pixels = mask holes), batch vs incremental
    rng = np.random.default_rng(5000 + seed)
    A = _A()
    W, H = int(rng.integers(24, 160)), int(rng.integers(20, 120))
    cam = S.camera(W, H, float(rng.uniform(0.6, 1.5) * W))
    mw, mh = int(rng.integers(9, 420)), int(rng.integers(9, 360))
    origin = (float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50)), float(rng.uniform(-5, 5)))
    desc = O.mosaic_desc(mw, mh, 400.0 + float(rng.uniform(-10, 10)), origin)
    F = int(rng.integers(1, 24))
    colored = bool(seed % 2)
    incremental = seed % 3 == 0
    poses = synth.make_lawnmower_poses(F, 0.4 * min(mw, mh), 400.0 + float(rng.uniform(20, 300)),
                                       6000 + seed, tilt_deg=float(rng.choice([3.0, 15.0, 45.0])),
                                       center=(origin[0], origin[1]))
    if not incremental:   # batch() offsets both axes by width/2: keep the strip in view
        poses[:, 0] += (mw - mh) / 2.0
    frames = synth.make_frames(F, H, W, 3 if colored else 1, salt=seed)
    frames = np.where(frames < int(rng.choice([1, 40, 160])), 0, frames).astype(np.uint8)
    T_C_B = np.r_[rng.uniform(-0.3, 0.3, 3), rng.normal(size=4) * 0.03 + np.array([1.0, 0, 0, 0])]
    T_C_B[3:] /= np.linalg.norm(T_C_B[3:])
    fm = O.ForwardMosaic(cam, desc, T_C_B)
    with A.OrthoForwardHomography(_ncam(cam, T_C_B), _settings(desc)) as mosaic:
        if incremental:
            for k in range(F):
                assert fm.update(poses[k], frames[k]) == O.OK
                mosaic.updateOrthomosaic(poses[k], frames[k])
        else:
            assert fm.batch(poses, [f for f in frames]) == O.OK
            mosaic.batch(poses, [f for f in frames])
        res, mask = mosaic.result()
    _assert_same(res, fm.result, "result")
    _assert_same(mask, fm.mask, "mask")
