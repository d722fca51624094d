"""GPU against the CONSISTENCY-CHECK build of the reference's loop sources: the text of dsm.cc,
ortho-backward-grid.cc and ortho-from-pcl.cc compiled unchanged over the builder-written stand-in
headers of oracle/refkit/ (oracle/_ref/libref_loops_*.so -- built where /root/reference exists, they
travel to the GPU box with the other built libraries).  NOT a reference build (oracle/refkit/refkit.h)
and no pin: one more route by which a mis-read loop in the restated oracle AND in the kernels
would show.  The same bars as against the restated oracle: DSM heights within 1e-4 m with the same
NaN pattern, mosaic layers bit for bit."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not O.have_loops(), reason="oracle/_ref/libref_loops_*.so not built")]

LAYERS = ["elevation_angle", "observation_index", "num_observations", "ortho", "colored_ortho"]


def _settings(g):
    import aerial_mapper_amd as A
    return A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)


@pytest.mark.parametrize("mode", ["default", "fast"])
@pytest.mark.parametrize("res,n,radius,ce,cn", [(0.5, 40000, 1, 0.0, 0.0), (0.25, 90000, 1, 2.5, -1.0),
                                               (1.0, 9000, 4, 0.0, 0.0)])
def test_dsm_equals_the_references_own_process(res, n, radius, ce, cn, mode):
    import aerial_mapper_amd as A
    sc = S.Scene(110.0, 80.0, res, n, seed=300 + radius, point_extent=62.0)
    pts = np.ascontiguousarray(sc.points[np.abs(sc.points[:, 1] - 11.0) > 3.5])   # a gap: ladder, NaN
    g = sc.grid
    rc, want, _ = O.dsm_process(pts, g, radius, ce, cn, which="loops")
    assert rc == O.OK
    with A.AerialGridMap(_settings(g)) as m:
        if mode == "fast":
            m.set_dsm_precision(False)          # (opt-in; new maps start reference-identical)
        A.Dsm(A.DsmSettings(radius, center_easting=ce, center_northing=cn), m).process(pts, m)
        got = m.get("elevation")
    # identical NaN pattern; default mode: the reference's floats (>= 99.9 % bit for bit, the rest
    # one rounding of the double sum away); fast mode: the contract's 1e-4 m
    same = S.assert_dsm_close(got, want, tol=1e-4 if mode == "fast" else 1e-6)
    if mode == "default":
        assert same >= 0.999, same


@pytest.mark.parametrize("colored,kw", [(False, dict()),
                                        (True, dict(distortion=O.DIST_RADTAN, dist=(-0.12, 0.03, 0.002, -0.001))),
                                        (False, dict(distortion=O.DIST_EQUIDISTANT, dist=(0.02, -0.01, 0.004, -0.001)))])
def test_mosaic_equals_the_references_own_process(colored, kw):
    import aerial_mapper_amd as A
    sc = S.Scene(100.0, 80.0, 0.5, 36000, seed=310 + int(colored), cam=S.camera(**kw), colored=colored,
                 num_frames=12, tilt_deg=10.0)
    g = sc.grid
    rc, elev, _ = O.dsm_process(sc.points, g, which="loops")
    assert rc == O.OK
    want = O.new_layers(g)
    want["elevation"] = elev.copy()
    cam = sc.cam
    nc = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height,
                   distortion=cam.distortion, dist=tuple(cam.dist))
    with A.AerialGridMap(_settings(g)) as m:
        m.set("elevation", elev)
        mosaic = A.OrthoBackwardGrid(nc, A.OrthoSettings(colored_ortho=colored), m)
        for lo, hi in ((0, 7), (7, 12)):        # a batch, then another appended
            assert O.ortho_process(g, cam, sc.poses[lo:hi], sc.T_C_B, sc.frames[lo:hi], want,
                                   colored=colored, which="loops") == O.OK
            mosaic.process(sc.poses[lo:hi], sc.frames[lo:hi], m)
        got = {n: m.get(n) for n in LAYERS}
    names = [n for n in LAYERS if n != ("ortho" if colored else "colored_ortho")]
    # (equidistant cameras too since round 4: the device's atan is the correctly rounded value,
    # amhip_atan_cr.h, which the host's libm returns in 99.9 % of its calls -- tests/test_atan_cr.py;
    # the one call in a thousand where glibc is an ulp off would still have to meet a pixel or
    # image-box boundary to that ulp to show)
    S.assert_layers_equal(got, want, names)
    assert (~np.isnan(want["observation_index"])).mean() > 0.3


def test_from_pcl_equals_the_references_own_process():
    import aerial_mapper_amd as A
    g = O.make_grid(90.0, 70.0, 0.5, 4.0, -3.0)
    n = 30000
    pts = synth.make_points(n, 52.0, 93, center=(4.0, -3.0))
    inten = ((np.arange(n) * 37) % 256).astype(np.int32)
    rc, want = O.ortho_from_pcl(pts, inten, g, 2, False, which="loops")
    assert rc == O.OK
    with A.AerialGridMap(_settings(g)) as m:
        A.OrthoFromPcl(A.OrthoFromPclSettings(interpolation_radius=2)).process(pts, inten, m)
        got = m.get("ortho")
    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 1e-4


@pytest.mark.parametrize("incremental,colored", [(False, False), (False, True), (True, False)])
def test_forward_mosaic_equals_the_references_own_flow(incremental, colored):
    """ortho-forward-homography.cc compiled unchanged (over the oracle's restatements of the
    OpenCV / aslam operations): the mosaic it hands to cv::imwrite, bit for bit."""
    import aerial_mapper_amd as A
    rng = np.random.default_rng(31 + 2 * int(incremental) + int(colored))
    cam = S.camera(192, 108, 140.0)
    desc = O.mosaic_desc(300, 220, 400.0, (2.0, -3.0, 0.0))
    T_C_B = (0.2, -0.1, 0.05, 0.9987502603949663, 0.0, 0.049979169270678331, 0.0)
    poses = synth.make_lawnmower_poses(9, 45.0, 470.0, 6, tilt_deg=8.0)
    shape = (108, 192, 3) if colored else (108, 192)
    frames = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(9)]
    ref = O.ReferenceForwardMosaic(cam, desc, T_C_B)
    nc = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height, cam.distortion,
                   tuple(cam.dist), T_C_B)
    st = A.OrthoForwardHomographySettings(
        ground_plane_elevation_m=desc.ground_plane_elevation_m,
        width_mosaic_pixels=desc.width_mosaic_pixels,
        height_mosaic_pixels=desc.height_mosaic_pixels, origin=tuple(desc.origin))
    with A.OrthoForwardHomography(nc, st) as mosaic:
        if incremental:
            for k in range(9):
                assert ref.update(poses[k], frames[k]) == O.OK
                mosaic.updateOrthomosaic(poses[k], frames[k])
                assert np.array_equal(mosaic.result()[0], ref.result), k
        else:
            assert ref.batch(poses, frames) == O.OK
            mosaic.batch(poses, frames)
            assert np.array_equal(mosaic.result()[0], ref.result)
    assert (ref.result != 0).mean() > 0.05
