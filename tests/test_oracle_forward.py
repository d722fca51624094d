"""CPU tests of the forward-homography oracle (oracle/amo_forward.cc): the
committed golden fixtures and known-answer cases derived from
ortho-forward-homography.cc and the OpenCV pieces it calls (restated; the
reference has no tests for this path -> parity unpinned)."""
import numpy as np
import pytest

import golden_io as G
import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth


def _mosaic_of(d):
    m = d["mosaic"]
    return O.mosaic_desc(int(m[0]), int(m[1]), float(m[2]), [float(v) for v in m[3:6]])


@pytest.mark.parametrize("name", G.names("fwd"))
def test_oracle_reproduces_forward_golden(name):
    d = G.load(name)
    fm = O.ForwardMosaic(G.camera_of(d), _mosaic_of(d), d["T_C_B"])
    frames = d["frames"]
    if bool(d["incremental"]):
        sums = []
        for k in range(frames.shape[0]):
            assert fm.update(d["T_G_B"][k], frames[k]) == O.OK
            sums.append(int(fm.result.astype(np.int64).sum()))
        assert sums == [int(v) for v in d["step_checksums"]]
    else:
        assert fm.batch(d["T_G_B"], [f for f in frames]) == O.OK
    assert np.array_equal(fm.result, d["result"])
    assert np.array_equal(fm.mask, d["mask"])


def _nadir_pose(px, py, h):
    # R_G_C = Rx(pi): camera x -> +x, y -> -y, optical axis -> -z
    return np.array([px, py, h, 0.0, 1.0, 0.0, 0.0])


def test_nadir_homography_known_answer():
    # ray of pixel (u, v) = ((u-cu)/f, (v-cv)/f, 1) -> ground point
    # (px + s rx, py - s ry) with s = h - ground; mosaic (x, y) = (G_y + W/2, G_x + H/2)
    cam = S.camera(96, 54, 70.0)
    desc = O.mosaic_desc(200, 120, 400.0, (1.0, -2.0, 0.0))
    T = _nadir_pose(11.0, 7.0, 470.0)
    rc, M = O.fwd_homography(cam, desc, T, batch_quirk=False)
    assert rc == O.OK
    s = 470.0 - 400.0
    for (u, v) in [(0.0, 0.0), (95.0, 53.0), (47.5, 26.5), (10.0, 40.0)]:
        p = M @ np.array([u, v, 1.0])
        x, y = p[0] / p[2], p[1] / p[2]
        gx = 11.0 + s * (u - cam.cu) / cam.fu - 1.0
        gy = 7.0 - s * (v - cam.cv) / cam.fv + 2.0
        assert abs(x - (gy + 100.0)) < 2e-4 and abs(y - (gx + 60.0)) < 2e-4
    # batch() offsets BOTH axes by width/2 (ortho-forward-homography.cc:155-158)
    rc, Mb = O.fwd_homography(cam, desc, T, batch_quirk=True)
    pb = Mb @ np.array([47.5, 26.5, 1.0])
    p = M @ np.array([47.5, 26.5, 1.0])
    assert abs((pb[1] / pb[2] - p[1] / p[2]) - (100.0 - 60.0)) < 2e-4
    assert abs(pb[0] / pb[2] - p[0] / p[2]) < 2e-4


def test_distance_transform_is_exact_l1():
    rng = np.random.default_rng(3)
    for shape, density in [((23, 31), 0.9), ((40, 17), 0.98), ((12, 12), 0.5)]:
        mask = (rng.uniform(size=shape) < density).astype(np.uint8) * 255
        mask[rng.integers(shape[0]), rng.integers(shape[1])] = 0
        got = O.fwd_distance_l1(mask)
        zy, zx = np.nonzero(mask == 0)
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        want = np.min(np.abs(yy[..., None] - zy) + np.abs(xx[..., None] - zx), axis=-1)
        assert np.array_equal(got, want.astype(np.float32))
    # no zero pixel at all: "infinite" everywhere (weight saturates at 1)
    assert O.fwd_distance_l1(np.full((5, 7), 255, np.uint8)).min() > 1e6


def test_single_frame_feather_weights_and_normalisation():
    # one constant frame seen from a level camera: inside the footprint the
    # weight is min(0.02 * L1 distance to the footprint's outside, 1) and
    # result = (short)((short)(v * w) / (w + 1e-5))
    cam = S.camera(64, 48, 50.0)
    desc = O.mosaic_desc(140, 120, 400.0)
    frame = np.full((48, 64), 200, np.uint8)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(_nadir_pose(0.0, 0.0, 460.0)[None], [frame]) == O.OK
    on = fm.mask > 0
    assert 0.2 < on.mean() < 0.6
    dist = O.fwd_distance_l1((on * 255).astype(np.uint8))
    # the mask of the fed image is exactly where the result is covered
    w = np.minimum(dist * np.float32(0.02), np.float32(1.0)).astype(np.float32)
    fed = (np.float32(200.0) * w).astype(np.int16)
    want = (fed.astype(np.float32) / (w + np.float32(1e-5))).astype(np.int16)
    want[~on] = 0
    want[want <= 0] = 0
    assert np.array_equal(fm.result[..., 0], want)
    assert np.array_equal(fm.result[..., 0], fm.result[..., 1])
    assert np.array_equal(fm.result[..., 0], fm.result[..., 2])


def test_incremental_keeps_previous_result_where_new_frame_is_absent():
    cam = S.camera(64, 48, 50.0)
    desc = O.mosaic_desc(160, 120, 400.0)
    a = np.full((48, 64), 180, np.uint8)
    b = np.full((48, 64), 90, np.uint8)
    fm = O.ForwardMosaic(cam, desc)
    assert fm.update(_nadir_pose(-20.0, 0.0, 460.0), a) == O.OK
    first = fm.result.copy()
    first_mask = fm.mask.copy()
    assert fm.update(_nadir_pose(25.0, 0.0, 460.0), b) == O.OK
    only_first = (first_mask > 0) & (fm.result[..., 0] > 100)
    assert only_first.any()
    # union of both footprints is covered
    assert (fm.mask > 0).sum() > (first_mask > 0).sum()
    # far from the second footprint the first result survives up to the
    # re-normalisation (short)((short)(r * w) / (w + eps)) <= r
    assert (fm.result[..., 0][only_first] <= first[..., 0][only_first]).all()


def test_zero_pixels_inside_a_frame_are_holes():
    # addImage: mask = image > 0.1 -- black pixels of the frame do not count
    cam = S.camera(64, 48, 50.0)
    desc = O.mosaic_desc(140, 120, 400.0)
    frame = np.full((48, 64), 150, np.uint8)
    frame[20:28, 30:40] = 0
    fm = O.ForwardMosaic(cam, desc)
    assert fm.batch(_nadir_pose(0.0, 0.0, 460.0)[None], [frame]) == O.OK
    full = O.ForwardMosaic(cam, desc)
    assert full.batch(_nadir_pose(0.0, 0.0, 460.0)[None], [np.full((48, 64), 150, np.uint8)]) == O.OK
    assert ((full.mask > 0) & (fm.mask == 0)).sum() > 50


def test_l1_distance_transform_equals_scipys_taxicab_transform():
    """The one OpenCV piece of the forward mosaic that CAN be pinned in this image (VERDICT r1
    #5): cv::distanceTransform(DIST_L1, 3) is the exact city-block distance to the nearest zero
    pixel; scipy.ndimage.distance_transform_cdt(metric='taxicab') computes the same quantity
    with an independent implementation.  (getPerspectiveTransform / warpPerspective /
    FeatherBlender stay 'parity unpinned': no OpenCV here.)"""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    for t in range(40):
        h, w = int(rng.integers(3, 140)), int(rng.integers(3, 180))
        m = (rng.random((h, w)) < rng.choice([0.02, 0.3, 0.7, 0.98])).astype(np.uint8) * 255
        if t == 0:
            m[:] = 255
            m[h // 2, w // 3] = 0
        if t == 1:   # footprint-shaped mask: zeros outside a quadrilateral
            yy, xx = np.mgrid[0:h, 0:w]
            m = (((xx + 0.3 * yy) > 0.2 * w) & ((xx - 0.2 * yy) < 0.8 * w) & (yy > 2)).astype(np.uint8) * 255
        if not (m == 0).any():
            continue
        want = ndimage.distance_transform_cdt(m != 0, metric="taxicab").astype(np.float32)
        got = O.fwd_distance_l1(np.ascontiguousarray(m))
        assert np.array_equal(got, want)


# ---- the restated OpenCV / aslam pieces against independent numpy / scipy evaluations of the
# ---- operations they stand for (OpenCV itself is not in the image: what CAN be pinned) ---------
import ctypes as C


def _cv():
    lib = O.lib()
    lib.amo_cv_get_perspective_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.amo_cv_invert3.argtypes = [C.c_void_p, C.c_void_p]
    lib.amo_cv_warp_nearest.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.amo_cv_undistort_image.argtypes = [C.POINTER(O.Camera), C.c_void_p, C.c_size_t, C.c_int,
                                           C.c_void_p]
    return lib


def test_get_perspective_transform_solves_the_four_point_system():
    """cv::getPerspectiveTransform: the homography through four correspondences, M[8] = 1 --
    against numpy.linalg.solve of the same 8 x 8 system, and by mapping the points."""
    lib = _cv()
    rng = np.random.default_rng(11)
    for _ in range(200):
        src = np.array([[0, 0], [1919, 0], [1919, 1079], [0, 1079]], np.float32) + \
            rng.uniform(-40, 40, (4, 2)).astype(np.float32)
        dst = (np.array([[100, 100], [900, 140], [860, 700], [120, 650]]) +
               rng.uniform(-60, 60, (4, 2))).astype(np.float32)
        M = np.zeros(9)
        assert lib.amo_cv_get_perspective_transform(src.ctypes.data, dst.ctypes.data, M.ctypes.data) == O.OK
        A, b = [], []
        for (sx, sy), (dx, dy) in zip(src.astype(np.float64), dst.astype(np.float64)):
            A += [[sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx], [0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy]]
            b += [dx, dy]
        want = np.append(np.linalg.solve(np.array(A), np.array(b)), 1.0)
        np.testing.assert_allclose(M, want, rtol=1e-9, atol=1e-12)
        p = (M.reshape(3, 3) @ np.c_[src.astype(np.float64), np.ones(4)].T).T
        np.testing.assert_allclose(p[:, :2] / p[:, 2:], dst, atol=1e-6)


def test_invert3_is_the_matrix_inverse():
    lib = _cv()
    rng = np.random.default_rng(12)
    for _ in range(200):
        S = rng.standard_normal((3, 3)) + 2.0 * np.eye(3)
        D = np.zeros((3, 3))
        assert lib.amo_cv_invert3(S.ctypes.data, D.ctypes.data) == O.OK
        np.testing.assert_allclose(D, np.linalg.inv(S), rtol=1e-10, atol=1e-12)
    Z = np.zeros((3, 3))
    assert lib.amo_cv_invert3(Z.ctypes.data, Z.copy().ctypes.data) != O.OK


def test_warp_nearest_is_the_inverse_mapped_nearest_pixel():
    """cv::warpPerspective(INTER_NEAREST, BORDER_CONSTANT): dst(x, y) = src(round(M^-1 (x, y, 1)))
    with round-half-even and 0 outside -- against a float64 numpy evaluation; pixels whose source
    coordinate lies within 1e-6 of a rounding tie may legitimately differ (evaluation order)."""
    lib = _cv()
    rng = np.random.default_rng(13)
    for ch in (1, 3):
        sh, sw, h, w = 90, 120, 150, 170
        src = rng.integers(1, 256, (sh, sw, ch), dtype=np.uint8)
        for _ in range(6):
            M = np.array([[1.2, 0.1, 15.0], [-0.08, 1.1, 20.0], [1e-4, -5e-5, 1.0]]) + \
                rng.standard_normal((3, 3)) * [[0.05, 0.05, 5], [0.05, 0.05, 5], [2e-5, 2e-5, 0]]
            dst = np.zeros((h, w, ch), np.uint8)
            assert lib.amo_cv_warp_nearest(src.ctypes.data, src.strides[0], sw, sh, ch,
                                           M.ctypes.data, w, h, dst.ctypes.data) == O.OK
            Mi = np.linalg.inv(M)
            ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
            X = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
            Y = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
            Wd = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
            fx, fy = X / Wd, Y / Wd
            ix, iy = np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)
            inside = (ix >= 0) & (iy >= 0) & (ix < sw) & (iy < sh)
            want = np.zeros_like(dst)
            want[inside] = src[iy[inside], ix[inside]]
            tie = (np.abs(np.abs(fx - np.floor(fx)) - 0.5) < 1e-6) | (np.abs(np.abs(fy - np.floor(fy)) - 0.5) < 1e-6)
            differ = (dst != want).any(axis=2)
            assert not (differ & ~tie).any()
            assert inside.mean() > 0.2


def test_undistort_is_a_bilinear_remap_with_five_fractional_bits():
    """aslam's MappedUndistorter = cv::remap(INTER_LINEAR, BORDER_CONSTANT): coordinates quantised
    to 1/32 pixel, (sum + 2^14) >> 15.  On a smooth image it stays within 1 grey level of
    scipy.ndimage.map_coordinates(order=1) at the exact map; with no distortion it is the identity."""
    from scipy import ndimage
    lib = _cv()
    H, W = 96, 128
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.clip(120 + 60 * np.sin(0.05 * u) * np.cos(0.07 * v) + 0.3 * u, 0, 255).astype(np.uint8)
    cam = O.Camera()
    cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height = 100.0, 100.0, 63.5, 47.5, W, H
    cam.distortion = O.DIST_NONE
    out = np.zeros_like(img)
    assert lib.amo_cv_undistort_image(C.byref(cam), img.ctypes.data, img.strides[0], 1, out.ctypes.data) == O.OK
    assert np.array_equal(out, img)
    cam.distortion = O.DIST_RADTAN
    for k, val in enumerate((-0.12, 0.03, 1e-3, -5e-4)):
        cam.dist[k] = val
    assert lib.amo_cv_undistort_image(C.byref(cam), img.ctypes.data, img.strides[0], 1, out.ctypes.data) == O.OK
    x, y = (u - cam.cu) / cam.fu, (v - cam.cv) / cam.fv
    r2 = x * x + y * y
    k1, k2, p1, p2 = (cam.dist[k] for k in range(4))
    rad = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    mx, my = cam.fu * xd + cam.cu, cam.fv * yd + cam.cv
    want = ndimage.map_coordinates(img.astype(np.float64), [my, mx], order=1, mode="constant", cval=0.0)
    interior = (mx > 1) & (my > 1) & (mx < W - 2) & (my < H - 2)
    assert interior.mean() > 0.8
    assert np.abs(out.astype(np.float64) - want)[interior].max() <= 1.0
