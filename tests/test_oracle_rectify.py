"""stereo::Rectifier::rectifyStereoPair (aerial_mapper_dense_pcl/src/rectifier.cpp:34-128,
SURVEY 8f rank 3): the restated oracle (oracle/amo_rectify.cc) against the reference's own
rectifier.cpp compiled unchanged over oracle/refkit (flow pinned; the Eigen / OpenCV arithmetic
underneath is the oracle's adopted definition, amo_rectify.h: parity unpinned for it), and
against properties any correct planar rectification has."""
import numpy as np
import pytest

import oracle_ffi as O


def rig(seed, W=160, H=120, yaw=0.03, base=(6.0, 0.7, -0.4)):
    """Two nadir-looking cameras ~80 m above ground, a baseline mostly along x."""
    rng = np.random.default_rng(seed)
    K = np.array([[150.0, 0.0, (W - 1) / 2.0], [0.0, 150.0, (H - 1) / 2.0], [0.0, 0.0, 1.0]])

    def rot(rx, ry, rz):
        cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        return Rz @ Ry @ Rx
    down = rot(np.pi, 0.0, 0.0)                    # camera z-axis pointing at the ground
    R1 = rot(*(rng.normal(0, 0.02, 3))) @ down @ rot(0, 0, yaw)
    R2 = rot(*(rng.normal(0, 0.02, 3))) @ down @ rot(0, 0, -yaw)
    t1 = np.array([10.0, -4.0, 80.0])
    t2 = t1 + np.array(base)
    yy, xx = np.mgrid[0:H, 0:W]
    left = ((np.sin(xx * 0.21) + np.cos(yy * 0.17)) * 60 + 128 + rng.integers(-9, 9, (H, W))).clip(0, 255)
    right = ((np.sin(xx * 0.19 + 1.0) + np.cos(yy * 0.23)) * 60 + 128 + rng.integers(-9, 9, (H, W))).clip(0, 255)
    return K, R1, R2, t1, t2, left.astype(np.uint8), right.astype(np.uint8)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_restated_rectifier_equals_the_references_own_code(seed):
    if not O.have_loops():
        pytest.skip("oracle/_ref/libref_loops_rectify.so not built (needs /root/reference)")
    args = rig(seed, W=200 + 8 * seed, H=130 + seed)
    rc_a, a = O.rectify_stereo_pair(*args, which="port")
    rc_b, b = O.rectify_stereo_pair(*args, which="loops")
    assert rc_a == rc_b == O.OK
    assert a["baseline"] == b["baseline"] and np.array_equal(a["R_G_C"], b["R_G_C"])
    assert np.array_equal(a["maps"].view(np.uint32), b["maps"].view(np.uint32))
    for n in ("left", "right", "mask"):
        assert np.array_equal(a[n], b[n]), n


def test_rectification_properties():
    K, R1, R2, t1, t2, left, right = rig(7)
    rc, r = O.rectify_stereo_pair(K, R1, R2, t1, t2, left, right)
    assert rc == O.OK
    R = r["R_G_C"]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)           # a rotation
    b = t2 - t1
    assert abs(r["baseline"] - np.linalg.norm(b)) < 1e-12
    assert np.allclose(R[0], b / np.linalg.norm(b), atol=1e-12)  # its x axis is the baseline
    # epipolar lines are image rows: a world point projects to the same rectified row in both
    # views.  Check through the maps: rectified pixel (u, v) of view k comes from original
    # pixel maps_k(u, v); a ground point seen at (u1, v) left and (u2, v) right ...
    H, W = left.shape
    P = lambda Kc, Rc, t, X: (lambda c: (Kc @ c)[:2] / c[2])(Rc.T @ (X - t))
    for X in ([12.0, -3.0, 0.0], [15.0, 2.0, 4.0], [9.0, -8.0, 1.5]):
        X = np.array(X)
        v1 = P(K, R.T, t1, X)[1]          # rows of R = rectified axes: R_G_C_rect maps world -> camera
        v2 = P(K, R.T, t2, X)[1]
        assert abs(v1 - v2) < 1e-9
    # the maps are the inverse rectifying homographies: the centre of the mask maps inside the image
    m = r["mask"]
    assert 0.3 < (m == 255).mean() < 1.0
    ys, xs = np.nonzero(m == 255)
    cy, cx = int(ys.mean()), int(xs.mean())
    assert 0 <= r["maps"][0, cy, cx] < W and 0 <= r["maps"][1, cy, cx] < H
    # remap: where the map lands exactly on pixel centres of a constant image, the value is kept
    rc, c = O.rectify_stereo_pair(K, R1, R2, t1, t2, np.full_like(left, 77), np.full_like(right, 200))
    inside = (c["maps"][0] > 1) & (c["maps"][0] < W - 2) & (c["maps"][1] > 1) & (c["maps"][1] < H - 2)
    assert (c["left"][inside] == 77).all()
    outside = (c["maps"][0] < -2) | (c["maps"][0] > W + 1) | (c["maps"][1] < -2) | (c["maps"][1] > H + 1)
    assert (c["left"][outside] == 0).all()        # BORDER_CONSTANT 0


def test_remap_is_bilinear_sampling_of_the_maps():
    """cv::remap(INTER_LINEAR, BORDER_CONSTANT) as restated (5 fractional bits, (sum + 2^14) >> 15):
    within one grey level of scipy.ndimage.map_coordinates(order=1) at the returned maps on a smooth
    image pair (OpenCV is not in the image: what an independent implementation can pin)."""
    from scipy import ndimage
    K, R1, R2, t1, t2, _, _ = rig(9)
    H, W = 120, 160
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    left = (128 + 70 * np.sin(0.05 * xx) * np.cos(0.04 * yy)).astype(np.uint8)
    right = (120 + 60 * np.cos(0.03 * xx + 0.5) * np.sin(0.06 * yy)).astype(np.uint8)
    rc, r = O.rectify_stereo_pair(K, R1, R2, t1, t2, left, right)
    assert rc == O.OK
    for img, out, mx, my in ((left, r["left"], r["maps"][0], r["maps"][1]),
                             (right, r["right"], r["maps"][2], r["maps"][3])):
        want = ndimage.map_coordinates(img.astype(np.float64), [my.astype(np.float64), mx.astype(np.float64)],
                                       order=1, mode="constant", cval=0.0)
        interior = (mx > 1) & (my > 1) & (mx < W - 2) & (my < H - 2)
        assert interior.mean() > 0.5
        assert np.abs(out.astype(np.float64) - want)[interior].max() <= 1.0
