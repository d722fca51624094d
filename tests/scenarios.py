"""Seeded small scenarios shared by the CPU and GPU test-suites."""
import numpy as np

import oracle_ffi as O
from aerial_mapper_amd import synth


def camera(width=192, height=108, f=140.0, distortion=O.DIST_NONE, dist=(0, 0, 0, 0)):
    c = O.Camera()
    c.fu = c.fv = f
    c.cu = (width - 1) / 2.0
    c.cv = (height - 1) / 2.0
    c.width, c.height = width, height
    c.distortion = distortion
    for k in range(4):
        c.dist[k] = float(dist[k])
    return c


class Scene(object):
    """A DSM + ortho scenario: grid, cloud, camera, poses, frames."""

    def __init__(self, length_x, length_y, res, n_points, seed, center=(0.0, 0.0),
                 num_frames=12, cam=None, altitude=700.0, colored=False, tilt_deg=5.0,
                 point_extent=None):
        self.grid = O.make_grid(length_x, length_y, res, center[0], center[1])
        half = max(length_x, length_y) / 2.0 + 4.0 if point_extent is None else point_extent
        self.points = synth.make_points(n_points, half, seed, center=center)
        self.cam = cam or camera()
        self.colored = colored
        self.poses = synth.make_lawnmower_poses(num_frames, 0.45 * min(length_x, length_y),
                                                altitude, seed + 1, tilt_deg=tilt_deg,
                                                center=center)
        ch = 3 if colored else 1
        fr = synth.make_frames(num_frames, self.cam.height, self.cam.width, ch, salt=seed % 7)
        self.frames = [np.ascontiguousarray(fr[k]) for k in range(num_frames)]
        self.T_C_B = synth.IDENTITY_POSE.copy()
        self.center = center


def assert_dsm_close(got, want, tol=1e-4, lsb_cells=None):
    """DSM parity: identical NaN pattern, heights within `tol` metres
    (north_star: 1e-4 m).  Returns the fraction of bit-identical cells.

    tol <= 1e-6 is the FP64 mode's bar ("the reference's floats"): the GPU sums the same
    doubles in another order (1e-16 relative), so about one cell in 1e8 -- one whose double
    sits on a float rounding boundary -- comes out one float spacing away (DESIGN.md 4.4:
    99 999 999 of 1e8; soak of round 3: 2 cells in 2e8).  `lsb_cells` (default 2 at that bar,
    0 otherwise) such cells are allowed, each within ONE spacing of the stored float."""
    assert got.shape == want.shape
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "NaN pattern differs in %d cells" % int((gn != wn).sum())
    ok = ~wn
    if lsb_cells is None:
        lsb_cells = 2 if tol <= 1e-6 else 0
    if ok.any():
        g64, w64 = got[ok].astype(np.float64), want[ok].astype(np.float64)
        err = np.abs(g64 - w64)
        over = err > tol
        if over.any() and lsb_cells:
            one = np.spacing(np.abs(want[ok][over]).astype(np.float32)).astype(np.float64)
            assert int(over.sum()) <= lsb_cells and (err[over] <= one).all(), \
                "%d cells beyond %g m, max |dh| = %g m" % (int(over.sum()), tol, err.max())
        else:
            assert not over.any(), "max |dh| = %g m" % err.max()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (gn & wn)
    return float(same.mean())


def assert_layers_equal(got, want, names):
    for n in names:
        a, b = got[n], want[n]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.all(), "layer %s differs in %d cells (first at %s)" % (
            n, int((~eq).sum()), np.argwhere(~eq)[0])
