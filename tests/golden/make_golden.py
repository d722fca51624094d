#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- small input/output vectors of the hot path.

The reference has no tests or fixtures of its own and its translation units
cannot be built in this image, so these vectors are the ORACLE's: produced by the restated loops
over the reference's vendored nanoflann (oracle/_ref/liboracle_ref.so) -- or, identically bit for
bit (tests/test_reference_loops.py), by the consistency-check libraries of oracle/refkit/ where they
are built.  They pin the GPU path and the oracle against each other across rounds and on the GPU
box (where /root/reference does not exist); they are NOT reference outputs: parity stays
"unpinned" in the task's sense.

    python tests/golden/make_golden.py        (needs /root/reference)

Inputs are stored verbatim (not as seeds) so the fixtures do not depend on
numpy's generators staying bit-stable.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_ffi as O  # noqa: E402
import scenarios as S  # noqa: E402
from aerial_mapper_amd import synth  # noqa: E402

# the checker that writes the fixtures: the reference's OWN loops (dsm.cc, ortho-backward-grid.cc,
# ortho-from-pcl.cc, densifier.cpp, ortho-forward-homography.cc compiled unchanged against
# oracle/refkit/) where they are built, else the vendored-nanoflann build of the restated loops
WHICH = "loops" if O.have_loops() else "ref"
GRID = "ref" if O.have_ref() else "port"     # (geometry helpers live in the oracle libraries)


def grid_tuple(g):
    return np.array([g.length_x, g.length_y, g.resolution, g.pos_x, g.pos_y], np.float64)


def cam_tuple(c):
    return np.array([c.fu, c.fv, c.cu, c.cv, c.width, c.height, c.distortion] + list(c.dist),
                    np.float64)


def dsm_case(name, length_x, length_y, res, n, seed, radius=1, ce=0.0, cn=0.0,
             center=(0.0, 0.0), keep=None, extent=None, init=None):
    g = O.make_grid(length_x, length_y, res, center[0], center[1], which=GRID)
    half = (max(length_x, length_y) / 2.0 + 4.0) if extent is None else extent
    pts = synth.make_points(n, half, seed, center=center)
    if keep is not None:
        pts = np.ascontiguousarray(pts[keep(pts)])
    elev0 = None
    if init is not None:
        elev0 = init(g)
    rc, elev, _ = O.dsm_process(pts, g, radius, ce, cn,
                                elevation=None if elev0 is None else elev0.copy(), which=WHICH)
    assert rc == O.OK, rc
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="dsm", grid=grid_tuple(g),
                        points=pts, radius_sq=radius, center_easting=ce, center_northing=cn,
                        elevation_init=np.zeros(0, np.float32) if elev0 is None else elev0,
                        elevation=elev)
    print("%-28s %4dx%-4d pts=%6d  NaN cells=%d" % (name, g.rows, g.cols, pts.shape[0],
                                                   int(np.isnan(elev).sum())))


def ortho_case(name, length_x, length_y, res, n, seed, num_frames, altitude, colored=False,
               cam=None, batches=None, keep=None, nobs_init=0.0):
    sc = S.Scene(length_x, length_y, res, n, seed, num_frames=num_frames, altitude=altitude,
                 colored=colored, cam=cam or S.camera(96, 54, 70.0))
    if keep is not None:
        sc.points = np.ascontiguousarray(sc.points[keep(sc.points)])
    rc, elev, _ = O.dsm_process(sc.points, sc.grid, which=WHICH)
    assert rc == O.OK
    layers = O.new_layers(sc.grid)
    layers["elevation"] = elev
    layers["num_observations"][:] = nobs_init
    batches = batches or [(0, num_frames)]
    for lo, hi in batches:
        rc = O.ortho_process(sc.grid, sc.cam, sc.poses[lo:hi], sc.T_C_B, sc.frames[lo:hi], layers,
                             colored=colored, which=WHICH)
        assert rc == O.OK
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), kind="ortho", grid=grid_tuple(sc.grid),
        camera=cam_tuple(sc.cam), T_G_B=sc.poses, T_C_B=sc.T_C_B, frames=np.stack(sc.frames),
        colored=colored, batches=np.array(batches, np.int64), num_observations_init=nobs_init,
        elevation=elev, elevation_angle=layers["elevation_angle"],
        observation_index=layers["observation_index"],
        num_observations=layers["num_observations"], ortho=layers["ortho"],
        colored_ortho=layers["colored_ortho"])
    cov = float((~np.isnan(layers["observation_index"])).mean())
    print("%-28s %4dx%-4d F=%d coverage=%.3f" % (name, sc.grid.rows, sc.grid.cols, num_frames, cov))


def pcl_case(name, length_x, length_y, res, n, seed, radius, adaptive, extent=None,
             center=(0.0, 0.0), exact_at=None):
    g = O.make_grid(length_x, length_y, res, center[0], center[1], which=GRID)
    half = (max(length_x, length_y) / 2.0 + 4.0) if extent is None else extent
    pts = synth.make_points(n, half, seed, center=center)
    inten = ((np.arange(n) * 37 + seed) % 256).astype(np.int32)
    if exact_at is not None:
        x, y = O.cell_position(g, exact_at[0], exact_at[1], which=GRID)
        pts[3, :2] = (x, y)
        inten[3] = 249
    rc, ortho = O.ortho_from_pcl(pts, inten, g, radius, adaptive, which=WHICH)
    assert rc == O.OK
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="pcl", grid=grid_tuple(g),
                        points=pts, intensities=inten, radius_sq=radius, adaptive=adaptive,
                        ortho=ortho)
    print("%-28s %4dx%-4d pts=%6d untouched=%d" % (name, g.rows, g.cols, n,
                                                 int((ortho == 255.0).sum())))


def fwd_case(name, cam, mosaic_wh, ground, origin, num_frames, half_extent, altitude, seed,
             colored=False, incremental=False, tilt=4.0):
    """ortho::OrthoForwardHomography (oracle/amo_forward.cc; OpenCV/aslam
    semantics restated -- parity unpinned): batch() or a sequence of
    updateOrthomosaic() calls."""
    desc = O.mosaic_desc(mosaic_wh[0], mosaic_wh[1], ground, origin)
    poses = synth.make_lawnmower_poses(num_frames, half_extent, altitude, seed, tilt_deg=tilt,
                                       center=(origin[0], origin[1]))
    frames = synth.make_frames(num_frames, cam.height, cam.width, 3 if colored else 1, salt=seed)
    # smooth the hash frames a little so that zero pixels (mask holes) exist but are rare
    frames = np.ascontiguousarray(np.where(frames < 6, 0, frames).astype(np.uint8))
    T_C_B = np.array([0.02, -0.01, 0.03, 1.0, 0.0, 0.0, 0.0])
    fm = O.ForwardMosaic(cam, desc, T_C_B, which=GRID)
    # the reference's own flow keeps its mask private: the mosaic comes from it, the mask from
    # the restated flow (the mosaics of the two are required to agree)
    rf = O.ReferenceForwardMosaic(cam, desc, T_C_B) if WHICH == "loops" else None
    steps = []
    if incremental:
        for k in range(num_frames):
            assert fm.update(poses[k], frames[k]) == O.OK
            if rf is not None:
                assert rf.update(poses[k], frames[k]) == O.OK
                assert np.array_equal(rf.result, fm.result)
            steps.append(fm.result.copy())
    else:
        assert fm.batch(poses, [f for f in frames]) == O.OK
        if rf is not None:
            assert rf.batch(poses, [f for f in frames]) == O.OK
            assert np.array_equal(rf.result, fm.result)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), kind="fwd", camera=cam_tuple(cam),
        mosaic=np.array([mosaic_wh[0], mosaic_wh[1], ground] + list(origin), np.float64),
        T_G_B=poses, T_C_B=T_C_B, frames=frames, colored=colored, incremental=incremental,
        result=fm.result, mask=fm.mask,
        step_checksums=np.array([int(s.astype(np.int64).sum()) for s in steps], np.int64))
    print("%-28s %4dx%-4d F=%d covered=%.3f" % (name, mosaic_wh[0], mosaic_wh[1], num_frames,
                                                float((fm.mask > 0).mean())))


def densify_case(name, h, w, seed):
    """stereo::Densifier::computePointCloud: disparity map -> world points, raster order."""
    rng = np.random.default_rng(seed)
    disp = rng.uniform(0.0, 80.0, (h, w)).astype(np.float32)
    disp[rng.random((h, w)) < 0.2] = rng.choice(np.array([0.0, 1.0, -1.0, 0.5], np.float32))
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    K = np.array([[260.0, 0, (w - 1) / 2.0], [0, 265.5, (h - 1) / 2.0], [0, 0, 1]])
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    qw, qx, qy, qz = q
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    t = np.array([12.5, -40.0, 430.0])
    pts, inten = O.densify(disp, img, K, 0.83, R, t, which=WHICH if WHICH == "loops" else GRID)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="densify", disparity=disp,
                        image_left=img, K=K, baseline=0.83, R_G_C=R, t_G_C1=t, points=pts,
                        intensities=inten)
    print("%-28s %4dx%-4d points=%d" % (name, w, h, pts.shape[0]))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "fwd":
        return main_fwd()
    if not O.have_ref():
        raise SystemExit("oracle/_ref/liboracle_ref.so missing: run `make -C oracle` where "
                         "/root/reference exists")
    assert O.lib("ref").amo_uses_vendored_nanoflann() == 1
    dsm_case("dsm_sparse_1m", 70.0, 50.0, 1.0, 3800, 101)
    dsm_case("dsm_dense_quarter", 16.0, 12.0, 0.25, 2600, 102, extent=10.0)
    dsm_case("dsm_holes_ladder", 60.0, 40.0, 0.5, 700, 103)
    dsm_case("dsm_left_half_only", 48.0, 36.0, 0.5, 5000, 104, keep=lambda p: p[:, 0] < -6.0)
    dsm_case("dsm_radius3_offsets", 40.0, 30.0, 0.5, 4000, 105, radius=3, ce=5.5, cn=-2.25,
             center=(5.5, -2.25), extent=32.0)
    dsm_case("dsm_incremental_overwrite", 40.0, 30.0, 0.5, 900, 106,
             keep=lambda p: p[:, 1] > 2.0,
             init=lambda g: np.full((g.cols, g.rows), 123.5, np.float32))
    cam_small = S.camera(96, 54, 70.0)
    ortho_case("ortho_gray", 70.0, 50.0, 1.0, 9000, 201, 7, 470.0)
    ortho_case("ortho_colored", 60.0, 44.0, 1.0, 7000, 202, 6, 470.0, colored=True)
    ortho_case("ortho_incremental_3batches", 70.0, 50.0, 1.0, 9000, 203, 9, 470.0,
               batches=[(0, 3), (3, 7), (7, 9)], nobs_init=0.75)
    ortho_case("ortho_nan_elevation", 60.0, 44.0, 1.0, 2500, 204, 5, 480.0,
               keep=lambda p: p[:, 0] < 4.0)
    ortho_case("ortho_radtan", 50.0, 40.0, 1.0, 5000, 205, 5, 470.0,
               cam=S.camera(96, 54, 70.0, O.DIST_RADTAN, (-0.28, 0.07, 2e-4, -1e-4)))
    del cam_small
    pcl_case("pcl_radius2", 50.0, 36.0, 1.0, 2200, 301, 2, False, exact_at=(9, 4))
    pcl_case("pcl_dense_half_metre", 30.0, 22.0, 0.5, 5000, 302, 1, False, center=(2.0, 1.0))
    pcl_case("pcl_adaptive_corner", 40.0, 30.0, 1.0, 250, 303, 2, True, extent=8.0,
             center=(-10.0, -6.0))
    densify_case("densify_small", 60, 88, 501)
    main_fwd()


def main_fwd():
    fwd_case("fwd_batch_gray", S.camera(96, 54, 70.0), (160, 120), 400.0, (0.0, 0.0, 0.0), 8,
             40.0, 470.0, 401)
    fwd_case("fwd_batch_colored", S.camera(80, 60, 64.0), (128, 144), 400.0, (3.0, -2.0, 0.0), 6,
             36.0, 465.0, 402, colored=True)
    fwd_case("fwd_incremental_gray", S.camera(96, 54, 70.0), (150, 110), 400.0, (0.0, 0.0, 0.0),
             5, 30.0, 470.0, 403, incremental=True)
    fwd_case("fwd_batch_radtan", S.camera(96, 54, 70.0, O.DIST_RADTAN, (-0.28, 0.07, 2e-4, -1e-4)),
             (160, 120), 400.0, (0.0, 0.0, 0.0), 6, 36.0, 470.0, 404)


if __name__ == "__main__":
    main()
