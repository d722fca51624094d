"""The reference's OWN demo executables -- aerial_mapper_demos/src/dsm/main-dsm.cc:46-113 and
ortho/main-ortho-backward-grid.cc:66-160, compiled UNCHANGED (oracle/Makefile, target `demos`,
over oracle/demokit: gflags / ROS / loaders for an on-disk dataset this test writes) -- once
against the reference's own dsm.cc / ortho-backward-grid.cc / aerial-mapper-grid-map.cc
(oracle/_ref/demo_*_ref, CPU) and once against the drop-in classes of include/ +
libaerial_mapper_shim.so (oracle/_ref/demo_*_dropin, GPU).  Both end in
map.publishUntilShutdown(); the kit's publisher writes the six layers instead.  The drop-in
must leave the same map behind."""
import os
import subprocess

import numpy as np
import pytest

from aerial_mapper_amd import synth
import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
LAYERS = ["ortho", "elevation", "elevation_angle", "num_observations", "observation_index",
          "colored_ortho"]
CE, CN, DE, DN, RES = 12.0, -7.0, 100.0, 80.0, 0.5
W, H, F = 160, 120, 9

MORE = ("demo_incremental_ref", "demo_incremental_dropin", "demo_forward_ref", "demo_forward_dropin",
        "demo_frompcl_ref", "demo_frompcl_dropin")
needs_more_demos = pytest.mark.skipif(
    not all(os.path.exists(os.path.join(REFDIR, n)) for n in MORE),
    reason="oracle/_ref/demo_{incremental,forward,frompcl}_* not built (needs /root/reference at build time)")

needs_demos = pytest.mark.skipif(
    not all(os.path.exists(os.path.join(REFDIR, n)) for n in
            ("demo_dsm_ref", "demo_ortho_ref", "demo_dsm_dropin", "demo_ortho_dropin")),
    reason="oracle/_ref/demo_* not built (needs /root/reference at build time)")


def _write_dataset(d):
    rng = np.random.default_rng(11)
    n = 4 * 110 * 90
    x = CE + (rng.random(n) - 0.5) * 110.0
    y = CN + (rng.random(n) - 0.5) * 90.0
    z = 400.0 + 6.0 * np.sin(0.05 * x) * np.cos(0.04 * y) + 0.05 * rng.random(n)
    # dsm.cc:42-43 subtracts center_NORTHING from x and center_EASTING from y
    with open(os.path.join(d, "cloud.txt"), "w") as f:
        for k in range(n):
            f.write("%.17g %.17g %.17g %d\n" % (x[k] + CN, y[k] + CE, z[k], int(rng.integers(0, 256))))
        f.write("0 0 -150 7\n")     # dropped by the loader (z <= -100)
    s45 = np.sqrt(0.5)
    with open(os.path.join(d, "poses.txt"), "w") as f:
        for k in range(F):
            q = np.array([0.01 * (k - 4), s45, s45 + 0.005 * k, 0.004 * (4 - k)])
            q /= np.linalg.norm(q)
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" %
                    (CE - 40.0 + 10.0 * k, CN + ((k % 3) - 1) * 18.0, 470.0, q[0], q[1], q[2], q[3]))
    with open(os.path.join(d, "rig.txt"), "w") as f:
        f.write("120 120 %.17g %.17g %d %d 1 -0.05 0.01 0.0002 -0.0001  0.02 -0.01 0.03 1 0 0 0\n" %
                ((W - 1) / 2.0, (H - 1) / 2.0, W, H))
    frames = synth.make_frames(F, H, W, 1, salt=3)
    for k in range(F):
        with open(os.path.join(d, "img_%d.jpg" % k), "wb") as f:   # (a PGM: see oracle/demokit)
            f.write(b"P5 %d %d 255\n" % (W, H))
            f.write(np.ascontiguousarray(frames[k]).tobytes())


def _run(exe, flags, outdir, env_extra=None):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, AMHIP_DEMO_OUT=outdir)
    for k in ("AERIAL_MAPPER_HIP_DEVICES", "AMHIP_DSM_FAST", "AMHIP_DSM_EXACT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([os.path.join(REFDIR, exe)] + flags, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    if not os.path.exists(os.path.join(outdir, "shape.txt")):
        return None        # (a main without a map to publish: the forward mosaic)
    rows, cols = (int(v) for v in open(os.path.join(outdir, "shape.txt")).read().split())
    return {n: np.fromfile(os.path.join(outdir, n + ".f32"), np.float32).reshape(cols, rows)
            for n in LAYERS}


def _dsm_flags(d):
    return ["--data_directory=" + d + "/", "--filename_camera_rig=rig.txt", "--filename_poses=poses.txt",
            "--prefix_images=img_", "--filename_point_cloud=" + os.path.join(d, "cloud.txt"),
            "--center_easting=%r" % CE, "--center_northing=%r" % CN, "--delta_easting=%r" % DE,
            "--delta_northing=%r" % DN, "--resolution=%r" % RES]


def _ortho_flags(d):
    return ["--backward_grid_data_directory=" + d + "/", "--backward_grid_filename_camera_rig=rig.txt",
            "--backward_grid_filename_poses=poses.txt", "--backward_grid_prefix_images=img_",
            "--load_point_cloud_from_file=true", "--point_cloud_filename=" + os.path.join(d, "cloud.txt"),
            "--backward_grid_center_easting=%r" % CE, "--backward_grid_center_northing=%r" % CN,
            "--backward_grid_delta_easting=%r" % DE, "--backward_grid_delta_northing=%r" % DN,
            "--backward_grid_resolution=%r" % RES, "--backward_grid_show_orthomosaic_opencv=false",
            "--backward_grid_save_orthomosaic_jpg=false"]


def _same_bits(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))


def _compare(got, want, exact):
    ge, we = got["elevation"], want["elevation"]
    assert np.array_equal(np.isnan(ge), np.isnan(we))
    ok = ~np.isnan(we)
    assert ok.mean() > 0.9
    S.assert_dsm_close(ge, we, tol=1e-6 if exact else 1e-4)
    covered = ~np.isnan(want["observation_index"])
    assert covered.mean() > 0.3
    if _same_bits(ge, we).all():
        for n in LAYERS:
            assert _same_bits(got[n], want[n]).all(), n
    else:
        # single-precision DSM: a height may differ by one float spacing (3e-5 m here).  The view
        # angle is a continuous function of it -- it moves by ~1e-7 rad, i.e. at most a float
        # spacing of the stored angle -- and a near tie between two views can flip.
        both = ~np.isnan(got["elevation_angle"]) & ~np.isnan(want["elevation_angle"])
        d = np.abs(got["elevation_angle"][both].astype(np.float64) - want["elevation_angle"][both])
        assert d.max() <= 2.4e-7, d.max()
        for n in ("num_observations", "observation_index", "ortho", "colored_ortho"):
            assert (~_same_bits(got[n], want[n])).mean() < 1e-3, n


@needs_demos
def test_reference_demo_binaries_run_on_the_cpu(tmp_path):
    """(no GPU) the reference-side executables alone: main-dsm.cc's map == the elevation of
    main-ortho-backward-grid.cc's map, the mosaic covers the scene."""
    d = str(tmp_path)
    _write_dataset(d)
    a = _run("demo_dsm_ref", _dsm_flags(d), os.path.join(d, "out_dsm_ref"))
    b = _run("demo_ortho_ref", _ortho_flags(d), os.path.join(d, "out_ortho_ref"))
    assert _same_bits(a["elevation"], b["elevation"]).all()
    assert np.isnan(a["observation_index"]).all() and (~np.isnan(b["observation_index"])).mean() > 0.3
    assert (a["ortho"] == 255.0).all() and (b["ortho"] != 255.0).mean() > 0.3


@needs_demos
@pytest.mark.gpu
@pytest.mark.parametrize("env,exact", [({}, True), ({"AMHIP_DSM_FAST": "1"}, False),
                                       ({"AERIAL_MAPPER_HIP_DEVICES": "0,0,0"}, True)])
def test_unchanged_demo_mains_leave_the_same_map_on_the_drop_in(tmp_path, env, exact):
    d = str(tmp_path)
    _write_dataset(d)
    want_dsm = _run("demo_dsm_ref", _dsm_flags(d), os.path.join(d, "ref_dsm"))
    got_dsm = _run("demo_dsm_dropin", _dsm_flags(d), os.path.join(d, "gpu_dsm"), env)
    ge, we = got_dsm["elevation"], want_dsm["elevation"]
    assert np.array_equal(np.isnan(ge), np.isnan(we))
    S.assert_dsm_close(ge, we, tol=1e-6 if exact else 1e-4)
    want = _run("demo_ortho_ref", _ortho_flags(d), os.path.join(d, "ref_ortho"))
    got = _run("demo_ortho_dropin", _ortho_flags(d), os.path.join(d, "gpu_ortho"), env)
    _compare(got, want, exact)


# ---- the other three mains of aerial_mapper_demos/src/ortho ---------------

def _incremental_flags(d):
    return [f for f in _ortho_flags(d) if not f.startswith(("--load_point_cloud_from_file",
                                                            "--point_cloud_filename"))] + \
        ["--dense_pcl_use_every_nth_image=1"]


def _write_stereo_clouds(d):
    """One cloud per stereo pair k = 1 .. F-1 (what stereo::Stereo::addFrame would hand back,
    main-ortho-backward-grid-incremental.cc:143-146): the points of cloud.txt within 32 m of the
    k-th camera, in the file's coordinates (dsm.cc:42-43's offsets already folded in)."""
    pts = np.loadtxt(os.path.join(d, "cloud.txt"))[:-1, :3]
    poses = np.loadtxt(os.path.join(d, "poses.txt"))
    for k in range(1, F):
        # undo the fold for the distance test: file x = map x + CN, file y = map y + CE
        dx, dy = pts[:, 0] - CN - poses[k, 0], pts[:, 1] - CE - poses[k, 1]
        sel = pts[dx * dx + dy * dy < 32.0 ** 2]
        if k == 5:
            sel = sel[:0]           # an empty pair: "Passed empty point cloud to DSM module"
        with open(os.path.join(d, "stereo_%d.txt" % k), "w") as f:
            for p in sel:
                f.write("%.17g %.17g %.17g\n" % tuple(p))


@needs_more_demos
@pytest.mark.gpu
@pytest.mark.parametrize("env,exact", [({}, True), ({"AMHIP_DSM_FAST": "1"}, False),
                                       ({"AERIAL_MAPPER_HIP_DEVICES": "0,0"}, True)])
def test_unchanged_incremental_main_leaves_the_same_map_on_the_drop_in(tmp_path, env, exact):
    """main-ortho-backward-grid-incremental.cc:64-169 -- BASELINE configs[4]'s call sequence:
    per stereo pair Dsm::process of that pair's cloud onto the persistent elevation layer, then
    OrthoBackwardGrid::process of the images since the last pair onto the persistent angle /
    index / ortho layers, then publishOnce.  The session's residency logic (what is still on the
    device, what the host changed) is what this exercises."""
    d = str(tmp_path)
    _write_dataset(d)
    _write_stereo_clouds(d)
    e = dict(env, AMHIP_DEMO_STEREO_PREFIX=os.path.join(d, "stereo_"), AMHIP_DEMO_KEEP_RUNNING="1")
    want = _run("demo_incremental_ref", _incremental_flags(d), os.path.join(d, "ref_inc"), e)
    got = _run("demo_incremental_dropin", _incremental_flags(d), os.path.join(d, "gpu_inc"), e)
    assert (~np.isnan(want["elevation"])).mean() > 0.5 and np.isnan(want["elevation"]).any()
    _compare_partial(got, want, exact)


def _compare_partial(got, want, exact):
    """_compare for a map the clouds cover only partly."""
    ge, we = got["elevation"], want["elevation"]
    assert np.array_equal(np.isnan(ge), np.isnan(we))
    S.assert_dsm_close(ge, we, tol=1e-6 if exact else 1e-4)
    assert (~np.isnan(want["observation_index"])).mean() > 0.2
    if _same_bits(ge, we).all():
        for n in LAYERS:
            assert _same_bits(got[n], want[n]).all(), n
    else:
        assert not exact or _same_bits(ge, we).mean() > 0.999
        for n in ("num_observations", "observation_index", "ortho", "colored_ortho"):
            assert (~_same_bits(got[n], want[n])).mean() < 1e-3, n


def _forward_flags(d, batch):
    return ["--forward_homography_data_directory=" + d + "/", "--forward_homography_filename_camera_rig=rig.txt",
            "--forward_homography_filename_poses=poses.txt", "--forward_homography_prefix_images=img_",
            "--forward_homography_origin_easting_m=%r" % CE, "--forward_homography_origin_northing_m=%r" % CN,
            "--forward_homography_origin_elevation_m=0", "--forward_homography_ground_plane_elevation_m=400",
            "--forward_homography_width_mosaic_pixels=240", "--forward_homography_height_mosaic_pixels=200",
            "--forward_homography_batch=%s" % ("true" if batch else "false")]


@needs_more_demos
@pytest.mark.gpu
@pytest.mark.parametrize("batch", [True, False])
def test_unchanged_forward_homography_main_writes_the_same_mosaic(tmp_path, batch):
    """main-ortho-forward-homography.cc:80-102: OrthoForwardHomography::batch /
    ::updateOrthomosaic per image.  Reference side: the reference's class over refkit's restated
    OpenCV (what it hands to cv::imwrite); drop-in: the GPU mosaic, written where the reference
    writes (Settings::filename_mosaic_output, here its default) as a PPM.  (The OpenCV pieces
    themselves stay "parity unpinned".)"""
    d = str(tmp_path)
    _write_dataset(d)
    out = "/tmp/result.jpg.ppm"       # ortho::Settings::filename_mosaic_output's default + ".ppm"
    if os.path.exists(out):
        os.remove(out)
    assert _run("demo_forward_ref", _forward_flags(d, batch), os.path.join(d, "ref_fwd")) is None
    rows, cols, ch = (int(v) for v in open(os.path.join(d, "ref_fwd", "mosaic_shape.txt")).read().split())
    want = np.fromfile(os.path.join(d, "ref_fwd", "mosaic.i16"), np.int16).reshape(rows, cols, ch)
    assert (rows, cols, ch) == (200, 240, 3) and (want != 0).mean() > 0.2
    _run("demo_forward_dropin", _forward_flags(d, batch), os.path.join(d, "gpu_fwd"))
    raw = open(out, "rb").read()
    os.remove(out)
    head = b"P6\n240 200\n255\n"
    assert raw.startswith(head)
    got = np.frombuffer(raw, np.uint8, rows * cols * 3, len(head)).reshape(rows, cols, 3)
    assert np.array_equal(got, np.clip(want, 0, 255).astype(np.uint8))


def _frompcl_flags(d, adaptive):
    return ["--data_directory=" + d + "/", "--filename_camera_rig=rig.txt", "--filename_poses=poses.txt",
            "--prefix_images=img_", "--load_point_cloud_from_file=true",
            "--filename_point_cloud=" + os.path.join(d, "cloud.txt"),
            "--ortho_from_pcl_center_easting=%r" % CE, "--ortho_from_pcl_center_northing=%r" % CN,
            "--ortho_from_pcl_delta_easting=%r" % DE, "--ortho_from_pcl_delta_northing=%r" % DN,
            "--ortho_from_pcl_resolution=%r" % RES, "--ortho_from_pcl_interpolation_radius=2",
            "--ortho_from_pcl_show_orthomosaic_opencv=false",
            "--ortho_from_pcl_use_adaptive_interpolation=%s" % ("true" if adaptive else "false")]


@needs_more_demos
@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_unchanged_ortho_from_pcl_main_leaves_the_same_map_on_the_drop_in(tmp_path, adaptive):
    """main-ortho-from-pcl.cc:59-148 with --load_point_cloud_from_file (no centre offsets here:
    ortho-from-pcl.cc:30-31, so the cloud covers the map only partly -- the rest stays 255, or is
    reached by the x10 / x100 retries when adaptive)."""
    d = str(tmp_path)
    _write_dataset(d)
    want = _run("demo_frompcl_ref", _frompcl_flags(d, adaptive), os.path.join(d, "ref_pcl"))
    got = _run("demo_frompcl_dropin", _frompcl_flags(d, adaptive), os.path.join(d, "gpu_pcl"))
    a, b = got["ortho"], want["ortho"]
    assert ((a == 255.0) == (b == 255.0)).all()
    assert (b != 255.0).mean() > (0.95 if adaptive else 0.5)
    assert np.abs(a.astype(np.float64) - b).max() <= 1e-3        # intensities 0 .. 255
    for n in LAYERS:
        if n != "ortho":                                          # untouched by this path
            assert _same_bits(got[n], want[n]).all(), n


@needs_more_demos
def test_more_reference_demo_binaries_run_on_the_cpu(tmp_path):
    """(no GPU) the reference-side executables of the three mains alone."""
    d = str(tmp_path)
    _write_dataset(d)
    _write_stereo_clouds(d)
    e = dict(AMHIP_DEMO_STEREO_PREFIX=os.path.join(d, "stereo_"), AMHIP_DEMO_KEEP_RUNNING="1")
    inc = _run("demo_incremental_ref", _incremental_flags(d), os.path.join(d, "ref_inc"), e)
    full = _run("demo_ortho_ref", _ortho_flags(d), os.path.join(d, "ref_ortho"))
    # the incremental map is the batch map wherever a stereo cloud reached, NaN elsewhere
    cov = ~np.isnan(inc["elevation"])
    assert 0.5 < cov.mean() < 1.0 and (~np.isnan(inc["observation_index"])).mean() > 0.2
    # (cells at the rim of a patch see only part of their neighbourhood: close, not equal)
    dz = np.abs(inc["elevation"][cov] - full["elevation"][cov])
    assert dz.max() < 2.0 and np.median(dz) < 1e-3
    assert _run("demo_forward_ref", _forward_flags(d, True), os.path.join(d, "ref_fwd")) is None
    assert os.path.getsize(os.path.join(d, "ref_fwd", "mosaic.i16")) == 200 * 240 * 3 * 2
    pcl = _run("demo_frompcl_ref", _frompcl_flags(d, False), os.path.join(d, "ref_pcl"))
    assert 0.5 < (pcl["ortho"] != 255.0).mean() < 1.0
