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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
LAYERS = ["ortho", "elevation", "elevation_angle", "num_observations", "observation_index",
          "colored_ortho"]
CE, CN, DE, DN, RES = 12.0, -7.0, 100.0, 80.0, 0.5
W, H, F = 160, 120, 9

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
    env.pop("AERIAL_MAPPER_HIP_DEVICES", None)
    env.update(env_extra or {})
    r = subprocess.run([os.path.join(REFDIR, exe)] + flags, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
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
    err = float(np.abs(ge[ok].astype(np.float64) - we[ok]).max())
    assert err <= (1e-6 if exact else 1e-4), err
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
@pytest.mark.parametrize("env,exact", [({"AMHIP_DSM_EXACT": "1"}, True), ({}, False),
                                       ({"AMHIP_DSM_EXACT": "1", "AERIAL_MAPPER_HIP_DEVICES": "0,0,0"}, True)])
def test_unchanged_demo_mains_leave_the_same_map_on_the_drop_in(tmp_path, env, exact):
    d = str(tmp_path)
    _write_dataset(d)
    want_dsm = _run("demo_dsm_ref", _dsm_flags(d), os.path.join(d, "ref_dsm"))
    got_dsm = _run("demo_dsm_dropin", _dsm_flags(d), os.path.join(d, "gpu_dsm"), env)
    ge, we = got_dsm["elevation"], want_dsm["elevation"]
    assert np.array_equal(np.isnan(ge), np.isnan(we))
    ok = ~np.isnan(we)
    assert float(np.abs(ge[ok].astype(np.float64) - we[ok]).max()) <= (1e-6 if exact else 1e-4)
    want = _run("demo_ortho_ref", _ortho_flags(d), os.path.join(d, "ref_ortho"))
    got = _run("demo_ortho_dropin", _ortho_flags(d), os.path.join(d, "gpu_ortho"), env)
    _compare(got, want, exact)
