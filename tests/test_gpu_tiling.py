"""GPU: windows of one map (the multi-GPU tiling) reproduce the full-map result,
and the HIP halo-selection kernel agrees with the host-side masks."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fp64_gather(monkeypatch):
    """These tests compare windows with the full map to 1e-6 m: that is a statement about the
    tiling (same neighbour sets), so the contexts use the FP64 gather, whose result does not move
    by a float spacing with the order of the points.  The single-precision default is covered at
    the contract's 1e-4 m by tests/test_gpu_bench_multirank.py and tests/test_gpu_dsm_fast.py."""
    monkeypatch.setenv("AMHIP_DSM_EXACT", "1")

LAYERS = ["elevation_angle", "observation_index", "ortho"]


def test_windows_reproduce_full_map():
    import aerial_mapper_amd as A
    from aerial_mapper_amd import tiling
    sc = S.Scene(200.0, 160.0, 0.5, 90000, seed=70, num_frames=10, altitude=480.0)
    sc.points = np.ascontiguousarray(sc.points[(sc.points[:, 0] < 60.0)])  # leave a hole
    g = sc.grid
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    nc = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
    with A.AerialGridMap(st) as m:
        A.Dsm(A.DsmSettings(), m).process(sc.points, m)
        A.OrthoBackwardGrid(nc, A.OrthoSettings(), m).process(sc.poses, sc.frames, m)
        full = {n: m.get(n) for n in ["elevation"] + LAYERS}
    rc, oracle_elev, _ = O.dsm_process(sc.points, g)
    S.assert_dsm_close(full["elevation"], oracle_elev)

    layout = tiling.TileLayout(g.rows, g.cols, 2, 2, align_i=64, align_j=32)
    margin = tiling.halo_margin(1, g.resolution)
    cx, cy = tiling.cell_coords(sc.points, g)
    for rank in range(layout.world):
        win = layout.window(rank)
        i0, j0, r, c = win
        sub = np.ascontiguousarray(sc.points[tiling.in_window(cx, cy, win, margin / g.resolution)])
        with A.AerialGridMap(st, window=win) as m:
            assert (m.rows, m.cols) == (r, c)
            A.Dsm(A.DsmSettings(), m).process(sub, m)
            A.OrthoBackwardGrid(nc, A.OrthoSettings(), m).process(sc.poses, sc.frames, m)
            part = {n: m.get(n) for n in ["elevation"] + LAYERS}
        want_e = full["elevation"][j0:j0 + c, i0:i0 + r]
        # identical neighbour sets; bins are anchored differently, so only the
        # order of the double sums may move
        S.assert_dsm_close(part["elevation"], want_e, tol=1e-6)
        same = part["elevation"].view(np.uint32) == want_e.view(np.uint32)
        assert same.mean() > 0.999
        for n in LAYERS:
            a, b = part[n][same], full[n][j0:j0 + c, i0:i0 + r][same]
            eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert eq.all(), (rank, n, int((~eq).sum()))


def test_halo_select_kernel_matches_masks():
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import tiling
    sc = S.Scene(160.0, 128.0, 0.5, 70000, seed=71)
    g = sc.grid
    ce, cn = 3.5, -1.25
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    layout = tiling.TileLayout(g.rows, g.cols, 2, 2)
    margin = tiling.halo_margin(1, g.resolution)
    cx, cy = tiling.cell_coords(sc.points, g, ce, cn)
    wins = layout.windows()
    with A.AerialGridMap(st, window=wins[0]) as m:
        dev = torch.from_numpy(sc.points).cuda()
        got = tiling.select_for_windows(dev, g, wins, margin, ce, cn, map_=m)
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    for w, t in zip(wins, got):
        want = sc.points[tiling.in_window(cx, cy, w, margin / g.resolution)]
        assert t.shape[0] == want.shape[0] and want.shape[0] > 0
        assert np.array_equal(key(t.cpu().numpy()), key(want))


class _ThreadComm(object):
    """Stand-in for torch.distributed inside ONE process: every 'rank' is a
    thread with its own window context on the same GPU.  Lets the CUDA side of
    route_points() (HIP halo selection, buffers, workspace reuse) run on a
    1-GPU box; the RCCL calls themselves are covered by the driver's multi-GPU
    run and, on CPU tensors, by tests/test_tiling_gloo.py."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.counts = {}
        self.rows = {}

    def bind(self, rank):
        parent = self

        class _C(object):
            def exchange_counts(self, sc):
                import torch
                parent.counts[rank] = sc.cpu()
                parent.barrier.wait()
                out = torch.stack([parent.counts[r][rank] for r in range(parent.world)]).to(sc.device)
                parent.barrier.wait()
                return out

            def exchange_rows(self, out_rows, in_rows, recv_counts, send_counts):
                import torch
                torch.cuda.synchronize()
                parent.rows[rank] = (in_rows, send_counts)
                parent.barrier.wait()
                pos = 0
                for r in range(parent.world):
                    src, sc = parent.rows[r]
                    off = sum(sc[:rank])
                    n = sc[rank]
                    assert n == recv_counts[r]
                    out_rows[pos:pos + n] = src[off:off + n]
                    pos += n
                torch.cuda.synchronize()
                parent.barrier.wait()
            def exchange_equal(self, out_rows, in_rows):
                import torch
                torch.cuda.synchronize()
                parent.rows[rank] = in_rows
                parent.barrier.wait()
                cap = in_rows.shape[0] // parent.world
                for r in range(parent.world):
                    out_rows[r * cap:(r + 1) * cap] = parent.rows[r][rank * cap:(rank + 1) * cap]
                torch.cuda.synchronize()
                parent.barrier.wait()
        return _C()


def test_route_points_cuda_path_two_ranks_one_gpu():
    import threading
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import tiling
    sc = S.Scene(256.0, 96.0, 0.5, 80000, seed=72, point_extent=140.0)
    g = sc.grid
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    layout = tiling.TileLayout(g.rows, g.cols, 2, 1)
    cx, cy = tiling.cell_coords(sc.points, g)
    rc, full, _ = O.dsm_process(sc.points[tiling.in_window(cx, cy, (0, 0, g.rows, g.cols), 0)], g)
    comm = _ThreadComm(2)
    out, errs = {}, []

    def run(rank):
        try:
            win = layout.window(rank)
            own = sc.points[tiling.owner_mask(cx, cy, win)]
            n = own.shape[0]
            buf = torch.empty((n + 20000, 3), dtype=torch.float64, device="cuda")
            buf[:n] = torch.from_numpy(np.ascontiguousarray(own)).cuda()
            with A.AerialGridMap(st, window=win) as m:
                cloud = tiling.route_points(buf[:n], g, layout, rank, radius_sq=1, map_=m,
                                            assume_owned=True, cap=8000, workspace=buf,
                                            comm=comm.bind(rank))
                assert cloud.data_ptr() == buf.data_ptr() and cloud.shape[0] > n
                A.Dsm(A.DsmSettings(), m).process(cloud, m)
                out[rank] = (win, m.get("elevation"))
        except Exception as e:  # surface in the main thread
            errs.append(e)
            comm.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    for rank in range(2):
        (i0, j0, r, c), elev = out[rank]
        S.assert_dsm_close(elev, full[j0:j0 + c, i0:i0 + r], tol=1e-6)


@pytest.mark.parametrize("npts,lx,ly,tiles", [(80000, 256.0, 96.0, (2, 1)), (90000, 200.0, 160.0, (2, 2)),
                                                (2600000, 800.0, 500.0, (2, 1)),
                                                (150000, 768.0, 64.0, (8, 1))])   # a node's 8 GPUs
def test_tiled_dsm_selects_the_halo_in_its_binning_pass(npts, lx, ly, tiles):
    """tiling.TiledDsm (amhip_dsm_tiled_begin_dev / _finish_dev): small clouds take a selection
    pass of their own (one-level sort), the large one the three-pass sort whose count kernel
    selects on the way; NaN padding rows are dropped; every window equals the full-map DSM."""
    import threading
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth, tiling
    res = 0.5
    g = O.make_grid(lx, ly, res, 12.5, -40.0)
    rng = np.random.default_rng(73)
    pts = np.empty((npts, 3))
    pts[:, 0] = rng.uniform(g.pos_x - lx / 2 - 3.0, g.pos_x + lx / 2 + 3.0, npts)
    pts[:, 1] = rng.uniform(g.pos_y - ly / 2 - 3.0, g.pos_y + ly / 2 + 3.0, npts)
    pts[:, 2] = synth.terrain_height(pts[:, 0], pts[:, 1]) + rng.uniform(-0.5, 0.5, npts)
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    layout = tiling.TileLayout(g.rows, g.cols, tiles[0], tiles[1])
    world = layout.world
    cx, cy = tiling.cell_coords(pts, g)
    # The cloud overhangs the map by 3 m on every side.  The reference's kd-tree -- and the
    # single-GPU path -- use those points for the border cells; in a tiled run the border
    # windows own them (owner_mask(..., layout)): the full-map DSM below sees the WHOLE cloud.
    inside_any = np.zeros(npts, bool)
    for r in range(world):
        mk = tiling.owner_mask(cx, cy, layout.window(r), layout)
        assert not (inside_any & mk).any()          # exactly one owner each
        inside_any |= mk
    assert inside_any.all() and (cx < -0.5).any() and (cy > g.cols - 0.5).any()
    with A.AerialGridMap(st) as m:
        A.Dsm(A.DsmSettings(), m).process(np.ascontiguousarray(pts), m)
        full = m.get("elevation")
    cap = tiling.halo_strip_rows(npts / ((lx + 6) * (ly + 6)), max(lx, ly), 1, res, slack=2.0)
    comm = _ThreadComm(world)
    out, errs = {}, []

    def run(rank):
        try:
            win = layout.window(rank)
            own = pts[tiling.owner_mask(cx, cy, win, layout)]
            n = own.shape[0]
            buf = torch.full((n + world * cap, 3), 7.0, dtype=torch.float64, device="cuda")
            buf[:n] = torch.from_numpy(np.ascontiguousarray(own)).cuda()
            with A.AerialGridMap(st, window=win) as m:
                m.enable_timing(True)
                t = tiling.TiledDsm(A.DsmSettings(), m, layout, rank, cap, comm=comm.bind(rank))
                for _ in range(2):                       # a second step re-uses every buffer
                    m.reset()
                    t.process(buf, n)
                got = buf[n:n + t.recv_rows]       # (rows of the geometric neighbours only)
                got = got[~torch.isnan(got[:, 0])].cpu().numpy()
                out[rank] = (win, m.get("elevation"), got, int(t.counts.sum().item()),
                             m.kernel_times()["k_halo_select"][1])
        except Exception as e:  # surface in the main thread
            errs.append(e)
            comm.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    assert not errs, errs
    margin = tiling.halo_margin(1, res)
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    sent_total = recv_total = 0
    for rank in range(world):
        (i0, j0, r, c), elev, got, sent, select_launches = out[rank]
        win = layout.window(rank)
        want = pts[tiling.in_window(cx, cy, win, margin / res) &
                   ~tiling.owner_mask(cx, cy, win, layout)]
        assert want.shape[0] > 0
        assert got.shape == want.shape and np.array_equal(key(got), key(want))
        sent_total += sent
        recv_total += got.shape[0]
        # the big cloud: selected inside k_dsm_p3_count, no pass of its own
        assert select_launches == (0 if npts > 1000000 else 2)
        S.assert_dsm_close(elev, full[j0:j0 + c, i0:i0 + r], tol=1e-6)
    assert sent_total == recv_total


def test_tiled_dsm_reports_halo_overflow():
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import tiling
    g = O.make_grid(128.0, 64.0, 0.5)
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    layout = tiling.TileLayout(g.rows, g.cols, 2, 1)
    rng = np.random.default_rng(5)
    pts = np.c_[rng.uniform(-3.0, 3.0, 5000), rng.uniform(-30.0, 30.0, 5000), np.full(5000, 400.0)]
    cx, cy = tiling.cell_coords(pts, g)
    win = layout.window(0)
    own = np.ascontiguousarray(pts[tiling.owner_mask(cx, cy, win)])
    n, cap = own.shape[0], 16

    class _Alone(object):
        def exchange_rows(self, out_rows, in_rows, recv_counts, send_counts):
            out_rows.fill_(float("nan"))

    buf = torch.empty((n + 2 * cap, 3), dtype=torch.float64, device="cuda")
    buf[:n] = torch.from_numpy(own).cuda()
    with A.AerialGridMap(st, window=win) as m:
        t = tiling.TiledDsm(A.DsmSettings(), m, layout, 0, cap, comm=_Alone())
        # the selection does not fit its 16 send rows: the STEP fails (device-side check in the
        # finish call, surfaced by the synchronize), not a later audit
        with pytest.raises(A.AmhipError) as ei:
            t.process(buf, n)
        assert ei.value.status == A.hip_lib.ERR_HALO_OVERFLOW
        with pytest.raises(RuntimeError, match="halo rows"):
            t.check_overflow()
        with pytest.raises(A.AmhipError):            # finish without begin
            A.hip_lib.check(A.hip_lib.load().amhip_dsm_tiled_finish_dev(m.handle))

        # another DSM call between begin and finish cancels the pending one
        class _Intruder(object):
            def exchange_rows(self, out_rows, in_rows, recv_counts, send_counts):
                out_rows.fill_(float("nan"))
                A.Dsm(A.DsmSettings(), m).process(buf[:n], m)

        t = tiling.TiledDsm(A.DsmSettings(), m, layout, 0, 4096, comm=_Intruder())
        big = torch.empty((n + 2 * 4096, 3), dtype=torch.float64, device="cuda")
        big[:n] = buf[:n]
        with pytest.raises(A.AmhipError):
            t.process(big, n)
