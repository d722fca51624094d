"""One survey map tiled over several GPUs (one process per GPU).

The reference is a single process; this is new design.
Cells are independent once a rank holds every point within the LAST fallback
radius (< sqrt(7) m, dsm.cc:133-144) of its window, plus the frames.  So:

  * the map is cut into windows (`TileLayout`), every rank creates
    `AerialGridMap(settings, device, window=...)` -- cell positions, hence
    results, are those of the full map;
  * points are handed to the rank that owns their cell; the only data-path
    collective is the HALO exchange: every rank selects the points other
    windows need within `halo_margin()` metres (HIP kernel
    `amhip_halo_select_dev` on the GPU) and ships them with ONE all_to_all
    (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
    tests).  Volume: perimeter x margin x density, a few MB per rank -- latency
    bound, not bandwidth bound;
  * frames / poses are replicated; the mosaic needs no exchange.

`route_points()` is the general entry: it also works for clouds that arrive
partitioned by source instead of by tile (then it moves everything once).
"""
import ctypes as C
import math

import numpy as np

MAX_DESTS = 8


class TileLayout(object):
    """tiles_i x tiles_j windows of a (rows x cols) map; rank r owns window
    (r % tiles_i, r // tiles_i).  Window edges are multiples of `align` cells
    (the DSM gather tile is 64 x 32) except at the map border."""

    def __init__(self, rows, cols, tiles_i, tiles_j, align_i=64, align_j=32):
        self.rows, self.cols = int(rows), int(cols)
        self.tiles_i, self.tiles_j = int(tiles_i), int(tiles_j)
        self.edges_i = self._edges(self.rows, self.tiles_i, align_i)
        self.edges_j = self._edges(self.cols, self.tiles_j, align_j)

    @staticmethod
    def _edges(n, parts, align):
        edges = [0]
        for k in range(1, parts):
            e = int(round(n * k / float(parts) / align)) * align
            e = min(max(e, edges[-1] + 1), n - (parts - k))
            edges.append(e)
        edges.append(n)
        return edges

    @property
    def world(self):
        return self.tiles_i * self.tiles_j

    def window(self, rank):
        ti, tj = rank % self.tiles_i, rank // self.tiles_i
        i0, i1 = self.edges_i[ti], self.edges_i[ti + 1]
        j0, j1 = self.edges_j[tj], self.edges_j[tj + 1]
        return (i0, j0, i1 - i0, j1 - j0)

    def windows(self):
        return [self.window(r) for r in range(self.world)]

    @staticmethod
    def for_world(rows, cols, world):
        """Near-square factorisation, more tiles along the longer axis."""
        best = (1, world)
        for a in range(1, world + 1):
            if world % a == 0:
                b = world // a
                if abs(rows / float(a) - cols / float(b)) < \
                        abs(rows / float(best[0]) - cols / float(best[1])):
                    best = (a, b)
        return TileLayout(rows, cols, best[0], best[1])


def halo_margin(radius_sq, resolution):
    """Metres a window has to be grown by so that it contains every point any of
    its cells can reach: sqrt of the largest squared radius the ladder of
    dsm.cc:127-144 tries, plus one cell of slack."""
    tmax = float(radius_sq)
    lam = 1.0
    while True:
        tmax = max(tmax, lam * radius_sq)
        lam *= 1.1
        if lam * radius_sq > 7.0:
            break
    return math.sqrt(tmax) + resolution


def cell_coords(points_xy, grid, center_easting=0.0, center_northing=0.0):
    """Continuous cell coordinates (cell i has its centre at ci == i) of points
    in the frame the DSM bins them in (dsm.cc:42-43 offsets applied).  Works on
    numpy arrays and torch tensors."""
    base_x = grid.pos_x + (0.5 * grid.length_x - 0.5 * grid.resolution)
    base_y = grid.pos_y + (0.5 * grid.length_y - 0.5 * grid.resolution)
    inv = 1.0 / grid.resolution
    cx = (base_x - (points_xy[:, 0] - center_northing)) * inv
    cy = (base_y - (points_xy[:, 1] - center_easting)) * inv
    return cx, cy


def in_window(cx, cy, window, margin_cells=0.0):
    i0, j0, r, c = window
    return (cx >= i0 - 0.5 - margin_cells) & (cx <= i0 + r - 0.5 + margin_cells) & \
           (cy >= j0 - 0.5 - margin_cells) & (cy <= j0 + c - 0.5 + margin_cells)


def owner_mask(cx, cy, window, layout=None):
    """Points whose CELL lies in the window (half-open: each point has exactly
    one owner).  With `layout`, a window on the map's border also owns the points beyond that
    border (its interval is open-ended on the map's outer sides): the reference's kd-tree uses
    off-map points within the search radius for border cells (dsm.cc:36-52 keeps every point),
    and so does the single-GPU path here, so a tiled run must not lose them -- every point of
    a cloud that overhangs the map then has exactly one owner."""
    i0, j0, r, c = window
    lo_i, hi_i, lo_j, hi_j = i0 - 0.5, i0 + r - 0.5, j0 - 0.5, j0 + c - 0.5
    if layout is not None:
        inf = float("inf")
        if i0 == 0:
            lo_i = -inf
        if i0 + r == layout.rows:
            hi_i = inf
        if j0 == 0:
            lo_j = -inf
        if j0 + c == layout.cols:
            hi_j = inf
    return (cx >= lo_i) & (cx < hi_i) & (cy >= lo_j) & (cy < hi_j)


def select_for_windows(points, grid, windows, margin_m, center_easting=0.0,
                       center_northing=0.0, map_=None, cap=None):
    """For every window in `windows`: the points of `points` inside it grown by
    margin_m.  Returns (list of tensors).  CUDA tensors go through the HIP
    kernel of the context `map_` (<= 8 windows per call); CPU tensors / numpy
    arrays through plain masking (gloo tests)."""
    import torch
    if isinstance(points, np.ndarray):
        points = torch.from_numpy(points)
    if not points.is_cuda:
        cx, cy = cell_coords(points, grid, center_easting, center_northing)
        mc = margin_m / grid.resolution
        return [points[in_window(cx, cy, w, mc)] for w in windows]
    assert map_ is not None, "a device context is needed for CUDA clouds"
    from . import hip_lib as L
    lib = L.load()
    n = points.shape[0]
    out = []
    for lo in range(0, len(windows), MAX_DESTS):
        ws = windows[lo:lo + MAX_DESTS]
        nd = len(ws)
        if cap is None:
            # perimeter strip estimate with generous slack, never more than n
            cap_d = n
        else:
            cap_d = int(cap)
        cap_d = max(cap_d, 1)
        buf = torch.empty((nd, cap_d, 3), dtype=torch.float64, device=points.device)
        counts = torch.zeros(nd, dtype=torch.int64, device=points.device)
        wins = (C.c_int32 * (4 * nd))(*[int(v) for w in ws for v in w])
        map_.wait_for_torch(points)
        L.check(lib.amhip_halo_select_dev(
            map_.handle, C.c_void_p(points.data_ptr()), n, center_easting, center_northing,
            wins, nd, float(margin_m), C.c_void_p(buf.data_ptr()), cap_d,
            C.c_void_p(counts.data_ptr())))
        map_.synchronize()  # the kernel ran on the context's stream
        cnt = counts.cpu().tolist()
        for d in range(nd):
            if cnt[d] > cap_d:
                raise RuntimeError("halo buffer too small: %d > %d" % (cnt[d], cap_d))
            out.append(buf[d, :cnt[d]])
    return out


class TorchComm(object):
    """The two collectives route_points() needs, over torch.distributed
    (backend nccl = RCCL on the GPUs, gloo in the CPU tests).  via_host=True
    stages device tensors through host memory: only for rehearsing the N > 1
    code path with several ranks on ONE GPU over gloo (bench.py,
    AMHIP_BENCH_ONE_GPU=1) -- RCCL refuses two ranks on one device."""

    def __init__(self, group=None, via_host=False):
        self.group = group
        self.via_host = via_host

    def exchange_counts(self, send_counts_tensor):
        import torch
        import torch.distributed as dist
        if self.via_host:
            src = send_counts_tensor.cpu()
            recv = torch.empty_like(src)
            dist.all_to_all_single(recv, src, group=self.group)
            return recv.to(send_counts_tensor.device)
        recv = torch.empty_like(send_counts_tensor)
        dist.all_to_all_single(recv, send_counts_tensor, group=self.group)
        return recv

    def exchange_equal(self, out_rows, in_rows):
        """all_to_all of equal splits (TiledDsm): block r of in_rows goes to rank r."""
        import torch
        import torch.distributed as dist
        if self.via_host:
            tmp = torch.empty(out_rows.shape, dtype=out_rows.dtype)
            dist.all_to_all_single(tmp, in_rows.cpu(), group=self.group)
            out_rows.copy_(tmp)
            return
        dist.all_to_all_single(out_rows, in_rows, group=self.group)

    def exchange_rows(self, out_rows, in_rows, recv_counts, send_counts):
        import torch
        import torch.distributed as dist
        if self.via_host:
            tmp = torch.empty(out_rows.shape, dtype=out_rows.dtype)
            dist.all_to_all_single(tmp, in_rows.cpu(), output_split_sizes=recv_counts,
                                   input_split_sizes=send_counts, group=self.group)
            out_rows.copy_(tmp)
            return
        dist.all_to_all_single(out_rows, in_rows, output_split_sizes=recv_counts,
                               input_split_sizes=send_counts, group=self.group)


def route_points(points, grid, layout, rank, group=None, radius_sq=1, center_easting=0.0,
                 center_northing=0.0, map_=None, assume_owned=False, cap=None,
                 workspace=None, comm=None):
    """Exchange points so that this rank ends up with every point inside its
    window grown by the halo margin.

    points        (N,3) float64 tensor this rank currently holds (any subset of
                  the cloud; CUDA for nccl/RCCL, CPU for gloo)
    assume_owned  True: all of `points` already belong to this rank's window
                  (pre-partitioned cloud) -> they are kept in place and only the
                  halo strips travel.
    workspace     optional (M,3) tensor whose first N rows ARE `points` (same
                  storage) and M - N >= the halo volume: the received points are
                  written behind the owned ones and no copy of the cloud is made
                  (only with assume_owned).
    Returns (N',3): kept points followed by the received ones.
    """
    import torch
    world = layout.world
    comm = comm or TorchComm(group)
    margin = halo_margin(radius_sq, grid.resolution)
    if isinstance(points, np.ndarray):
        points = torch.from_numpy(points)
    if world == 1:
        return points
    wins = layout.windows()
    others = [r for r in range(world) if r != rank]
    if assume_owned:
        keep = points
        sends = select_for_windows(points, grid, [wins[r] for r in others], margin,
                                   center_easting, center_northing, map_, cap)
    else:
        sel = select_for_windows(points, grid, wins, margin, center_easting, center_northing,
                                 map_, cap)
        keep = sel[rank]
        sends = [sel[r] for r in others]
    send_counts = [0] * world
    for r, t in zip(others, sends):
        send_counts[r] = int(t.shape[0])
    # 1) counts, 2) payload: ONE all_to_all each
    sc = torch.tensor(send_counts, dtype=torch.int64, device=points.device)
    rc = comm.exchange_counts(sc)
    recv_counts = [int(v) for v in rc.cpu().tolist()]
    send_buf = torch.cat([t.reshape(-1, 3) for t in sends], 0) if sends else points[:0]
    total_recv = sum(recv_counts)
    nk = keep.shape[0]
    if workspace is not None and assume_owned and workspace.data_ptr() == points.data_ptr() \
            and workspace.shape[0] >= nk + total_recv:
        out = workspace[:nk + total_recv]
    else:
        out = torch.empty((nk + total_recv, 3), dtype=points.dtype, device=points.device)
        out[:nk] = keep
    recv_view = out[nk:]
    comm.exchange_rows(recv_view, send_buf.contiguous(), recv_counts, send_counts)
    return out


def neighbours(layout, rank, margin_m, resolution, slack_cells=1.0):
    """Ranks whose window, grown by the halo margin, can hold a point of `rank`'s window (and
    vice versa: the relation is symmetric).  Host geometry -- at most 8 in a 2-D tiling.

    Same arithmetic as the device selection (make_halo_params / k_halo_select): a point OWNED by
    `rank` has continuous cell coordinates in [i0 - 0.5, i0 + r - 0.5) x [j0 - 0.5, j0 + c - 0.5);
    window q takes it iff  a0 - 0.5 - mc <= cx <= a0 + ar - 0.5 + mc  (same for y) with
    mc = margin_m / resolution -- i.e. iff  a0 - mc < i0 + r  and  a0 + ar + mc >= i0.  The
    comparison here is non-strict and `slack_cells` wider: a SUPERSET of the selection's
    destinations, never less (an extra neighbour only receives NaN rows)."""
    mc = float(margin_m) / float(resolution) + float(slack_cells)
    i0, j0, r, c = layout.window(rank)
    out = []
    for q in range(layout.world):
        if q == rank:
            continue
        a0, b0, ar, ac = layout.window(q)
        if a0 - mc <= i0 + r and a0 + ar + mc >= i0 and \
                b0 - mc <= j0 + c and b0 + ac + mc >= j0:
            out.append(q)
    return out


class TiledDsm(object):
    """dsm::Dsm::process of one window of a tiled map, the halo exchange folded in
    (amhip_dsm_tiled_begin_dev / _finish_dev): the DSM's binning pass selects the
    points the NEIGHBOURING windows need on the way, ONE all_to_all ships them --
    `cap` rows to / from every geometric neighbour, nothing to anyone else (split
    sizes are host geometry: no count exchange; unused rows are NaN and dropped by the
    receiver's binning) -- and nothing synchronises with the host.  This replaces the
    halo all-reduce of BASELINE.json's wording by the exact option SURVEY 8e (i): halo
    POINTS travel, every window then runs the single-GPU kernels unchanged.  A
    selection that does not fit its `cap` rows raises AMHIP_ERR_HALO_OVERFLOW at the
    next synchronize (device-side check in the finish call: the step fails, not a
    later audit).

    workspace  (n_owned + len(neighbours) * cap, 3) float64 CUDA tensor whose first
               n_owned rows are this rank's own points; the received rows land behind
    cap        rows per (source, destination) pair: a few times edge x margin x
               density (halo_strip_rows())

    PRECONDITION: the n_owned rows are points whose CELL lies in this rank's window
    (owner_mask(..., layout): a border window also owns what lies beyond its outer sides).  Only geometric neighbours are destinations, so a stray point that some
    non-neighbour needs would never travel; process() therefore verifies the precondition on
    the device (check_owned: default = on its first call, True = every call) and raises instead
    of dropping points silently.  Clouds partitioned any other way go through route_points().
    """

    def __init__(self, settings, map_, layout, rank, cap, comm=None):
        import torch
        self.settings, self.map, self.layout, self.rank = settings, map_, layout, rank
        self.cap = int(cap)
        self.comm = comm or TorchComm()
        dev = torch.device("cuda", map_.device)
        self._margin = halo_margin(settings.interpolation_radius, map_.grid.resolution)
        self.nbrs = neighbours(layout, rank, self._margin, map_.grid.resolution)
        self._owned_checked = False
        nn = len(self.nbrs)
        if nn > MAX_DESTS:
            raise ValueError("more than %d neighbouring windows" % MAX_DESTS)
        self.recv_rows = nn * self.cap
        self.send = torch.empty((max(nn, 1) * self.cap, 3), dtype=torch.float64, device=dev)
        self.counts = torch.zeros(max(nn, 1), dtype=torch.int64, device=dev)
        wins = [layout.window(q) for q in self.nbrs]   # ascending rank = the all_to_all's order
        self._wins = (C.c_int32 * (4 * max(nn, 1)))(*[int(v) for w in wins for v in w])
        self.splits = [self.cap if q in self.nbrs else 0 for q in range(layout.world)]

    def verify_owned(self, workspace, n_owned):
        """Raise unless every one of the first n_owned rows lies in this rank's window (a pass
        of torch ops over the rows: not part of a steady-state step)."""
        s, m = self.settings, self.map
        cx, cy = cell_coords(workspace[:n_owned], m.grid, s.center_easting, s.center_northing)
        # (border windows own the points beyond their outer sides: a cloud that overhangs the map
        # is legitimate -- only a point in ANOTHER window's territory can fail to travel)
        stray = int((~owner_mask(cx, cy, self.layout.window(self.rank), self.layout)).sum().item())
        if stray:
            raise ValueError("TiledDsm: %d of the %d owned points lie outside this rank's window; "
                             "only neighbouring windows receive halo rows -- route such clouds "
                             "with route_points(assume_owned=False)" % (stray, n_owned))

    def process(self, workspace, n_owned, sync=True, check_owned=None):
        import torch
        from . import hip_lib as L
        lib = L.load()
        cap, m, s = self.cap, self.map, self.settings
        nn = len(self.nbrs)
        n_total = n_owned + nn * cap
        assert workspace.is_cuda and workspace.dtype == torch.float64 and workspace.is_contiguous()
        assert workspace.shape[0] >= n_total and workspace.shape[1] == 3
        if check_owned or (check_owned is None and not self._owned_checked):
            self.verify_owned(workspace, n_owned)
            self._owned_checked = True
        if nn == 0:
            from .mapper import Dsm
            Dsm(s, m).process(workspace[:n_owned], m, sync=sync)
            return
        self.send.fill_(float("nan"))
        m.wait_for_torch(workspace)
        m._touched("elevation")
        L.check(lib.amhip_dsm_tiled_begin_dev(
            m.handle, C.c_void_p(workspace.data_ptr()), n_owned, n_total, s.interpolation_radius,
            s.center_easting, s.center_northing, self._wins, nn, float(self._margin),
            C.c_void_p(self.send.data_ptr()), cap, C.c_void_p(self.counts.data_ptr())))
        m.torch_waits()
        self.comm.exchange_rows(workspace[n_owned:n_total], self.send, self.splits, self.splits)
        m.wait_for_torch(workspace)
        L.check(lib.amhip_dsm_tiled_finish_dev(m.handle))
        if sync:
            m.synchronize()   # raises AMHIP_ERR_HALO_OVERFLOW if a selection did not fit
        else:
            # (the gather still reads `workspace`: see Dsm.process -- ordered automatically when the
            # map runs on torch's current stream, otherwise wait)
            m.torch_waits()

    def check_overflow(self):
        """(kept for callers that run unsynchronised steps) the same check from the counts."""
        worst = int(self.counts.max().item())
        if worst > self.cap:
            raise RuntimeError("halo rows: %d points for one window, capacity %d" % (worst, self.cap))


def halo_strip_rows(points_per_m2, edge_m, radius_sq, resolution, slack=1.5):
    """Rows to reserve per (source, destination) pair of TiledDsm: the points of a strip
    one halo margin deep along the longest shared edge, with slack."""
    return int(slack * points_per_m2 * edge_m * halo_margin(radius_sq, resolution)) + 4096
