"""world_size-2 (and 4) CPU tests of the multi-GPU path: tile layout, halo
routing over torch.distributed (gloo here, RCCL on the GPUs), and the property
that makes tiling exact: a tile's DSM computed from the routed subset equals
the full-cloud DSM on that tile's cells."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tiles, assume_owned, via_host, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from aerial_mapper_amd import synth, tiling
    import oracle_ffi as OO
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = OO.make_grid(160.0, 128.0, 0.5)
        layout = tiling.TileLayout(g.rows, g.cols, tiles[0], tiles[1])
        assert layout.world == world
        cloud = synth.make_points(60000, 90.0, 77)  # spills over the map border
        cx, cy = tiling.cell_coords(cloud, g)
        if assume_owned:
            # (border windows own the points beyond the map's border: owner_mask(..., layout))
            mine = cloud[tiling.owner_mask(cx, cy, layout.window(rank), layout)]
        else:
            mine = cloud[rank::world]  # partitioned by source, not by tile
        got = tiling.route_points(torch.from_numpy(np.ascontiguousarray(mine)), g, layout, rank,
                                  radius_sq=1, assume_owned=assume_owned,
                                  comm=tiling.TorchComm(via_host=via_host)).numpy()
        margin = tiling.halo_margin(1, g.resolution)
        want_mask = tiling.in_window(cx, cy, layout.window(rank), margin / g.resolution)
        if assume_owned:
            # a pre-partitioned cloud keeps ALL of a rank's owned points, including the off-map
            # ones beyond its grown window (the binning drops those later)
            want_mask |= tiling.owner_mask(cx, cy, layout.window(rank), layout)
            owners = sum(tiling.owner_mask(cx, cy, layout.window(r), layout).astype(int) for r in range(world))
            assert (owners == 1).all() and ((cx < -0.5) | (cy < -0.5)).any()
        want = cloud[want_mask]
        key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        same_set = got.shape == want.shape and np.array_equal(key(got), key(want))
        # exactness of tiling: DSM of the routed subset == DSM of the whole cloud
        # on this window's cells (oracle, full-map geometry)
        # (the WHOLE cloud, off-map points included: the reference's kd-tree keeps them all)
        full = OO.dsm_process(cloud, g)[1]
        part = OO.dsm_process(got, g)[1]
        i0, j0, r, c = layout.window(rank)
        a, b = full[j0:j0 + c, i0:i0 + r], part[j0:j0 + c, i0:i0 + r]
        # same neighbour sets; the kd-tree visiting order (hence the last bits
        # of the double sums) may differ between the two trees
        nan_ok = np.array_equal(np.isnan(a), np.isnan(b))
        err = float(np.nanmax(np.abs(a.astype(np.float64) - b))) if (~np.isnan(a)).any() else 0.0
        ret[rank] = (same_set, nan_ok, err, int(got.shape[0]), int((~np.isnan(a)).sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tiles,assume_owned,via_host",
                         [(2, (2, 1), True, False), (2, (1, 2), False, False), (4, (2, 2), True, False),
                          (2, (2, 1), True, True), (8, (2, 4), True, False)])   # (8: the driver node's 2 x 4)
def test_route_points_gloo(world, tiles, assume_owned, via_host):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tiles, assume_owned, via_host, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(world):
        same_set, nan_ok, err, n, filled = ret[r]
        assert same_set, "rank %d holds the wrong point set" % r
        assert nan_ok and err <= 1e-6 and n > 0 and filled > 0, (r, nan_ok, err, n, filled)


def test_tile_layout_covers_map_once():
    from aerial_mapper_amd import tiling
    for rows, cols, ti, tj in [(10000, 10000, 2, 4), (1000, 777, 3, 2), (64, 32, 1, 1),
                               (40000, 10000, 8, 1)]:
        lay = tiling.TileLayout(rows, cols, ti, tj)
        cover = np.zeros((rows, cols), np.int32) if rows * cols < 5e7 else None
        area = 0
        for r in range(lay.world):
            i0, j0, nr, nc = lay.window(r)
            assert nr > 0 and nc > 0 and i0 + nr <= rows and j0 + nc <= cols
            area += nr * nc
            if cover is not None:
                cover[i0:i0 + nr, j0:j0 + nc] += 1
        assert area == rows * cols
        if cover is not None:
            assert (cover == 1).all()
    lay = tiling.TileLayout.for_world(40000, 10000, 8)
    assert (lay.tiles_i, lay.tiles_j) == (8, 1) or lay.tiles_i * lay.tiles_j == 8


def test_halo_margin_is_last_ladder_radius():
    from aerial_mapper_amd import tiling
    assert abs(tiling.halo_margin(1, 0.25) - (np.sqrt(1.1 ** 20) + 0.25)) < 1e-9
    assert abs(tiling.halo_margin(9, 1.0) - (3.0 + 1.0)) < 1e-12


def _worker_neighbours(rank, world, port, tiles, ret):
    """The exchange protocol of tiling.TiledDsm without the GPU: every rank fills `cap` rows per
    GEOMETRIC neighbour (what k_dsm_p3_count<true> / k_halo_select do on the device, here by
    masking), NaN-pads them, and ONE all_to_all with split sizes cap / 0 ships them; the NaN
    rows are dropped like the receiver's binning drops them."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from aerial_mapper_amd import synth, tiling
    import oracle_ffi as OO
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = OO.make_grid(192.0, 160.0, 0.5)
        layout = tiling.TileLayout(g.rows, g.cols, tiles[0], tiles[1])
        cloud = synth.make_points(70000, 100.0, 78)
        cx, cy = tiling.cell_coords(cloud, g)
        own_mask = tiling.owner_mask(cx, cy, layout.window(rank), layout)
        margin = tiling.halo_margin(1, g.resolution)
        mc = margin / g.resolution
        nbrs = tiling.neighbours(layout, rank, margin, g.resolution)
        cap = 6000
        send = np.full((max(len(nbrs), 1) * cap, 3), np.nan)
        for k, q in enumerate(nbrs):
            sel = cloud[own_mask & tiling.in_window(cx, cy, layout.window(q), mc)]
            assert sel.shape[0] <= cap
            send[k * cap:k * cap + sel.shape[0]] = sel
        splits = [cap if q in nbrs else 0 for q in range(world)]
        recv = torch.empty((len(nbrs) * cap, 3), dtype=torch.float64)
        tiling.TorchComm().exchange_rows(recv, torch.from_numpy(send[:len(nbrs) * cap]), splits, splits)
        got = recv.numpy()
        got = got[~np.isnan(got[:, 0])]
        want = cloud[tiling.in_window(cx, cy, layout.window(rank), mc) & ~own_mask]
        key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        # a rank that is NOT a neighbour holds nothing this window needs
        for q in range(world):
            if q != rank and q not in nbrs:
                stray = tiling.owner_mask(cx, cy, layout.window(q), layout) & \
                    tiling.in_window(cx, cy, layout.window(rank), mc)
                assert not stray.any()
        ret[rank] = (got.shape == want.shape and np.array_equal(key(got), key(want)), len(nbrs),
                     int(got.shape[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tiles,max_nbrs", [(2, (2, 1), 1), (4, (2, 2), 3), (6, (3, 2), 5),
                                                  (4, (4, 1), 2), (8, (2, 4), 5)])
def test_neighbour_only_exchange_gloo(world, tiles, max_nbrs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_neighbours, args=(r, world, port, tiles, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(world):
        ok, nn, n = ret[r]
        assert ok, "rank %d received the wrong halo set" % r
        assert 1 <= nn <= max_nbrs and n > 0


def test_neighbours_of_a_node_sized_layout():
    from aerial_mapper_amd import tiling
    lay = tiling.TileLayout(40000, 40000, 2, 4)           # configs[3]: 2 x 4 windows
    mm = tiling.halo_margin(1, 0.25)
    for r in range(8):
        nb = tiling.neighbours(lay, r, mm, 0.25)
        ti, tj = r % 2, r // 2
        want = sorted(q for q in range(8) if q != r and abs(q % 2 - ti) <= 1 and abs(q // 2 - tj) <= 1)
        assert nb == want and len(nb) <= tiling.MAX_DESTS
        for q in nb:                                       # symmetric
            assert r in tiling.neighbours(lay, q, mm, 0.25)
    strip = tiling.TileLayout(80000, 10000, 8, 1)          # the weak-scaling strip of bench.py
    assert [len(tiling.neighbours(strip, r, mm, 0.25)) for r in range(8)] == [1, 2, 2, 2, 2, 2, 2, 1]


def test_neighbour_relation_covers_the_device_selection_at_exact_boundaries():
    """neighbours() against the selection's own predicate (make_halo_params / k_halo_select:
    a0 - 0.5 - mc <= cx <= a0 + ar - 0.5 + mc) for owned points sitting EXACTLY on the bounds,
    incl. margins that are whole numbers of cells (a float margin compared with
    integer edges)."""
    from aerial_mapper_amd import tiling
    for res, radius_sq in [(0.25, 1), (0.5, 1), (1.0, 9), (0.2, 4)]:
        margin = tiling.halo_margin(radius_sq, res)
        for margin_m in (margin, 3.0 * res, 11.0 * res):       # whole-cell margins too
            mc = margin_m / res
            lay = tiling.TileLayout(1024, 640, 4, 3)
            for r in range(lay.world):
                i0, j0, nr, nc = lay.window(r)
                nb = set(tiling.neighbours(lay, r, margin_m, res))
                # extreme owned positions (the half-open cell range of the window)
                xs = [i0 - 0.5, np.nextafter(i0 + nr - 0.5, -np.inf)]
                ys = [j0 - 0.5, np.nextafter(j0 + nc - 0.5, -np.inf)]
                for q in range(lay.world):
                    if q == r:
                        continue
                    a0, b0, ar, ac = lay.window(q)
                    takes = any(a0 - 0.5 - mc <= x <= a0 + ar - 0.5 + mc and
                                b0 - 0.5 - mc <= y <= b0 + ac - 0.5 + mc
                                for x in xs + [min(max(a0 - 0.5 - mc, xs[0]), xs[1])]
                                for y in ys + [min(max(b0 - 0.5 - mc, ys[0]), ys[1])])
                    if takes:
                        assert q in nb, (res, margin_m, r, q)
