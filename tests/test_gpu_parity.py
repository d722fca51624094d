"""GPU parity tests: the HIP path, driven through the C ABI, against the CPU
oracle on identical seeded inputs.  Run on the MI355X box: pytest -m gpu.

Tolerances (BASELINE.json north_star): DSM heights within 1e-4 m with an
identical NaN pattern; ortho layers cell for cell -- observation_index and the
sampled pixel exact (stricter than the 1-LSB allowance), elevation_angle
bit-identical floats.
"""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S
from aerial_mapper_amd import synth

pytestmark = pytest.mark.gpu

ORTHO_LAYERS = ["elevation_angle", "observation_index", "num_observations", "ortho",
                "colored_ortho"]


def _A():
    import aerial_mapper_amd as A
    return A


# Every DSM test runs in both arithmetic modes of the gather (amhip_ctx_set_dsm_precision):
# "exact" (the library's default: FP64 everywhere, the reference's floats) and "fast" (opt-in:
# single precision under exact guards).  The bar is the same -- identical NaN pattern, heights
# within 1e-4 m -- only the share of bit-identical floats differs.
_EXACT = False


@pytest.fixture(autouse=True, params=["fast", "exact"])
def dsm_mode(request):
    global _EXACT
    if request.param == "exact" and not request.node.name.startswith("test_dsm"):
        pytest.skip("not a DSM test: one mode is enough")
    _EXACT = request.param == "exact"
    yield request.param
    _EXACT = False


def _map_for(scene, A):
    g = scene.grid
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    m = A.AerialGridMap(st)
    m.set_dsm_precision(_EXACT)
    assert (m.rows, m.cols) == (g.rows, g.cols)
    return m


def _dsm_both(scene, radius=1, ce=0.0, cn=0.0, elevation0=None):
    A = _A()
    rc, want, _ = O.dsm_process(scene.points, scene.grid, radius, ce, cn,
                                elevation=None if elevation0 is None else elevation0.copy())
    assert rc == O.OK
    with _map_for(scene, A) as m:
        if elevation0 is not None:
            m.set("elevation", elevation0)
        A.Dsm(A.DsmSettings(radius, False, ce, cn), m).process(scene.points, m)
        got = m.get("elevation")
    return got, want


def test_library_loaded_is_in_tree():
    from aerial_mapper_amd import hip_lib
    lib = hip_lib.load()
    assert hip_lib.LIB_PATH.endswith("aerial_mapper_amd/lib/libaerial_mapper_hip.so")
    assert lib.amhip_abi_version() == hip_lib.ABI_VERSION


def test_dsm_sparse_1m_grid_with_fallback():
    # config-1 shape scaled down: ~1 pt/cell at 1.0 m, ~4 % of the cells need
    # the expanding-radius fallback (dsm.cc:133-144)
    sc = S.Scene(300.0, 200.0, 1.0, 66000, seed=42)
    got, want = _dsm_both(sc)
    frac = S.assert_dsm_close(got, want)
    # measured: 1.0 (FP64) / 0.9971 (single precision: f32 sums + the records' once-more-rounded
    # height offsets move a float spacing in 3 cells per thousand)
    assert frac > (0.9999 if _EXACT else 0.99)


def test_dsm_dense_quarter_metre():
    # config-2 density: 8 pts/m^2 at 0.25 m (radius = 4 cells)
    sc = S.Scene(60.0, 45.0, 0.25, int(8 * 68 * 68), seed=43, point_extent=34.0)
    got, want = _dsm_both(sc)
    assert not np.isnan(want).any()
    S.assert_dsm_close(got, want)


def test_dsm_holes_and_untouched_cells():
    # points only in the left third: most cells find nothing within the last
    # fallback radius and must stay untouched (NaN), the border band walks the
    # whole ladder
    sc = S.Scene(120.0, 90.0, 0.5, 20000, seed=44)
    keep = sc.points[:, 0] < -20.0
    sc.points = np.ascontiguousarray(sc.points[keep])
    got, want = _dsm_both(sc)
    assert np.isnan(want).sum() > 1000 and (~np.isnan(want)).sum() > 1000
    S.assert_dsm_close(got, want)


def test_dsm_very_sparse_every_level():
    # ~0.05 pts/m^2: nearest neighbours are spread over all 21 ladder levels
    sc = S.Scene(200.0, 160.0, 1.0, 1800, seed=45)
    got, want = _dsm_both(sc)
    assert np.isnan(want).any() and (~np.isnan(want)).any()
    S.assert_dsm_close(got, want)


def test_dsm_radius_2_and_9():
    sc = S.Scene(90.0, 70.0, 0.5, 30000, seed=46)
    for radius in (2, 9):
        got, want = _dsm_both(sc, radius=radius)
        S.assert_dsm_close(got, want)


def test_dsm_center_offsets_swap_quirk():
    # dsm.cc:42-43 subtracts center_NORTHING from x and center_EASTING from y
    ce, cn = 37.5, -12.25
    sc = S.Scene(80.0, 60.0, 0.5, 25000, seed=47, center=(ce, cn), point_extent=60.0)
    # cloud lives around (cn + ce, ce + cn) so that the quirk keeps it on the map
    sc.points[:, 0] += cn
    sc.points[:, 1] += ce
    got, want = _dsm_both(sc, ce=ce, cn=cn)
    assert (~np.isnan(want)).sum() > 100
    S.assert_dsm_close(got, want)


def test_dsm_incremental_keeps_previous_values():
    # second process() with a cloud covering part of the map overwrites only
    # cells that find neighbours (main-ortho-backward-grid-incremental.cc:153)
    sc = S.Scene(100.0, 80.0, 0.5, 40000, seed=48)
    first, want1 = _dsm_both(sc)
    S.assert_dsm_close(first, want1)
    sc2 = S.Scene(100.0, 80.0, 0.5, 9000, seed=49)
    sc2.points = np.ascontiguousarray(sc2.points[sc2.points[:, 1] > 10.0])
    sc2.points[:, 2] += 3.0
    got, want = _dsm_both(sc2, elevation0=want1)
    S.assert_dsm_close(got, want)
    assert (want == want1).any() and (want != want1).any()


def test_dsm_exact_hit_is_reported():
    A = _A()
    sc = S.Scene(40.0, 30.0, 1.0, 3000, seed=50)
    x, y = O.cell_position(sc.grid, 7, 11)
    sc.points[5] = (x, y, 400.0)
    rc, _, _ = O.dsm_process(sc.points, sc.grid)
    assert rc == O.ERR_EXACT_HIT
    with _map_for(sc, A) as m:
        with pytest.raises(A.AmhipError) as ei:
            A.Dsm(A.DsmSettings(), m).process(sc.points, m)
        assert ei.value.status == 2  # AMHIP_ERR_EXACT_HIT
        # the sticky status is cleared by the failing call
        m.synchronize()


def test_dsm_device_path_matches_host_path():
    import torch
    A = _A()
    sc = S.Scene(120.0, 100.0, 0.5, 60000, seed=51)
    with _map_for(sc, A) as m:
        d = A.Dsm(A.DsmSettings(), m)
        d.process(sc.points, m)
        host = m.get("elevation")
        m.reset()
        d.process(torch.from_numpy(sc.points).cuda(), m)
        dev = m.get("elevation")
        st = m.dsm_stats()
    assert st["points_binned"] > 0 and st["points_binned"] <= sc.points.shape[0]
    # same neighbour sets; only the (atomic) order inside a bin may differ: 1e-16 relative in
    # the FP64 sums, up to one spacing of the stored float (3e-5 m here) in the f32 sums
    S.assert_dsm_close(dev, host, tol=1e-6 if _EXACT else 1e-4)


def _ortho_both(scene, batches=None, cam=None, elevation=None):
    """Runs DSM(oracle) -> ortho on both sides over the same elevation layer."""
    A = _A()
    cam = cam or scene.cam
    if elevation is None:
        rc, elevation, _ = O.dsm_process(scene.points, scene.grid)
        assert rc == O.OK
    layers = O.new_layers(scene.grid)
    layers["elevation"] = elevation.copy()
    F = len(scene.frames)
    batches = batches or [(0, F)]
    with _map_for(scene, A) as m:
        m.set("elevation", elevation)
        nc = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height,
                       cam.distortion, tuple(cam.dist), scene.T_C_B)
        mosaic = A.OrthoBackwardGrid(nc, A.OrthoSettings(colored_ortho=scene.colored), m)
        for lo, hi in batches:
            rc = O.ortho_process(scene.grid, cam, scene.poses[lo:hi], scene.T_C_B,
                                 scene.frames[lo:hi], layers, colored=scene.colored)
            assert rc == O.OK
            mosaic.process(scene.poses[lo:hi], scene.frames[lo:hi], m)
        got = {n: m.get(n) for n in ORTHO_LAYERS}
    return got, layers


def _coverage(layers):
    return float((~np.isnan(layers["observation_index"])).mean())


def test_ortho_gray_batch():
    sc = S.Scene(150.0, 110.0, 0.5, 120000, seed=60, num_frames=14, altitude=480.0)
    got, want = _ortho_both(sc)
    assert 0.3 < _coverage(want) <= 1.0
    S.assert_layers_equal(got, want, ORTHO_LAYERS)


def test_ortho_colored_batch():
    sc = S.Scene(120.0, 100.0, 0.5, 90000, seed=61, num_frames=10, altitude=470.0,
                 colored=True)
    got, want = _ortho_both(sc)
    assert _coverage(want) > 0.2
    S.assert_layers_equal(got, want, ORTHO_LAYERS)
    # gray layer untouched by the colour path
    assert (want["ortho"] == 255.0).all()


def test_ortho_incremental_batches_persist():
    # observation_index is the index WITHIN the batch (ortho-backward-grid.cc:182)
    sc = S.Scene(150.0, 110.0, 0.5, 120000, seed=62, num_frames=15, altitude=480.0)
    got, want = _ortho_both(sc, batches=[(0, 5), (5, 11), (11, 15)])
    S.assert_layers_equal(got, want, ORTHO_LAYERS)


def test_ortho_nan_elevation_stays_untouched():
    sc = S.Scene(120.0, 90.0, 0.5, 20000, seed=63, num_frames=8, altitude=500.0)
    sc.points = np.ascontiguousarray(sc.points[sc.points[:, 0] < 0.0])
    got, want = _ortho_both(sc)
    nan_elev = np.isnan(want["elevation"])
    assert nan_elev.any()
    assert (want["ortho"][nan_elev] == 255.0).all()
    S.assert_layers_equal(got, want, ORTHO_LAYERS)


@pytest.mark.parametrize("kind", ["radtan", "equidistant"])
def test_ortho_distortion_models(kind):
    if kind == "radtan":
        cam = S.camera(distortion=O.DIST_RADTAN, dist=(-0.28, 0.07, 2e-4, -1e-4))
    else:
        cam = S.camera(distortion=O.DIST_EQUIDISTANT, dist=(-0.01, 0.02, -0.005, 0.001))
    sc = S.Scene(90.0, 70.0, 0.5, 50000, seed=64, num_frames=6, altitude=470.0, cam=cam)
    got, want = _ortho_both(sc)
    assert _coverage(want) > 0.1
    mism = {n: int(((got[n].view(np.uint32) != want[n].view(np.uint32)) &
                    ~(np.isnan(got[n]) & np.isnan(want[n]))).sum()) for n in ORTHO_LAYERS}
    if kind == "radtan":
        assert sum(mism.values()) == 0, mism
    else:
        # atan comes from two different libms (glibc / ROCm device libs):
        # allow a handful of last-bit flips, never a different frame
        assert mism["observation_index"] <= 2 and mism["ortho"] <= 2, mism


def test_ortho_num_observations_doubles():
    # `num_observations += num_observations` doubles a non-zero layer once per
    # accepted update (ortho-backward-grid.cc:183)
    A = _A()
    sc = S.Scene(80.0, 60.0, 1.0, 12000, seed=65, num_frames=6, altitude=470.0)
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    layers = O.new_layers(sc.grid)
    layers["elevation"] = elevation.copy()
    layers["num_observations"][:] = 1.5
    O.ortho_process(sc.grid, sc.cam, sc.poses, sc.T_C_B, sc.frames, layers)
    with _map_for(sc, A) as m:
        m.set("elevation", elevation)
        m.set("num_observations", np.full_like(elevation, 1.5))
        nc = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
        A.OrthoBackwardGrid(nc, A.OrthoSettings(), m).process(sc.poses, sc.frames, m)
        got = {n: m.get(n) for n in ORTHO_LAYERS}
    assert (layers["num_observations"] > 1.5).any()
    S.assert_layers_equal(got, layers, ORTHO_LAYERS)


def test_ortho_device_frames_match_host_frames():
    import torch
    A = _A()
    sc = S.Scene(100.0, 80.0, 0.5, 60000, seed=66, num_frames=9, altitude=480.0)
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    nc = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
    outs = []
    for dev in (False, True):
        with _map_for(sc, A) as m:
            m.set("elevation", elevation)
            mosaic = A.OrthoBackwardGrid(nc, A.OrthoSettings(), m)
            imgs = torch.from_numpy(np.stack(sc.frames)).cuda() if dev else sc.frames
            mosaic.process(sc.poses, imgs, m)
            outs.append({n: m.get(n) for n in ORTHO_LAYERS})
    S.assert_layers_equal(outs[1], outs[0], ORTHO_LAYERS)


def test_pipeline_dsm_then_ortho_on_device():
    """The bench's shape at toy size: device-resident cloud and frames, DSM
    feeding the mosaic without leaving HBM."""
    import torch
    A = _A()
    sc = S.Scene(150.0, 120.0, 0.25, int(8 * 160 * 160), seed=67, num_frames=12,
                 altitude=470.0, point_extent=80.0)
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    layers = O.new_layers(sc.grid)
    layers["elevation"] = elevation.copy()
    O.ortho_process(sc.grid, sc.cam, sc.poses, sc.T_C_B, sc.frames, layers)
    with _map_for(sc, A) as m:
        pts = torch.from_numpy(sc.points).cuda()
        imgs = torch.from_numpy(np.stack(sc.frames)).cuda()
        A.Dsm(A.DsmSettings(), m).process(pts, m, sync=False)
        nc = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
        A.OrthoBackwardGrid(nc, A.OrthoSettings(), m).process(sc.poses, imgs, m, sync=False)
        m.synchronize()
        got_elev = m.get("elevation")
        got = {n: m.get(n) for n in ORTHO_LAYERS}
    S.assert_dsm_close(got_elev, elevation)
    # the mosaic ran on the GPU's own DSM: where both DSMs are bit-identical the
    # ortho layers must be too
    same = got_elev.view(np.uint32) == elevation.view(np.uint32)
    # (FP64 mode: the reference's floats; single-precision mode: within 1e-4 m -- checked above --,
    # most of them still the reference's floats)
    assert same.mean() > (0.9999 if _EXACT else 0.99)
    for n in ORTHO_LAYERS:
        a, b = got[n][same], layers[n][same]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.mean() > 0.9999, (n, float(eq.mean()))


def test_dsm_fine_grid_takes_global_gather():
    # 0.05 m cells: the first search radius spans 20 cells > kMaxW0, so every
    # cell goes through the global-memory gather kernel
    sc = S.Scene(12.0, 9.0, 0.05, 9000, seed=80, point_extent=8.0)
    got, want = _dsm_both(sc)
    S.assert_dsm_close(got, want)


def test_dsm_clustered_cloud_overflows_lds_tiles():
    # 400 pts/m^2 in one corner: those tiles exceed the LDS point capacity and
    # fall back to the global path, the rest of the map stays on the LDS path
    sc = S.Scene(60.0, 40.0, 0.25, 20000, seed=81)
    rng = np.random.default_rng(5)
    dense = np.empty((40000, 3))
    dense[:, 0] = rng.uniform(10.0, 20.0, 40000)
    dense[:, 1] = rng.uniform(5.0, 15.0, 40000)
    dense[:, 2] = 400.0 + rng.uniform(-0.5, 0.5, 40000)
    sc.points = np.ascontiguousarray(np.concatenate([sc.points, dense]))
    got, want = _dsm_both(sc)
    S.assert_dsm_close(got, want)


def test_dsm_uneven_density_takes_every_capacity_class():
    # Real clouds are not uniform (overlapping strips, partial coverage).  The gather's
    # main launch sizes its LDS image for the MEAN density; denser tiles go to list launches
    # with more LDS per workgroup (<= 2752 and <= 5600 points per tile region), denser still
    # to the global-memory path.  Same for the sort's placement pass (sub-partitions beyond
    # 2048 points: 6144-point workgroups, beyond that direct placement).  ~1.6 M points, so
    # the three-pass sort runs: base 0.4 pts/cell + patches of 1.2, 2.5, 5 and 12 pts/cell.
    rng = np.random.default_rng(17)
    lx, ly, res = 520.0, 380.0, 0.25
    g = O.make_grid(lx, ly, res)
    n0 = int(0.4 * g.rows * g.cols)
    parts = [np.c_[rng.uniform(-lx / 2 - 2, lx / 2 + 2, n0), rng.uniform(-ly / 2 - 2, ly / 2 + 2, n0)]]
    for k, dens in enumerate([1.2, 2.5, 5.0, 12.0]):
        side = 60.0 if dens < 6 else 30.0
        cx0, cy0 = -200.0 + 110.0 * k, -100.0 + 50.0 * k
        nk = int(dens * (side / res) ** 2)
        parts.append(np.c_[rng.uniform(cx0, cx0 + side, nk), rng.uniform(cy0, cy0 + side, nk)])
    xy = np.concatenate(parts)
    pts = np.empty((xy.shape[0], 3))
    pts[:, :2] = xy
    pts[:, 2] = synth.terrain_height(xy[:, 0], xy[:, 1]) + rng.uniform(-0.3, 0.3, xy.shape[0])
    assert pts.shape[0] > (1 << 20)
    A = _A()
    rc, want, _ = O.dsm_process(pts, g)
    assert rc == O.OK
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, lx, ly, res)) as m:
        A.Dsm(A.DsmSettings(), m).process(pts, m)
        got = m.get("elevation")
        # the same cloud as a sparse call onto the materialized layer (list launches only)
        few = np.ascontiguousarray(pts[-int(12.0 * (30.0 / res) ** 2):])
        A.Dsm(A.DsmSettings(), m).process(few, m)
        got2 = m.get("elevation")
    S.assert_dsm_close(got, want)
    rc, want2, _ = O.dsm_process(few, g, elevation=want.copy())
    assert rc == O.OK
    S.assert_dsm_close(got2, want2)


def test_dsm_class_launches_skipped_after_a_uniform_call_still_serve_an_uneven_cloud():
    """Round 5's launch policy: after a call whose capacity-class lists and big-sub-partition list
    were all empty, the next call of the same geometry does not launch those kernels -- the main
    gather launch then takes every class (denser tiles walk the global bins) and the placement
    kernel places over-full sub-partitions itself.  The pinned counters that call leaves bring the
    class launches back for the one after.  All three calls against the oracle, and the skipped
    form bit for bit against the full one (FP64 mode)."""
    rng = np.random.default_rng(23)
    lx, ly, res = 520.0, 380.0, 0.25
    g = O.make_grid(lx, ly, res)
    n0 = int(0.5 * g.rows * g.cols)
    uni = np.empty((n0, 3))
    uni[:, 0] = rng.uniform(-lx / 2 - 2, lx / 2 + 2, n0)
    uni[:, 1] = rng.uniform(-ly / 2 - 2, ly / 2 + 2, n0)
    uni[:, 2] = synth.terrain_height(uni[:, 0], uni[:, 1]) + rng.uniform(-0.3, 0.3, n0)
    parts = [uni[:int(0.8 * n0), :2]]
    for k, dens in enumerate([1.5, 3.0, 6.0, 14.0]):
        side = 60.0 if dens < 6 else 30.0
        cx0, cy0 = -200.0 + 110.0 * k, -100.0 + 50.0 * k
        nk = int(dens * (side / res) ** 2)
        parts.append(np.c_[rng.uniform(cx0, cx0 + side, nk), rng.uniform(cy0, cy0 + side, nk)])
    xy = np.concatenate(parts)
    # (about as many points as the uniform cloud: the sort reuses its plan's geometry)
    uneven = np.empty((xy.shape[0], 3))
    uneven[:, :2] = xy
    uneven[:, 2] = synth.terrain_height(xy[:, 0], xy[:, 1]) + rng.uniform(-0.3, 0.3, xy.shape[0])
    assert uni.shape[0] > (1 << 20) and uneven.shape[0] > (1 << 20)
    A = _A()
    rc, want_u, _ = O.dsm_process(uni, g)
    assert rc == O.OK
    rc, want_e, _ = O.dsm_process(uneven, g)
    assert rc == O.OK
    tol = 1e-6 if _EXACT else 1e-4
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, lx, ly, res)) as m:
        m.set_dsm_precision(_EXACT)
        dsm = A.Dsm(A.DsmSettings(), m)
        for _ in range(2):                   # (the second call reads the first one's counters: all zero)
            m.reset()
            dsm.process(uni, m)
            S.assert_dsm_close(m.get("elevation"), want_u, tol=tol)
        st = m.dsm_gather_stats()
        assert st["class1"] == 0 and st["class2"] == 0 and st["beyond_lds"] == 0
        m.reset()
        dsm.process(uneven, m)               # class launches skipped: the main launch serves every tile
        skipped = m.get("elevation")
        S.assert_dsm_close(skipped, want_e, tol=tol)
        st = m.dsm_gather_stats()
        assert st["class1"] > 0 and st["class2"] > 0   # (classified all the same: the next call's policy)
        m.reset()
        dsm.process(uneven, m)               # ... and launched again
        full = m.get("elevation")
        S.assert_dsm_close(full, want_e, tol=tol)
        if _EXACT:
            assert ((skipped.view(np.uint32) == full.view(np.uint32)) | (np.isnan(skipped) & np.isnan(full))).all()


@pytest.mark.parametrize("dens", [4.0, 9.0, 30.0])
def test_dsm_dense_clouds_take_the_wave_per_cell_path(dens):
    # Dense stereo clouds put tens of points into a 0.25 m cell and hundreds of neighbours
    # into a search disc: more than any LDS image of a tile holds.  Those tiles run one WAVE
    # per cell over the global bins (lanes over the candidates, butterfly reduction).
    rng = np.random.default_rng(int(dens))
    lx, ly, res = 72.0, 44.0, 0.25
    g = O.make_grid(lx, ly, res)
    n = int(dens * g.rows * g.cols)
    pts = np.empty((n, 3))
    pts[:, 0] = rng.uniform(-lx / 2 - 1.5, lx / 2 + 1.5, n)
    pts[:, 1] = rng.uniform(-ly / 2 - 1.5, ly / 2 + 1.5, n)
    pts[:, 2] = synth.terrain_height(pts[:, 0], pts[:, 1]) + rng.uniform(-2.0, 2.0, n)
    pts = pts[~((pts[:, 0] > 5.0) & (pts[:, 0] < 12.0) & (pts[:, 1] > -3.0) & (pts[:, 1] < 4.0))]  # a hole: ladder
    A = _A()
    rc, want, _ = O.dsm_process(pts, g)
    assert rc == O.OK
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, lx, ly, res)) as m:
        A.Dsm(A.DsmSettings(), m).process(pts, m)
        got = m.get("elevation")
    assert np.isnan(want).any() and (~np.isnan(want)).mean() > 0.9
    S.assert_dsm_close(got, want)


@pytest.mark.parametrize("knobs", [
    {"sort_one_level": "1"},
    {"p3_min_points": "0"},
    {"p3_min_points": "0", "p3_target": "48"},
    # sub-partitions beyond one LDS image (contexts of > 130 M points; here forced at test size):
    # placed in rounds, a single over-full bin directly -- k_dsm_p3_place(_rec)_big / place_rounds
    {"p3_min_points": "0", "p3_target": "4000", "p3_cap": "64", "p3_rounds_cap": "96"},
    {"p3_min_points": "0", "p3_target": "4000", "p3_cap": "64", "p3_rounds_cap": "96",
     "p3_rounds_reread": "1"},
], ids=["one-level", "three-pass", "three-pass-many-blocks", "three-pass-rounds", "three-pass-rounds-reread"])
def test_dsm_every_sort_path_matches(knobs):
    # The binning sort has two implementations (one-level counting sort for
    # clouds below 2^20 points, three-pass partition sort for large clouds; round 1's
    # two-level stripe sort is gone); each is forced through its environment knob in a child
    # process, on a uniform cloud, a clustered one (over-full LDS partitions)
    # and the intensity variant (OrthoFromPcl).
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, oracle_ffi as O, scenarios as S, aerial_mapper_amd as A\n"
        "def run(sc):\n"
        "    rc, want, _ = O.dsm_process(sc.points, sc.grid)\n"
        "    g = sc.grid\n"
        "    m = A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution))\n"
        "    for exact in (True, False):\n"
        "        m.reset(); m.set_dsm_precision(exact)\n"
        "        A.Dsm(A.DsmSettings(), m).process(sc.points, m)\n"
        "        S.assert_dsm_close(m.get('elevation'), want, tol=1e-6 if exact else 1e-4)\n"
        "    inten = (np.arange(sc.points.shape[0]) %% 251).astype(np.int32)\n"
        "    rc, want_o = O.ortho_from_pcl(sc.points, inten, g)\n"
        "    A.OrthoFromPcl(A.OrthoFromPclSettings()).process(sc.points, inten, m)\n"
        "    np.testing.assert_allclose(m.get('ortho'), want_o, rtol=0, atol=1e-3)\n"
        "run(S.Scene(150.0, 110.0, 0.5, 70000, seed=82))\n"
        "sc = S.Scene(60.0, 40.0, 0.25, 20000, seed=81)\n"
        "rng = np.random.default_rng(5)\n"
        "dense = np.empty((40000, 3))\n"
        "dense[:, 0] = rng.uniform(10.0, 20.0, 40000)\n"
        "dense[:, 1] = rng.uniform(5.0, 15.0, 40000)\n"
        "dense[:, 2] = 400.0 + rng.uniform(-0.5, 0.5, 40000)\n"
        "sc.points = np.ascontiguousarray(np.concatenate([sc.points, dense]))\n"
        "run(sc)\n"
        "print('SORT_PATH_OK')\n" % (S.__file__.rsplit('/tests/', 1)[0], S.__file__.rsplit('/', 1)[0]))
    from conftest import tuning_env
    env = tuning_env(**knobs)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"SORT_PATH_OK" in r.stdout, r.stdout.decode()[-2000:]


def test_ortho_many_frames_cross_the_cull_chunk():
    # 1100 frames > the 1024-frame cull pass: the candidate list is rebuilt per
    # chunk while the per-cell fold state carries over
    cam = S.camera(48, 36, 40.0)
    sc = S.Scene(60.0, 40.0, 1.0, 9000, seed=83, num_frames=1100, altitude=470.0, cam=cam,
                 tilt_deg=8.0)
    got, want = _ortho_both(sc)
    assert _coverage(want) > 0.5
    S.assert_layers_equal(got, want, ORTHO_LAYERS)
    assert np.nanmax(want["observation_index"]) > 1024


def test_utm_scale_coordinates():
    # real surveys live at UTM magnitudes (easting ~4.6e5, northing ~5.3e6): the
    # map is centred there, the DSM's centre offsets stay 0 (like every shipped
    # flag file), the cloud and the camera poses carry the large coordinates
    c = (464980.25, 5272690.5)
    sc = S.Scene(120.0, 90.0, 0.5, 60000, seed=84, center=c, num_frames=9, altitude=480.0)
    got, want = _dsm_both(sc)
    assert (~np.isnan(want)).mean() > 0.99
    S.assert_dsm_close(got, want)
    g2, w2 = _ortho_both(sc, elevation=want)
    assert _coverage(w2) > 0.5
    S.assert_layers_equal(g2, w2, ORTHO_LAYERS)


def test_reset_restores_every_layer_whatever_wrote_it():
    # amhip_layers_reset only refills layers that may have been written since the
    # previous reset; whoever writes (kernels, uploads, a torch view) must be seen
    import torch
    A = _A()
    sc = S.Scene(60.0, 44.0, 1.0, 5000, seed=90, num_frames=4, altitude=470.0, colored=True)
    init = {"ortho": 255.0, "elevation": np.nan, "elevation_angle": 0.0, "num_observations": 0.0,
            "observation_index": np.nan, "colored_ortho": np.nan}

    def pristine(m):
        for name, v in init.items():
            a = m.get(name)
            ok = np.isnan(a).all() if np.isnan(v) else (a == v).all()
            assert ok, name

    with _map_for(sc, A) as m:
        m.reset()
        pristine(m)
        A.Dsm(A.DsmSettings(), m).process(sc.points, m)
        ncam = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
        A.OrthoBackwardGrid(ncam, A.OrthoSettings(colored_ortho=True), m).process(
            sc.poses, sc.frames, m)
        assert not np.isnan(m.get("colored_ortho")).all()
        m.reset()
        pristine(m)
        m.reset()          # nothing written in between: no fill needed, still pristine
        pristine(m)
        m.set("num_observations", np.full((m.cols, m.rows), 3.0, np.float32))
        m.reset()
        pristine(m)
        t = m.as_torch("elevation_angle")
        m.reset()
        t.fill_(7.0)       # a write the library cannot see
        torch.cuda.synchronize()
        m.reset()
        m.synchronize()
        pristine(m)


def test_dsm_running_product_is_rescaled():
    # The division-free IDW keeps prod(d^2) of a cell's neighbours; 150 points a few
    # millimetres from a cell centre drive it far below 1e-300 unless the kernel
    # rescales N, D, P together (exact powers of two) between window rows.
    sc = S.Scene(40.0, 30.0, 0.25, 4000, seed=91)
    rng = np.random.default_rng(12)
    x, y = O.cell_position(sc.grid, 57, 41)
    near = np.empty((150, 3))
    ang = rng.uniform(0, 2 * np.pi, 150)
    rad = rng.uniform(1e-3, 4e-3, 150)
    near[:, 0] = x + rad * np.cos(ang)
    near[:, 1] = y + rad * np.sin(ang)
    near[:, 2] = 400.0 + rng.uniform(-2.0, 2.0, 150)
    # and a cell whose neighbours are all far away (product grows instead: d^2 up to 1 only,
    # so the large side is exercised with radius 9: d^2 up to 9, ~700 neighbours)
    sc.points = np.ascontiguousarray(np.concatenate([sc.points, near]))
    got, want = _dsm_both(sc)
    S.assert_dsm_close(got, want)
    assert abs(float(got[41, 57]) - float(want[41, 57])) <= 1e-4
    sc9 = S.Scene(30.0, 24.0, 0.5, 18000, seed=92)
    got, want = _dsm_both(sc9, radius=9)
    S.assert_dsm_close(got, want)


@pytest.mark.parametrize("lx,ly,res,n", [(1.0, 1.0, 1.0, 1), (1.0, 1.0, 1.0, 40), (7.0, 1.0, 1.0, 30),
                                         (1.0, 9.0, 0.5, 25), (3.0, 2.0, 0.25, 3), (65.0, 17.0, 1.0, 900),
                                         (16.25, 64.25, 0.25, 5000)])
def test_dsm_degenerate_and_odd_grid_shapes(lx, ly, res, n):
    # 1x1, single row / column, sizes just past a tile edge (64 x 16 cells), one point
    sc = S.Scene(lx, ly, res, n, seed=93 + n, point_extent=max(lx, ly) / 2.0 + 1.5)
    got, want = _dsm_both(sc)
    assert got.shape == want.shape == (sc.grid.cols, sc.grid.rows)
    S.assert_dsm_close(got, want)


def test_dsm_ignores_non_finite_points():
    # NaN / inf coordinates never satisfy d2 < T in the reference's search; a NaN
    # height does reach the interpolation (and poisons that cell) in both
    sc = S.Scene(30.0, 20.0, 0.5, 3000, seed=94)
    pts = sc.points.copy()
    pts[5, 0] = np.nan
    pts[17, 1] = np.inf
    pts[29, 0] = -np.inf
    pts[41, 2] = np.nan
    sc.points = np.ascontiguousarray(pts)
    got, want = _dsm_both(sc)
    S.assert_dsm_close(got, want)
    assert np.isnan(want).sum() > 0


def _lazy_reset_scenario():
    """DSM with large holes (whole tiles without points), a clustered corner
    (over-full LDS tiles), the mosaic on top (tiles without elevation), then a
    second round after reset().  Returns every layer after each stage."""
    A = _A()
    sc = S.Scene(90.0, 70.0, 0.25, 30000, seed=95, num_frames=5, altitude=470.0)
    pts = sc.points[(sc.points[:, 0] < 5.0) | (sc.points[:, 1] > 20.0)]
    rng = np.random.default_rng(3)
    dense = np.c_[rng.uniform(-40, -32, 30000), rng.uniform(-30, -22, 30000),
                  400.0 + rng.uniform(-0.5, 0.5, 30000)]
    pts = np.ascontiguousarray(np.concatenate([pts, dense]))
    inten = (np.arange(pts.shape[0]) % 251).astype(np.int32)
    ncam = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
    names = ["ortho", "elevation", "elevation_angle", "num_observations", "observation_index",
             "colored_ortho"]
    out = []
    with _map_for(sc, A) as m:
        # (two RUNS are compared layer by layer: the FP64 gather, whose heights do not move by a
        # float spacing with the order of the points inside a bin)
        m.set_dsm_precision(True)
        for rnd in range(2):
            A.Dsm(A.DsmSettings(), m).process(pts if rnd == 0 else pts[::3], m)
            A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m).process(sc.poses, sc.frames, m)
            out.append({n: m.get(n) for n in names})
            m.reset()
            A.Dsm(A.DsmSettings(), m).process(np.zeros((0, 3)), m)      # empty cloud: no-op
            A.OrthoFromPcl(A.OrthoFromPclSettings()).process(pts[::2], inten[::2], m)
            out.append({n: m.get(n) for n in names})
            m.reset()
    return out


def test_lazy_reset_is_indistinguishable_from_eager_fills():
    # amhip_layers_reset writes nothing; the producers fuse the fill.  Every layer
    # must hold exactly what plain fills would have left (AMHIP_TUNING=eager_reset).
    import os
    import pickle
    import subprocess
    import sys
    lazy = _lazy_reset_scenario()
    code = ("import sys, pickle; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_parity as T\n"
            "sys.stdout.buffer.write(pickle.dumps(T._lazy_reset_scenario()))\n"
            % (S.__file__.rsplit('/tests/', 1)[0], S.__file__.rsplit('/', 1)[0]))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AMHIP_TUNING="eager_reset"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    eager = pickle.loads(r.stdout)
    assert len(eager) == len(lazy) == 4
    for a, b in zip(lazy, eager):
        for n in a:
            # (the DSM's double sums may differ in order between runs: compare the
            # pattern exactly and the values to the parity tolerance)
            assert np.array_equal(np.isnan(a[n]), np.isnan(b[n])), n
            np.testing.assert_allclose(np.nan_to_num(a[n]), np.nan_to_num(b[n]), rtol=0,
                                       atol=1e-4 if n in ("elevation", "ortho") else 0, err_msg=n)
    # untouched regions really hold the initial values
    first = lazy[0]
    assert np.isnan(first["elevation"]).sum() > 1000
    assert (first["ortho"][np.isnan(first["elevation"])] == 255.0).all()
    assert (first["elevation_angle"][np.isnan(first["elevation"])] == 0.0).all()
    assert np.isnan(first["colored_ortho"]).all() and (first["num_observations"] == 0).all()
    second = lazy[1]
    assert np.isnan(second["elevation"]).all()          # reset + empty cloud
    assert (second["ortho"] == 255.0).sum() > 1000 and (second["ortho"] != 255.0).sum() > 1000


@pytest.mark.parametrize("kind,dist", [
    ("radtan", (-0.28, 0.07, 2e-4, -1e-4)),      # monotone radial polynomial
    ("radtan", (-0.30, 0.0, 1e-3, -5e-4)),       # folds back at rho ~ 1.8: far cells "visible"
    ("equidistant", (-0.02, 0.004, -0.001, 0.0002)),
    ("equidistant", (-0.8, 0.0, 0.0, 0.0)),      # theta_d returns to 0 at 64 deg off axis
])
def test_ortho_distorted_cameras_are_culled_conservatively(kind, dist):
    # 30 m above ground over a 200 m map: most frames see a cell only far off
    # axis or not at all.  With a distortion model the kernel culls frames
    # against a cone derived from the distortion polynomial (incl. the regions
    # where it folds back into the image, which the reference counts as
    # visible); the result must equal the brute-force oracle bit for bit.
    A = _A()
    model = O.DIST_RADTAN if kind == "radtan" else O.DIST_EQUIDISTANT
    sc = S.Scene(200.0, 160.0, 1.0, 60000, seed=96, num_frames=30, altitude=430.0,
                 cam=S.camera(96, 54, 70.0, model, dist), tilt_deg=6.0)
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    assert rc == O.OK
    layers = O.new_layers(sc.grid)
    layers["elevation"] = elevation.copy()
    assert O.ortho_process(sc.grid, sc.cam, sc.poses, sc.T_C_B, sc.frames, layers) == O.OK
    seen = ~np.isnan(layers["observation_index"])
    assert 0.05 < seen.mean()
    with _map_for(sc, A) as m:
        m.set("elevation", elevation)
        ncam = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height,
                         model, dist)
        A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m).process(sc.poses, sc.frames, m)
        got = {n: m.get(n) for n in ORTHO_LAYERS}
    if kind == "equidistant":
        # atan comes from two libms: allow the documented handful of cells
        same = got["observation_index"].view(np.uint32) == layers["observation_index"].view(np.uint32)
        same |= np.isnan(got["observation_index"]) & np.isnan(layers["observation_index"])
        assert (~same).sum() <= 4
    else:
        S.assert_layers_equal(got, layers, ORTHO_LAYERS)


# (found by tools/soak.py -- 127, 148: wide search windows, the LDS image's cell table leaves
# less room for points than the nominal capacity classes assume; 1301: a cluster of 20 000
# points, one trip's product of squared distances dips below 2^-1022 and climbs back)
@pytest.mark.parametrize("seed", list(range(14)) + [127, 148, 1301])
def test_dsm_random_configurations(seed):
    # randomized sweep over grid shape, resolution, squared radius, density,
    # map centre / dsm centre offsets and an optional dense cluster: exercises the
    # window tables (odd and even radii in cells), both tile heights, the LDS
    # capacity heuristics, the fallback ladder and the sort paths by size
    rng = np.random.default_rng(1000 + seed)
    res = float(rng.choice([0.1, 0.2, 0.25, 0.5, 1.0, 2.0]))
    cells_x = int(rng.integers(3, 500))
    cells_y = int(rng.integers(3, 400))
    while cells_x * cells_y > 160000:
        cells_x = max(3, cells_x // 2)
    lx, ly = cells_x * res, cells_y * res
    radius = int(rng.choice([1, 1, 2, 3, 5]))
    density = float(rng.choice([0.05, 0.3, 1.0, 4.0, 12.0, 40.0]))          # points per m^2
    n = int(min(250000, max(1, density * (lx + 6) * (ly + 6))))
    center = (float(rng.uniform(-500, 500)), float(rng.uniform(-500, 500)))
    ce, cn = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))) if seed % 3 == 0 else (0.0, 0.0)
    sc = S.Scene(lx, ly, res, n, seed=2000 + seed, center=center,
                 point_extent=max(lx, ly) / 2.0 + 3.0)
    pts = sc.points
    if seed % 4 == 1:      # a dense cluster somewhere inside
        k = min(20000, n)
        c0 = np.array([center[0] + rng.uniform(-lx / 4, lx / 4), center[1] + rng.uniform(-ly / 4, ly / 4)])
        cl = np.c_[c0[0] + rng.normal(0, 2 * res, k), c0[1] + rng.normal(0, 2 * res, k),
                   400.0 + rng.uniform(-1, 1, k)]
        pts = np.concatenate([pts, cl])
    # dsm.cc:42-43 subtracts center_NORTHING from x and center_EASTING from y
    pts = pts + np.array([cn, ce, 0.0])
    sc.points = np.ascontiguousarray(pts)
    # an exact hit would abort both implementations: nudge coincident points
    got, want = _dsm_both(sc, radius=radius, ce=ce, cn=cn)
    S.assert_dsm_close(got, want)


@pytest.mark.parametrize("seed", range(10))
def test_ortho_random_configurations(seed):
    # randomized sweep over image size / focal length / principal point, flight
    # altitude and tilt (up to horizon-grazing), T_C_B, batch splitting, colour,
    # distortion model and elevation holes; every layer must match the oracle
    rng = np.random.default_rng(3000 + seed)
    A = _A()
    W, H = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    f = float(rng.uniform(0.5, 1.6) * W)
    model = [O.DIST_NONE, O.DIST_NONE, O.DIST_RADTAN][seed % 3]
    dist = (float(rng.uniform(-0.3, 0.05)), float(rng.uniform(-0.02, 0.08)),
            float(rng.uniform(-1e-3, 1e-3)), float(rng.uniform(-1e-3, 1e-3))) if model else (0, 0, 0, 0)
    cam = S.camera(W, H, f, model, dist)
    cam.cu += float(rng.uniform(-3, 3))
    cam.cv += float(rng.uniform(-3, 3))
    res = float(rng.choice([0.25, 0.5, 1.0]))
    lx, ly = float(rng.integers(20, 120)), float(rng.integers(20, 100))
    F = int(rng.integers(1, 40))
    colored = bool(seed % 2)
    tilt = float(rng.choice([2.0, 8.0, 25.0, 70.0]))
    alt = 400.0 + float(rng.uniform(15.0, 200.0))
    sc = S.Scene(lx, ly, res, int(2.0 * lx * ly), seed=4000 + seed, num_frames=F, cam=cam,
                 altitude=alt, colored=colored, tilt_deg=tilt,
                 center=(float(rng.uniform(-200, 200)), float(rng.uniform(-200, 200))))
    if seed % 4 == 2:       # a strip without points -> NaN elevation
        sc.points = np.ascontiguousarray(sc.points[sc.points[:, 0] < sc.center[0] + lx / 6])
    T_C_B = np.r_[rng.uniform(-0.2, 0.2, 3), rng.normal(size=4) * 0.02 + np.array([1.0, 0, 0, 0])]
    T_C_B[3:] /= np.linalg.norm(T_C_B[3:])
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    assert rc == O.OK
    layers = O.new_layers(sc.grid)
    layers["elevation"] = elevation.copy()
    cuts = sorted(set([0, F] + [int(v) for v in rng.integers(0, F + 1, 2)]))
    with _map_for(sc, A) as m:
        m.set("elevation", elevation)
        ncam = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, W, H, model, dist, T_C_B)
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(colored_ortho=colored), m)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            assert O.ortho_process(sc.grid, cam, sc.poses[lo:hi], T_C_B, sc.frames[lo:hi], layers,
                                   colored=colored) == O.OK
            mosaic.process(sc.poses[lo:hi], sc.frames[lo:hi], m)
        got = {n: m.get(n) for n in ORTHO_LAYERS}
    S.assert_layers_equal(got, layers, ORTHO_LAYERS)


def test_dsm_sparse_calls_on_a_large_map_use_the_tile_list():
    # > 8192 gather tiles and far fewer points than cells: the gather walks the
    # list of occupied tiles with a fixed grid (incremental mapping: one stereo
    # pair at a time onto a big map).  Two successive clouds, the second partly
    # over the first; everything else keeps its previous value.
    A = _A()
    g = O.make_grid(752.0, 752.0, 0.25)
    assert (g.rows // 64 + 1) * (g.cols // 16) > 8192
    a = synth.make_points(160000, 70.0, 97, center=(-250.0, 180.0))
    b = synth.make_points(120000, 60.0, 98, center=(-190.0, 150.0))
    rc, want, _ = O.dsm_process(a, g)
    assert rc == O.OK
    rc, want, _ = O.dsm_process(b, g, elevation=want)
    assert rc == O.OK
    st = A.GridMapSettings(0.0, 0.0, 752.0, 752.0, 0.25)
    with A.AerialGridMap(st) as m:
        m.get("elevation")                      # materialize: the very first call is sparse too
        A.Dsm(A.DsmSettings(), m).process(a, m)
        A.Dsm(A.DsmSettings(), m).process(b, m)
        got = m.get("elevation")
    S.assert_dsm_close(got, want)
    assert 0.01 < (~np.isnan(want)).mean() < 0.2


def test_ortho_coarse_cull_with_the_tracked_height_range():
    # Small batches: the mosaic kernel first asks whether ANY frame can see the
    # tile given the range of heights the DSM calls have written (tracked on the
    # device), and leaves without reading the tile's elevation if not.  Two
    # clouds with very different heights (the range must be their union), frames
    # all over the map, batches of 1..5 frames, then a reset and another round.
    A = _A()
    sc = S.Scene(300.0, 220.0, 0.5, 60000, seed=99, num_frames=18, altitude=470.0, tilt_deg=8.0)
    lo = sc.points[sc.points[:, 0] < 20.0].copy()
    hi = sc.points[sc.points[:, 0] > -20.0].copy()
    hi[:, 2] += 90.0 * np.exp(-((hi[:, 0] - 80.0) ** 2 + hi[:, 1] ** 2) / 3000.0)   # a hill
    ncam = A.NCamera(sc.cam.fu, sc.cam.fv, sc.cam.cu, sc.cam.cv, sc.cam.width, sc.cam.height)
    with _map_for(sc, A) as m:
        dsm = A.Dsm(A.DsmSettings(), m)
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        for rnd in range(2):
            layers = O.new_layers(sc.grid)
            for cloud, cuts in ((lo, [0, 1, 4, 9]), (hi, [9, 14, 15, 18])):
                dsm.process(cloud, m)
                layers["elevation"] = m.get("elevation")       # both folds see the GPU's DSM
                for a, b in zip(cuts[:-1], cuts[1:]):
                    assert O.ortho_process(sc.grid, sc.cam, sc.poses[a:b], sc.T_C_B,
                                           sc.frames[a:b], layers) == O.OK
                    mosaic.process(sc.poses[a:b], sc.frames[a:b], m)
            got = {n: m.get(n) for n in ORTHO_LAYERS}
            S.assert_layers_equal(got, layers, ORTHO_LAYERS)
            assert (~np.isnan(layers["observation_index"])).mean() > 0.3
            assert np.nanmax(layers["elevation"]) > 460.0
            m.reset()


def test_dsm_small_cloud_on_a_large_map_runs_on_its_bounding_box(tuning):
    """Round 4: a cloud of < 2^20 points onto a materialized map of >= 4 M cells is binned and
    gathered on a SUB-window around its bounding box (amhip_api.hip: dsm_subwindow), writing into
    the full layer -- the incremental demo's call per stereo pair.  Same heights as the whole-window
    call (tuning knob dsm_no_subwindow) in both modes, cells outside the box untouched, clouds across
    the map's corner and wholly beyond its border included; against the oracle on the whole map."""
    A = _A()
    rows, cols, res = 2304, 2048, 0.5                      # 4.7 M cells
    lx, ly = rows * res, cols * res
    g = O.make_grid(lx, ly, res, 100.0, -50.0)
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    rng = np.random.default_rng(77)
    centres = [(-300.0, 200.0),                                            # inside the map
               (g.pos_x + lx / 2 - 10.0, g.pos_y - ly / 2 + 12.0),         # across a corner
               (g.pos_x - lx / 2 - 40.0, 0.0)]                             # wholly outside, beyond the radius
    clouds = [np.c_[rng.uniform(cx - 40.0, cx + 40.0, 60000), rng.uniform(cy - 25.0, cy + 25.0, 60000),
                    400.0 + rng.uniform(-1.0, 1.0, 60000)] for cx, cy in centres]
    clouds[2][:, 0] = rng.uniform(g.pos_x - lx / 2 - 60.0, g.pos_x - lx / 2 - 20.0, 60000)
    base = rng.uniform(300.0, 310.0, (cols, rows)).astype(np.float32)
    want = base.copy()                                     # the oracle: earlier content where no point reaches
    for pts in clouds:
        rc, want, _ = O.dsm_process(pts, g, 1, 0.0, 0.0, elevation=want)
        assert rc == O.OK
    for exact in (True, False):
        outs = {}
        for sub in (True, False):
            if sub:
                tuning(dsm_no_subwindow=None)
            else:
                tuning(dsm_no_subwindow=1)
            with A.AerialGridMap(st) as m:
                m.set_dsm_precision(exact)
                m.set("elevation", base)                   # a materialized layer with earlier content
                dsm = A.Dsm(A.DsmSettings(), m)
                for pts in clouds:
                    dsm.process(pts, m)
                outs[sub] = m.get("elevation")
        a, b = outs[True], outs[False]
        if exact:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        else:
            assert np.abs(a.astype(np.float64) - b).max() <= 1e-4
        changed = a != base
        assert 0.001 < changed.mean() < 0.05                # two patches of 80 x 50 m on a 1152 x 1024 m map
        S.assert_dsm_close(a, want, tol=1e-6 if exact else 1e-4)
