"""GPU tests of the DSM gather's single-precision mode (amhip_ctx_set_dsm_precision,
AMHIP_DSM_FAST, opt-in since round 3): its guards -- error budget per tile, near-centre points, decisions
within 2e-6 of the search radius -- must hand exactly the right work to the FP64 routines, so
that the contract holds everywhere: the reference's NaN pattern, heights within 1e-4 m (one
float spacing of the stored height where that is larger, i.e. above 1024 m).
Oracle: dsm.cc:113-184 restated (oracle/amo_dsm.cc), checked against the reference's own code
in tests/test_reference_loops.py."""
import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S

pytestmark = pytest.mark.gpu


def _run(scene, exact, radius=1):
    import aerial_mapper_amd as A
    g = scene.grid
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    with A.AerialGridMap(st) as m:
        m.set_dsm_precision(exact)
        A.Dsm(A.DsmSettings(radius), m).process(scene.points, m)
        return m.get("elevation")


def _oracle(scene, radius=1):
    rc, want, _ = O.dsm_process(scene.points, scene.grid, radius, 0.0, 0.0)
    assert rc == O.OK
    return want


def _spacing(z):
    """spacing of the float32 grid at |z|"""
    return np.spacing(np.abs(z).astype(np.float32)).astype(np.float64)


def _check(got, want, lsb=False):
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "NaN pattern differs in %d cells" % int((gn != wn).sum())
    ok = ~wn
    err = np.abs(got[ok].astype(np.float64) - want[ok].astype(np.float64))
    tol = np.maximum(1e-4, _spacing(want[ok])) if lsb else 1e-4
    assert (err <= tol).all(), "max |dh| = %g m" % err.max()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (gn & wn)
    return float(same.mean()), float(err.max()) if ok.any() else 0.0


def test_rough_heights_send_every_tile_to_the_fp64_kernel():
    # +-30 m of height noise: no tile has room under the f32 error bound
    sc = S.Scene(100.0, 80.0, 0.25, int(8 * 108 * 88), seed=301)
    rng = np.random.default_rng(5)
    sc.points[:, 2] += rng.uniform(-30.0, 30.0, sc.points.shape[0])
    want = _oracle(sc)
    fast, exact = _run(sc, False), _run(sc, True)
    frac_f, _ = _check(fast, want)
    frac_e, _ = _check(exact, want)
    assert frac_f > 0.999 and frac_e > 0.999    # both are the FP64 arithmetic


def test_a_step_in_the_terrain_mixes_both_kernels():
    # flat ground with a 25 m "building" in the middle: the tiles along its walls go to the
    # FP64 kernel, the rest stay in single precision
    sc = S.Scene(120.0, 100.0, 0.25, int(8 * 128 * 108), seed=302)
    x, y = sc.points[:, 0], sc.points[:, 1]
    sc.points[:, 2] += np.where((np.abs(x) < 20.0) & (np.abs(y) < 15.0), 25.0, 0.0)
    want = _oracle(sc)
    frac, err = _check(_run(sc, False), want)
    assert 0.3 < frac < 1.0


def test_slopes_and_canopy_noise_stay_inside_the_budget():
    # 30 % slope plus +-1.5 m of noise: height range of a tile region ~8 m, near the budget
    sc = S.Scene(100.0, 90.0, 0.25, int(8 * 108 * 98), seed=303)
    rng = np.random.default_rng(6)
    sc.points[:, 2] += 0.3 * sc.points[:, 0] + rng.uniform(-1.5, 1.5, sc.points.shape[0])
    want = _oracle(sc)
    _check(_run(sc, False), want)


def test_above_1024_m_the_bar_is_one_float_spacing():
    sc = S.Scene(80.0, 70.0, 0.25, int(8 * 88 * 78), seed=304)
    sc.points[:, 2] += 2600.0     # ~3000 m: one float spacing = 2.4e-4 m > 1e-4 m
    want = _oracle(sc)
    frac, err = _check(_run(sc, False), want, lsb=True)
    assert frac > 0.9             # the budget there is a quarter of a spacing
    frac_e, err_e = _check(_run(sc, True), want, lsb=True)
    assert frac_e > 0.999


def test_points_next_to_cell_centres():
    # returns a hair away from cell centres with heights metres apart (ground + canopy): the
    # weights 1/d2 are huge and differ by orders of magnitude
    sc = S.Scene(60.0, 50.0, 0.25, int(8 * 68 * 58), seed=305)
    g = sc.grid
    rng = np.random.default_rng(7)
    extra = []
    for k in range(400):
        i, j = int(rng.integers(5, g.rows - 5)), int(rng.integers(5, g.cols - 5))
        cx, cy = O.cell_position(g, i, j)
        for d, dz in ((10.0 ** -rng.uniform(1.5, 9.0), 0.0), (10.0 ** -rng.uniform(1.5, 9.0), 3.0)):
            a = rng.uniform(0, 2 * np.pi)
            extra.append((cx + d * np.cos(a), cy + d * np.sin(a), 400.0 + dz))
    sc.points = np.ascontiguousarray(np.vstack([sc.points, np.array(extra)]))
    want = _oracle(sc)
    _check(_run(sc, False), want)


def test_decisions_at_the_search_radius_are_the_references():
    # points at sqrt(T) (1 +- k 1e-9 .. 1e-6) from cell centres, 1000 m above the rest: one wrong
    # inclusion moves the cell by metres
    sc = S.Scene(60.0, 50.0, 0.25, int(8 * 68 * 58), seed=306)
    g = sc.grid
    rng = np.random.default_rng(8)
    extra = []
    for k in range(600):
        i, j = int(rng.integers(8, g.rows - 8)), int(rng.integers(8, g.cols - 8))
        cx, cy = O.cell_position(g, i, j)
        rel = rng.choice([0.0, 1e-15, 1e-12, 1e-9, 1e-8, 1e-7, 5e-7, 1e-6, 3e-6]) * rng.choice([-1.0, 1.0])
        d = 1.0 * (1.0 + rel)
        if k % 3:
            ca, sa = np.cos(rng.uniform(0, 2 * np.pi)), 0.0
            sa = np.sqrt(1.0 - ca * ca) * rng.choice([-1.0, 1.0])
        else:  # 3-4-5 directions: dx*dx + dy*dy lands on (or one ulp off) the radius itself
            ca, sa = rng.choice([0.6, -0.6]), rng.choice([0.8, -0.8])
            if k % 2:
                ca, sa = sa, ca
        extra.append((cx + d * ca, cy + d * sa, 1400.0))
    sc.points = np.ascontiguousarray(np.vstack([sc.points, np.array(extra)]))
    want = _oracle(sc)
    # (the outliers push their tiles over the height budget: those go to the FP64 kernel,
    # the decisions of the others are taken by the per-cell guard)
    _check(_run(sc, False), want, lsb=True)
    sc.points[-len(extra):, 2] = 404.0     # now inside the budget: the f32 kernel keeps the tiles
    want = _oracle(sc)
    _check(_run(sc, False), want)


@pytest.mark.parametrize("res,radius", [(0.3, 1), (0.1, 1), (0.7, 3)])
def test_utm_magnitudes_and_non_dyadic_resolutions(res, radius):
    ce, cn = 464980.3, 5272690.7
    sc = S.Scene(90.0 * res / 0.3, 70.0 * res / 0.3, res, 60000, seed=307, center=(ce, cn),
                 point_extent=50.0 * res / 0.3)
    want = _oracle(sc, radius)
    _check(_run(sc, False, radius), want)
    _check(_run(sc, True, radius), want)


# ---- the denser capacity classes (list launches of the single-precision kernel) and the
# ---- wave-per-block kernel in single precision (block_wave_f32) ------------------------------
def _dense_scene(pts_per_cell, seed, lx=60.0, ly=48.0, res=0.25):
    n = int(pts_per_cell * (lx / res + 24) * (ly / res + 24))
    return S.Scene(lx, ly, res, n, seed=seed)


@pytest.mark.parametrize("ppc", [2.0, 5.0, 12.0, 24.0])
def test_dense_clouds_in_single_precision(ppc):
    # 2 / 5 points per cell: classes 1 / 2 (f32 list launches); 12 / 24: the wave-per-block kernel
    sc = _dense_scene(ppc, 311)
    want = _oracle(sc)
    frac, err = _check(_run(sc, False), want)
    assert frac < 1.0 or err == 0.0          # (single precision did run: not all floats identical)
    _check(_run(sc, True), want)


@pytest.mark.parametrize("ppc", [5.0, 12.0])
def test_dense_clouds_with_rough_heights_fall_back_to_fp64(ppc):
    # +-30 m of noise: the class launches hand every tile back (lists 5 / 6), the wave-per-block
    # kernel redoes every block with block_wave()
    sc = _dense_scene(ppc, 312)
    rng = np.random.default_rng(7)
    sc.points[:, 2] += rng.uniform(-30.0, 30.0, sc.points.shape[0])
    want = _oracle(sc)
    frac, _ = _check(_run(sc, False), want)
    assert frac > 0.999


@pytest.mark.parametrize("ppc", [5.0, 12.0])
def test_dense_clouds_guards_near_centres_and_near_the_radius(ppc):
    # points 1e-9 .. 1e-2 m from cell centres with heights metres off, points at the search
    # radius (1 +- 1e-15 .. 3e-6) from centres: the cells concerned take the FP64 routines
    sc = _dense_scene(ppc, 313)
    g = sc.grid
    rng = np.random.default_rng(8)
    extra = []
    for k in range(400):
        i, j = int(rng.integers(8, g.rows - 8)), int(rng.integers(8, g.cols - 8))
        cx, cy = O.cell_position(g, i, j)
        if k % 2 == 0:
            d = 10.0 ** rng.uniform(-9, -2)
            a = rng.uniform(0, 2 * np.pi)
            extra.append((cx + d * np.cos(a), cy + d * np.sin(a), 400.0 + rng.uniform(-5, 5)))
        else:
            r = 1.0 * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-15, -5.5))
            a = rng.uniform(0, 2 * np.pi)
            extra.append((cx + r * np.cos(a), cy + r * np.sin(a), 400.0 + rng.uniform(-5, 5)))
    sc.points = np.concatenate([sc.points, np.asarray(extra)], 0)
    want = _oracle(sc)
    _check(_run(sc, False), want)


def _run_stats(scene, exact, radius=1):
    import aerial_mapper_amd as A
    g = scene.grid
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    with A.AerialGridMap(st) as m:
        m.set_dsm_precision(exact)
        A.Dsm(A.DsmSettings(radius), m).process(scene.points, m)
        return m.get("elevation"), m.dsm_gather_stats()


def test_rough_tiles_are_sorted_onto_the_fp64_lists_before_they_are_staged(tuning):
    """Round 3: after the three-pass sort the placement pass leaves every bin's height range, and
    the occupancy pre-pass applies the gather's own error bound to a tile's region BEFORE anything
    is staged -- tiles without room go straight to the FP64 lists (amhip_ctx_dsm_gather_stats
    counts them) and the result is what the kernel's own late rejection gave (one-level sort:
    no bin ranges)."""
    sc = S.Scene(160.0, 120.0, 0.25, int(8 * 168 * 128), seed=311)
    x, y = sc.points[:, 0], sc.points[:, 1]
    sc.points[:, 2] += np.where((np.abs(x) < 30.0) & (np.abs(y) < 20.0), 25.0, 0.0)   # a "building"
    want = _oracle(sc)
    tuning(p3_min_points=1000)       # three-pass sort for this small cloud
    got3, st3 = _run_stats(sc, False)
    tuning(p3_min_points=None)               # (< 2^20 points: the one-level sort)
    got1, st1 = _run_stats(sc, False)
    for got in (got3, got1):
        frac, _ = _check(got, want)
        assert 0.3 < frac < 1.0
    assert st3["tiles"] == st1["tiles"] > 0
    # the same tiles end on the FP64 lists either way: pre-classified, or handed back by the kernel
    assert st3["f32_to_fp64"] + st3["f32_to_fp64_beyond"] == st1["f32_to_fp64"] + st1["f32_to_fp64_beyond"]
    assert 0 < st3["f32_to_fp64"] < 0.5 * st3["tiles"]
    # smooth terrain: nothing is rejected, by either route
    sm = S.Scene(160.0, 120.0, 0.25, int(8 * 168 * 128), seed=312)
    tuning(p3_min_points=1000)
    got, st = _run_stats(sm, False)
    _check(got, _oracle(sm))
    assert st["f32_to_fp64"] == 0 and st["f32_to_fp64_beyond"] == 0


@pytest.mark.parametrize("density", [2.5, 6.0])
def test_rough_dense_tiles_of_the_capacity_classes_are_pre_classified_too(tuning, density):
    """Denser clouds (capacity classes 1 / 2 and the wide main launch) with +-30 m of noise: every
    occupied tile is rejected up front, onto list 5 / 6 where the FP64 image of the main launch
    does not fit a CU."""
    n = int(density * (4 * 88) * (4 * 68))
    sc = S.Scene(80.0, 60.0, 0.25, n, seed=313)
    rng = np.random.default_rng(9)
    sc.points[:, 2] += rng.uniform(-30.0, 30.0, sc.points.shape[0])
    want = _oracle(sc)
    tuning(p3_min_points=1000)
    got, st = _run_stats(sc, False)
    frac, _ = _check(got, want)
    assert frac > 0.999                                   # all of it is the FP64 arithmetic
    assert st["f32_to_fp64"] + st["f32_to_fp64_beyond"] + st["beyond_lds"] > 0.8 * st["tiles"]


def test_rough_scene_second_call_takes_the_dense_fp64_launch(tuning):
    """When the previous call pre-classified more than a tenth of its tiles, the FP64 kernel is
    launched densely over them (filtering on the occupancy byte) instead of walking list 4: same
    heights, same counts."""
    import aerial_mapper_amd as A
    sc = S.Scene(160.0, 120.0, 0.25, int(8 * 168 * 128), seed=314)
    rng = np.random.default_rng(11)
    rough = sc.points[:, 0] < 10.0                                   # two thirds of the map
    sc.points[rough, 2] += rng.uniform(-30.0, 30.0, int(rough.sum()))
    want = _oracle(sc)
    tuning(p3_min_points=1000)
    # (this test is about the two launch forms of the single-precision pipeline: keep the context
    # from leaving that pipeline altogether, test_rough_scene_switches_to_the_fp64_pipeline)
    tuning(dsm_no_rough_switch=1)
    g = sc.grid
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)) as m:
        m.set_dsm_precision(False)
        dsm = A.Dsm(A.DsmSettings(1), m)
        outs, stats = [], []
        for _ in range(3):
            m.reset()
            dsm.process(sc.points, m)
            outs.append(m.get("elevation"))
            stats.append(m.dsm_gather_stats())
    for got in outs:
        _check(got, want)
    assert 0.4 * stats[0]["tiles"] < stats[0]["f32_to_fp64"] < stats[0]["tiles"]
    assert stats[1]["f32_to_fp64"] == stats[0]["f32_to_fp64"] == stats[2]["f32_to_fp64"]
    # the rough two thirds are the FP64 arithmetic in either launch form: the reference's floats
    # (cells well inside the rough part: their whole tile region is rough)
    inner = np.zeros_like(want, bool)
    i_lo = int((g.length_x / 2.0 - (10.0 - 20.0)) / g.resolution) + 1     # x < -10 m  <=>  i > i_lo
    inner[:, i_lo:] = True
    for got in outs:
        same = got.view(np.uint32)[inner] == want.view(np.uint32)[inner]
        assert same.mean() > 0.999


# ---- the 20-byte sort records of the single-precision mode (amhip_sort.hip: make_record) ----------
class _Strip(object):
    """a long narrow map with its own cloud (scenarios.Scene spreads points over a square)"""

    def __init__(self, lx, ly, res, ppc, seed, z=lambda x, y, rng: 400.0 + 0.0 * x):
        self.grid = O.make_grid(lx, ly, res, 0.0, 0.0)
        rng = np.random.default_rng(seed)
        n = int(ppc * (lx + 8.0) * (ly + 8.0) / (res * res))
        x = rng.uniform(-lx / 2 - 4.0, lx / 2 + 4.0, n)
        y = rng.uniform(-ly / 2 - 4.0, ly / 2 + 4.0, n)
        self.points = np.ascontiguousarray(np.c_[x, y, z(x, y, rng)])


def test_relief_of_hundreds_of_metres_across_one_map():
    # the records carry heights as f32 offsets from the middle of the CLOUD's height range: far
    # from it the offsets' own rounding takes the budget and those tiles go to the FP64 kernel,
    # near it the single-precision kernel runs -- the contract holds across the map
    sc = _Strip(1600.0, 40.0, 0.25, 1.0, 311,
                z=lambda x, y, rng: 400.0 + 0.5 * x + rng.uniform(-0.2, 0.2, x.shape[0]))
    assert sc.points.shape[0] >= 1 << 20          # the three-pass sort
    want = _oracle(sc)
    frac, err = _check(_run(sc, False), want)
    frac_e, _ = _check(_run(sc, True), want)
    assert frac_e > 0.999 and 0.3 < frac < 1.0    # a mix of both kernels


def test_negative_heights_and_a_cloud_far_below_zero():
    sc = _Strip(300.0, 200.0, 0.25, 1.0, 312,
                z=lambda x, y, rng: -412.3 + 0.05 * y + rng.uniform(-0.3, 0.3, x.shape[0]))
    want = _oracle(sc)
    frac, err = _check(_run(sc, False), want)
    assert frac > 0.9


def test_maps_longer_than_the_records_cell_field_fall_back_to_fp64():
    # cells are 16-bit fields of the record: beyond 65 535 (incl. the margin) the single-precision
    # mode is not taken at all -- same bits as the FP64 mode
    sc = _Strip(17000.0, 6.0, 0.25, 0.5, 313)
    assert sc.grid.rows > 65535 or sc.grid.cols > 65535
    fast, exact = _run(sc, False), _run(sc, True)
    assert np.array_equal(fast.view(np.uint32), exact.view(np.uint32))
    _check(fast, _oracle(sc))


def test_rough_scene_switches_to_the_fp64_pipeline(tuning):
    """ the opt-in mode protects itself: when a call filed more than half of
    its tiles for the FP64 kernel, the following calls on the context run the FP64 pipeline
    outright (sorted doubles; the records' FP64 redo would stage from the unsorted cloud), and the
    single-precision pipeline is tried again after 15 calls.  A smooth scene never switches."""
    import aerial_mapper_amd as A
    tuning(p3_min_points=1000)
    tuning(dsm_no_rough_switch=None)
    sc = S.Scene(160.0, 120.0, 0.25, int(8 * 168 * 128), seed=315)
    rng = np.random.default_rng(12)
    rough_pts = sc.points.copy()
    rough_pts[:, 2] += rng.uniform(-30.0, 30.0, rough_pts.shape[0])          # every tile is rough
    rc, want_rough, _ = O.dsm_process(rough_pts, sc.grid, 1, 0.0, 0.0)
    assert rc == O.OK
    want_smooth = _oracle(sc)
    g = sc.grid
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)) as m:
        m.set_dsm_precision(False)
        dsm = A.Dsm(A.DsmSettings(1), m)
        fp64_tiles, fracs = [], []
        for k in range(20):
            m.reset()
            dsm.process(rough_pts, m)          # (sync=True: the tile counters are back before the next call)
            st = m.dsm_gather_stats()
            fp64_tiles.append(st["f32_to_fp64"] + st["f32_to_fp64_beyond"])
            frac, _ = _check(m.get("elevation"), want_rough)
            fracs.append(frac)
        # call 0 is the single-precision pipeline (it finds the scene rough), calls 1 .. 16 the FP64
        # pipeline (no tile is "sent" anywhere: the counters are zero), call 17 tries again
        assert fp64_tiles[0] > 0.5 * st["tiles"]
        assert all(v == 0 for v in fp64_tiles[1:17]), fp64_tiles
        assert fp64_tiles[17] == fp64_tiles[0] and fp64_tiles[18] == 0
        assert min(fracs[1:17]) >= 0.9999                # the FP64 arithmetic: the reference's floats
        # a smooth scene afterwards: the hold runs out, then the single-precision pipeline stays
        for k in range(20):
            m.reset()
            dsm.process(sc.points, m)
        st = m.dsm_gather_stats()
        frac, _ = _check(m.get("elevation"), want_smooth)
        assert st["f32_to_fp64"] == 0 and frac < 0.99999   # (f32 sums: a float spacing here and there)
