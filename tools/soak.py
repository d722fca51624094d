#!/usr/bin/env python3
"""Soak: the randomized DSM / mosaic parity tests of tests/test_gpu_parity.py over many more
seeds than the suite runs (GPU vs oracle, bit-exact mosaic layers, DSM <= 1e-4 m).
    python tools/soak.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
t0 = time.time()
import test_gpu_ortho_from_pcl as P
for seed in range(first, first + count):
    for name, fn in (("dsm", T.test_dsm_random_configurations), ("ortho", T.test_ortho_random_configurations),
                     ("fwd", getattr(T, "test_forward_random_configurations", None)),
                     ("from_pcl", P.test_from_pcl_random_configurations)):
        if fn is None:
            continue
        # (the DSM generators in both arithmetic modes of the gather: FP64 = the library's default,
        # single precision = the opt-in mode; the other tests do not depend on it)
        for exact in ((True, False) if name == "dsm" else (False,)):
            T._EXACT = exact
            try:
                fn(seed)
            except Exception as e:  # keep going: report every failing seed
                bad.append((name + ("/exact" if exact else "/fast" if name == "dsm" else ""), seed, repr(e)[:200]))
                traceback.print_exc()
            T._EXACT = False

# ---- larger, rougher configurations than the suite's generators reach ----------------
import numpy as np
import oracle_ffi as O
import scenarios as S
import aerial_mapper_amd as A
from aerial_mapper_amd import synth


def big_dsm(seed):
    """1-3 M points (three-pass sort), uneven density mixtures, radii 1..9, any resolution."""
    rng = np.random.default_rng(7000 + seed)
    res = float(rng.choice([0.1, 0.2, 0.25, 0.3, 0.5, 1.0]))
    cx, cy = int(rng.integers(300, 1100)), int(rng.integers(300, 900))
    lx, ly = cx * res, cy * res
    radius = int(rng.choice([1, 1, 2, 4, 9]))
    g = O.make_grid(lx, ly, res, float(rng.uniform(-1e3, 1e3)), float(rng.uniform(-1e3, 1e3)))
    n0 = int(rng.uniform(0.3, 1.5) * 1.1e6)
    parts = [np.c_[rng.uniform(g.pos_x - lx / 2 - 3, g.pos_x + lx / 2 + 3, n0),
                   rng.uniform(g.pos_y - ly / 2 - 3, g.pos_y + ly / 2 + 3, n0)]]
    for _ in range(int(rng.integers(0, 4))):       # denser patches: 2x .. 40x
        w, h = rng.uniform(0.05, 0.4) * lx, rng.uniform(0.05, 0.4) * ly
        x0 = rng.uniform(g.pos_x - lx / 2, g.pos_x + lx / 2 - w)
        y0 = rng.uniform(g.pos_y - ly / 2, g.pos_y + ly / 2 - h)
        nk = int(min(1.2e6, rng.choice([2, 5, 15, 40]) * n0 / (lx * ly) * w * h))
        parts.append(np.c_[rng.uniform(x0, x0 + w, nk), rng.uniform(y0, y0 + h, nk)])
    xy = np.concatenate(parts)
    if seed % 3 == 0:                              # a strip without points: the ladder / NaN cells
        xy = xy[np.abs(xy[:, 0] - g.pos_x) > 0.03 * lx]
    pts = np.empty((xy.shape[0], 3))
    pts[:, :2] = xy
    pts[:, 2] = synth.terrain_height(xy[:, 0], xy[:, 1]) + rng.uniform(-1.0, 1.0, xy.shape[0])
    if seed % 4 == 1:                              # walls / canopy: +-25 m in a third of the map, a 30 m step
        rough = xy[:, 1] > g.pos_y + ly / 6
        pts[rough, 2] += rng.uniform(-25.0, 25.0, int(rough.sum()))
        pts[xy[:, 0] > g.pos_x + lx / 4, 2] += 30.0
    rc, want, _ = O.dsm_process(pts, g, radius)
    assert rc == O.OK
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, lx, ly, res)) as m:
        for exact in (True, False):
            m.reset()
            m.set_dsm_precision(exact)
            for rep in range(2 if not exact else 1):    # (fast: the second call may take the dense FP64 launch)
                m.reset()
                A.Dsm(A.DsmSettings(radius), m).process(pts, m)
                got = m.get("elevation")
                # (FP64 mode: one cell in ~1e8 sits on a float rounding boundary and comes out one
                # spacing away -- the sums run in another order; assert_dsm_close allows two per map)
                same = S.assert_dsm_close(got, want, tol=1e-6 if exact else 1e-4)
                assert not exact or same > 0.99999, same


def big_ortho(seed):
    """Maps with whole tiles inside the frames (pruning), any camera model, rough terrain."""
    rng = np.random.default_rng(8000 + seed)
    res = float(rng.choice([0.25, 0.5, 1.0]))
    cx, cy = int(rng.integers(200, 700)), int(rng.integers(150, 600))
    lx, ly = cx * res, cy * res
    center = (float(rng.uniform(-5e5, 5e5)), float(rng.uniform(-5e6, 5e6))) if seed % 2 else (0.0, 0.0)
    g = O.make_grid(lx, ly, res, center[0], center[1])
    W, H = int(rng.integers(60, 260)), int(rng.integers(40, 160))
    f = float(rng.uniform(0.6, 1.8) * W)
    model = [O.DIST_NONE, O.DIST_RADTAN, O.DIST_EQUIDISTANT][seed % 3]
    if model == O.DIST_RADTAN:
        dist = (float(rng.uniform(-0.35, 0.15)), float(rng.uniform(-0.05, 0.1)),
                float(rng.uniform(-1e-3, 1e-3)), float(rng.uniform(-1e-3, 1e-3)))
    elif model == O.DIST_EQUIDISTANT:
        dist = tuple(float(v) for v in rng.uniform(-0.03, 0.03, 4))
    else:
        dist = (0, 0, 0, 0)
    cam = S.camera(W, H, f, model, dist)
    cam.fv = cam.fu * float(rng.uniform(0.9, 1.1))
    cam.cu += float(rng.uniform(-0.1, 0.1) * W)
    cam.cv += float(rng.uniform(-0.1, 0.1) * H)
    F = int(rng.integers(2, 70))
    alt = 400.0 + float(rng.uniform(60.0, 500.0))
    poses = synth.make_lawnmower_poses(F, 0.5 * max(lx, ly), alt, seed + 31, tilt_deg=float(rng.uniform(0, 35)),
                                       center=center)
    frames = [np.ascontiguousarray(x) for x in synth.make_frames(F, H, W, 1, salt=seed % 7)]
    j, i = np.meshgrid(np.arange(g.cols), np.arange(g.rows), indexing="ij")
    elev = (400.0 + 6.0 * np.sin(0.05 * i) * np.cos(0.04 * j) + rng.uniform(-3, 3, i.shape)).astype(np.float32)
    elev[rng.uniform(size=elev.shape) < 0.01] = np.nan
    layers = O.new_layers(g)
    layers["elevation"] = elev.copy()
    cuts = sorted(set([0, F] + [int(v) for v in rng.integers(0, F + 1, 2)]))
    T_C_B = synth.IDENTITY_POSE
    names = ["elevation_angle", "observation_index", "num_observations", "ortho", "colored_ortho"]
    with A.AerialGridMap(A.GridMapSettings(center[0], center[1], lx, ly, res)) as m:
        m.set("elevation", elev)
        ncam = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, W, H, model, dist, T_C_B)
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            assert O.ortho_process(g, cam, poses[lo:hi], T_C_B, frames[lo:hi], layers) == O.OK
            mosaic.process(poses[lo:hi], frames[lo:hi], m)
        got = {n: m.get(n) for n in names}
    if model == O.DIST_EQUIDISTANT:
        # atan comes from two libms: a handful of last-bit flips allowed, never many
        for n in names:
            a, b = got[n], layers[n]
            off = int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())
            assert off <= 4, (n, off)
    else:
        S.assert_layers_equal(got, layers, names)


nbig = max(1, count // 4)
for seed in range(first, first + nbig):
    for name, fn in (("big_dsm", big_dsm), ("big_ortho", big_ortho)):
        try:
            fn(seed)
        except Exception as e:
            bad.append((name, seed, repr(e)[:300]))
            traceback.print_exc()
print("soak: seeds %d..%d (+ %d large configurations each), %d failures, %.0f s" % (
    first, first + count - 1, nbig, len(bad), time.time() - t0))
for b in bad:
    print("FAILED", b)
sys.exit(1 if bad else 0)
