#!/usr/bin/env python3
"""Look at single seeds of tools/soak.py's big_dsm: per gather mode the number of cells that differ
from the oracle, the largest difference in float spacings, and whether a second run of the same
call gives the same bits.   python tools/soak_probe.py 40032 40229"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi as O
import aerial_mapper_amd as A
from aerial_mapper_amd import synth


def scene(seed):
    # (the generator of tools/soak.py big_dsm, kept in step with it)
    rng = np.random.default_rng(7000 + seed)
    res = float(rng.choice([0.1, 0.2, 0.25, 0.3, 0.5, 1.0]))
    cx, cy = int(rng.integers(300, 1100)), int(rng.integers(300, 900))
    lx, ly = cx * res, cy * res
    radius = int(rng.choice([1, 1, 2, 4, 9]))
    g = O.make_grid(lx, ly, res, float(rng.uniform(-1e3, 1e3)), float(rng.uniform(-1e3, 1e3)))
    n0 = int(rng.uniform(0.3, 1.5) * 1.1e6)
    parts = [np.c_[rng.uniform(g.pos_x - lx / 2 - 3, g.pos_x + lx / 2 + 3, n0),
                   rng.uniform(g.pos_y - ly / 2 - 3, g.pos_y + ly / 2 + 3, n0)]]
    for _ in range(int(rng.integers(0, 4))):
        w, h = rng.uniform(0.05, 0.4) * lx, rng.uniform(0.05, 0.4) * ly
        x0 = rng.uniform(g.pos_x - lx / 2, g.pos_x + lx / 2 - w)
        y0 = rng.uniform(g.pos_y - ly / 2, g.pos_y + ly / 2 - h)
        nk = int(min(1.2e6, rng.choice([2, 5, 15, 40]) * n0 / (lx * ly) * w * h))
        parts.append(np.c_[rng.uniform(x0, x0 + w, nk), rng.uniform(y0, y0 + h, nk)])
    xy = np.concatenate(parts)
    if seed % 3 == 0:
        xy = xy[np.abs(xy[:, 0] - g.pos_x) > 0.03 * lx]
    pts = np.empty((xy.shape[0], 3))
    pts[:, :2] = xy
    pts[:, 2] = synth.terrain_height(xy[:, 0], xy[:, 1]) + rng.uniform(-1.0, 1.0, xy.shape[0])
    if seed % 4 == 1:
        rough = xy[:, 1] > g.pos_y + ly / 6
        pts[rough, 2] += rng.uniform(-25.0, 25.0, int(rough.sum()))
        pts[xy[:, 0] > g.pos_x + lx / 4, 2] += 30.0
    return g, lx, ly, res, radius, pts


for seed in [int(a) for a in sys.argv[1:]]:
    g, lx, ly, res, radius, pts = scene(seed)
    rc, want, _ = O.dsm_process(pts, g, radius)
    assert rc == O.OK
    with A.AerialGridMap(A.GridMapSettings(g.pos_x, g.pos_y, lx, ly, res)) as m:
        for exact in (True, False):
            m.set_dsm_precision(exact)
            runs = []
            for rep in range(2):
                m.reset()
                A.Dsm(A.DsmSettings(radius), m).process(pts, m)
                runs.append(m.get("elevation"))
            got = runs[0]
            nan_equal = bool(np.array_equal(np.isnan(got), np.isnan(want)))
            ok = ~np.isnan(want)
            d = np.abs(got[ok].astype(np.float64) - want[ok].astype(np.float64))
            sp = np.spacing(np.abs(want[ok]).astype(np.float32)).astype(np.float64)
            diff = d > 0
            print({"seed": seed, "mode": "FP64" if exact else "single precision", "cells": int(ok.sum()),
                   "points": int(pts.shape[0]), "res": res, "radius": radius, "nan_pattern_equal": nan_equal,
                   "cells_differing": int(diff.sum()), "max_diff_in_float_spacings": float((d / sp).max()),
                   "max_diff_m": float(d.max()),
                   "second_run_bits_equal": bool(np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32)))})
