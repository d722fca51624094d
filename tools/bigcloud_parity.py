#!/usr/bin/env python3
"""Oracle-checked run of the big-cloud paths at FULL size: the sort's placement
in rounds (sub-partitions beyond one LDS image, registers and re-read forms -- contexts beyond
~130 M points) and clouds beyond 2^27 points, against the CPU oracle (the restated loops over the reference's
vendored nanoflann, oracle/_ref/liboracle_ref.so) on 600 x 600-cell
windows -- a corner, the middle, the far corner -- in both gather modes.  dsm.cc:36-52 has no size
regime; neither may the drop-in.

  python tools/bigcloud_parity.py [--points 140000000 --side 17000] [--points 400000000 --side 40000]
prints one JSON object per size (kept under profiles/).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def window_reference(O, pts_dev, res, L, i0, j0, s, which):
    """the reference's DSM of cells [i0, i0 + s) x [j0, j0 + s) of the L x L map centred on (0, 0):
    a grid of its own whose cell centres are the big map's (everything dyadic: exact)"""
    sub_len = s * res
    # x_i = (0 + L/2 - res/2) - res * i  ->  the sub-grid's centre
    cx = L / 2.0 - (i0 + s / 2.0) * res
    cy = L / 2.0 - (j0 + s / 2.0) * res
    g = O.make_grid(sub_len, sub_len, res, cx, cy, which="port")
    assert (g.rows, g.cols) == (s, s)
    halo = 3.0
    x, y = pts_dev[:, 0], pts_dev[:, 1]
    keep = (x > cx - sub_len / 2 - halo) & (x < cx + sub_len / 2 + halo) & \
           (y > cy - sub_len / 2 - halo) & (y < cy + sub_len / 2 + halo)
    sub = pts_dev[keep].cpu().numpy()
    rc, elev, _ = O.dsm_process(sub, g, 1, 0.0, 0.0, which=which)
    assert rc == 0
    return elev, int(sub.shape[0])


def run(n, side, res=0.25, s=600):
    import numpy as np
    import torch
    import aerial_mapper_amd as A
    import oracle_ffi as O
    from aerial_mapper_amd import synth
    dev = torch.device("cuda", 0)
    L = side * res
    which = "ref" if O.have_ref() else "port"   # (the oracle: restated loops over the vendored nanoflann)
    pts = synth.make_points_torch(n, L / 2.0 + 4.0, 45, dev)
    wins = [(0, 0), ((side - s) // 2, (side - s) // 2), (side - s, side - s), (0, side - s)]
    refs = [window_reference(O, pts, res, L, i0, j0, s, which) for i0, j0 in wins]
    out = {"points": n, "cells": side * side, "map": "%d x %d @ %.2f m" % (side, side, res),
           "oracle": {"ref": "vendored nanoflann under restated loops", "port": "restated loops"}[which],
           "windows": ["cells [%d, %d) x [%d, %d), %d points incl. 3 m halo" % (i0, i0 + s, j0, j0 + s, r[1])
                       for (i0, j0), r in zip(wins, refs)], "modes": {}}
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        dsm = A.Dsm(A.DsmSettings(), m)
        for mode in ("exact", "fast"):
            m.set_dsm_precision(mode == "exact")
            m.reset()
            dsm.process(pts, m)              # (warm-up)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.reset()
            dsm.process(pts, m)
            dt = time.perf_counter() - t0
            st = m.dsm_stats()
            elev = m.as_torch("elevation")
            e = {"ms_per_call": round(dt * 1e3, 2), "points_binned": st["points_binned"], "windows": []}
            for (i0, j0), (want, _) in zip(wins, refs):
                got = elev[j0:j0 + s, i0:i0 + s].cpu().numpy()
                ok = ~np.isnan(want)
                same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & ~ok)
                e["windows"].append({
                    "nan_pattern_equal": bool(np.array_equal(np.isnan(got), ~ok)),
                    "max_abs_err_m": float(np.abs(got[ok].astype(np.float64) - want[ok]).max()) if ok.any() else 0.0,
                    "bit_identical_frac": round(float(same.mean()), 9)})
            tol = 1e-6 if mode == "exact" else 1e-4
            e["pass"] = all(w["nan_pattern_equal"] and w["max_abs_err_m"] <= tol for w in e["windows"])
            out["modes"][mode] = e
            torch.cuda.synchronize()
    out["pass"] = all(v["pass"] for v in out["modes"].values())
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, action="append")
    ap.add_argument("--side", type=int, action="append")
    a = ap.parse_args()
    sizes = list(zip(a.points or [140_000_000, 400_000_000], a.side or [17000, 40000]))
    ok = True
    for n, side in sizes:
        r = run(n, side)
        print(json.dumps(r))
        sys.stdout.flush()
        ok = ok and r["pass"]
    sys.exit(0 if ok else 1)
