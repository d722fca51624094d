#!/usr/bin/env python3
"""Probe: DSM + mosaic time across resolutions / search radii (bin size = first radius in
cells: 1 at 1 m, 2 at 0.5 m, 4 at 0.25 m, 8+ at 0.1 m), 8 points per m^2, 40 M cells each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
for res, radius in ((1.0, 1), (0.5, 1), (0.25, 1), (0.125, 1), (0.1, 1), (0.5, 4), (1.0, 9)):
    side = 6400
    L = side * res
    n = min(int(8.0 * L * L), 120_000_000)
    m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
    dsm = A.Dsm(A.DsmSettings(interpolation_radius=radius), m)
    pts = synth.make_points_torch(n, L / 2 + 4, 43, dev)
    try:
        for _ in range(2):
            m.reset(); dsm.process(pts, m)
        m.enable_timing(True); m.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m.reset(); dsm.process(pts, m, sync=False)
        m.synchronize(); dt = (time.perf_counter() - t0) / 3
        kt = {k: round(v[0] / 3, 2) for k, v in m.kernel_times().items() if v[1]}
        nan = float(torch.isnan(m.as_torch("elevation")).float().mean())
        print("res %.3f R^2=%d  %4.1f pts/cell  %9d pts  %7.2f ms  %7.1f Mcells/s  nan %.4f  %s" % (
            res, radius, n / side / side, n, dt * 1e3, side * side / dt / 1e6, nan, kt))
    except Exception as e:
        print("res %.3f R^2=%d FAILED %r" % (res, radius, e))
    m.close(); del pts
