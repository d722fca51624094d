#!/usr/bin/env python3
"""Probe: DSM time per point / per cell as the cloud gets denser than the bench's 0.5
points per cell (dense stereo clouds reach tens of points per 0.25 m cell)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
N, res = 50_000_000, 0.25
DENS = [float(v) for v in os.environ.get("AMHIP_PROBE_DENSITIES", "0.5,1,2,4,8,16").split(",")]
for dens in DENS:
    side = int(round((N / dens) ** 0.5 / 64)) * 64
    L = side * res
    m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
    dsm = A.Dsm(A.DsmSettings(), m)
    pts = synth.make_points_torch(N, L / 2 + 4, 43, dev)
    for exact in (True, False):     # the library's default (FP64) and the opt-in single-precision mode
        m.set_dsm_precision(exact)
        for _ in range(2):
            m.reset(); dsm.process(pts, m)
        m.enable_timing(True); m.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m.reset(); dsm.process(pts, m, sync=False)
        m.synchronize(); dt = (time.perf_counter() - t0) / 3
        kt = {k: round(v[0] / 3, 2) for k, v in m.kernel_times().items() if v[1]}
        print("%5.1f pts/cell  %5d^2 cells  %-5s %7.2f ms  %6.1f Mpts/s  %s" %
              (dens, side, "FP64" if exact else "f32", dt * 1e3, N / dt / 1e6, kt))
    m.close(); del pts
