#!/usr/bin/env python3
"""Probe: the sort's passes with and without the speculative regions (tuning knob sort_no_speculation)
as the cloud grows from cfg3's 50 M points towards configs[3]'s 400 M on one GPU (0.5 points per
0.25 m cell throughout).  Prints the scatter / count / place slots per DSM call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import hip_lib
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
res = 0.25
for npts in [int(v) for v in os.environ.get("AMHIP_PROBE_POINTS", "100000000,200000000").split(",")]:
    side = int(round((npts / 0.5) ** 0.5 / 64)) * 64
    L = side * res
    pts = synth.make_points_torch(npts, L / 2 + 4, 43, dev)
    for spec in (True, False):
        if spec:
            hip_lib.set_tuning("sort_no_speculation", None)
        else:
            hip_lib.set_tuning("sort_no_speculation", 1)
        with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
            m.set_dsm_sort_reuse(True)      # (opt-in since round 5; the knob above switches it off)
            dsm = A.Dsm(A.DsmSettings(), m)
            for _ in range(2):
                m.reset(); dsm.process(pts, m)
            m.enable_timing(True); m.timing_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                m.reset(); dsm.process(pts, m, sync=False)
            m.synchronize(); dt = (time.perf_counter() - t0) / 3
            kt = {k: round(v[0] / 3, 2) for k, v in m.kernel_times().items() if v[1]}
            print("%4d M points %6d^2 cells  %s  %7.2f ms  %s  %s" %
                  (npts // 1000000, side, "speculative" if spec else "counting   ", dt * 1e3, kt, m.dsm_sort_stats()))
    del pts
