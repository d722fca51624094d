#!/usr/bin/env python3
"""DSM calls with clouds below the three-pass sort's threshold (incremental mapping: one stereo
pair's cloud onto a large resident map; small surveys): ms per Dsm::process under the sort
implementations (tuning knob sort_one_level, tuning knob p3_min_points=50000, default select).  Round 2, with the
two-level stripe sort still in: one-level 0.269 / 0.141 / 0.174 / 0.107 ms, two-level 0.255 / 0.169 / 0.263 /
0.118 ms, three-pass 0.245 / 0.158 / 0.172 / 0.120 ms on the four cases below -> the stripe sort was retired.
    python tools/small_cloud_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
cases = [("0.7 M points, patch of a 10 000^2 map", 2500.0, 0.25, 700_000, 150.0),
         ("0.25 M points, 2000^2 map", 500.0, 0.25, 250_000, 250.0),
         ("1.0 M points, 2000^2 map", 500.0, 0.25, 1_000_000, 250.0),
         ("60 K points, 1000^2 map @1 m", 1000.0, 1.0, 60_000, 500.0)]
for name, L, res, n, half in cases:
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res), device=0) as m:
        pts = synth.make_points_torch(n, (half, half), 7, dev, center=(100.0, -50.0) if half < L / 2 else (0.0, 0.0))
        dsm = A.Dsm(A.DsmSettings(1), m)
        m.reset()
        for _ in range(3):
            dsm.process(pts, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            dsm.process(pts, m, sync=False)
        m.synchronize()
        print("%-40s %.3f ms per call" % (name, (time.perf_counter() - t0) / K * 1e3))
