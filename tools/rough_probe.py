#!/usr/bin/env python3
"""DSM calls on the rough-terrain scene of bench.py (25 m steps through one in five gather tiles),
fast and exact mode: ms per call and where the tiles went.  Under rocprofv3 --kernel-trace --stats
it shows which launch takes the time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

side, res, n = 10000, 0.25, 50_000_000
L = side * res
dev = torch.device("cuda", 0)
m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
pts = synth.make_points_torch(n, L / 2 + 4.0, 43, dev)
u = (L / 2.0 - pts[:, 0]) / (64 * res)
v = (L / 2.0 - pts[:, 1]) / (16 * res)
ti, tj = torch.floor(u).to(torch.int64), torch.floor(v).to(torch.int64)
chosen = (((ti * 73856093) ^ (tj * 19349663)) % 5) == 0
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0     # share of the chosen tiles that get a step
if frac < 1.0:
    chosen &= (((ti * 2654435761) ^ (tj * 40503)) % 1000) < int(frac * 1000)
pts[:, 2] += 25.0 * (chosen & ((u - torch.floor(u)) > 0.5)).to(torch.float64)
del u, v, ti, tj, chosen
dsm = A.Dsm(A.DsmSettings(1), m)
for mode in ("fast", "exact"):
    m.set_dsm_precision(mode == "exact")
    for _ in range(2):
        m.reset()
        dsm.process(pts, m, sync=False)
    m.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.reset()
        dsm.process(pts, m, sync=False)
    torch.cuda.synchronize()
    print(mode, "ms per DSM call %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), m.dsm_gather_stats())
