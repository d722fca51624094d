#!/usr/bin/env python3
"""Probe: ortho::OrthoFromPcl::process at the bench's size, plain and with
use_adaptive_interpolation (x10 retry passes over the cells still empty)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res = 10000, 0.25
L = side * res
for n, adaptive, radius in ((50_000_000, False, 1), (50_000_000, True, 1), (2_000_000, False, 1), (2_000_000, True, 1)):
    pts = synth.make_points_torch(n, L / 2 + 4, 43, dev)
    inten = torch.randint(0, 256, (n,), dtype=torch.int32, device=dev)
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        op = A.OrthoFromPcl(A.OrthoFromPclSettings(interpolation_radius=radius,
                                                   use_adaptive_interpolation=adaptive))
        ts = []
        for rep in range(3):
            m.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            op.process(pts, inten, m, sync=False); m.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        filled = float((m.as_torch("ortho") != 255).float().mean())
    print("%9d points  adaptive=%-5s  %.2f ms  filled %.3f" % (n, adaptive, min(ts), filled))
    del pts, inten
