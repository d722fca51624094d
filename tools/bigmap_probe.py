#!/usr/bin/env python3
"""Probe: BASELINE.json configs[3] on ONE GPU -- 400 M points onto the whole 40 000 x 40 000
@0.25 m map (1.6e9 cells, 6.4 GB per layer), then 64-frame batches of a 2000-frame flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, N, F, B, W, H = 40000, 0.25, 400_000_000, 2000, 64, 1920, 1080
L = side * res
m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
pts = torch.empty((N, 3), dtype=torch.float64, device=dev)
q = N // 8
for k in range(8):          # generated in slices: the generator's temporaries stay small
    pts[k * q:(k + 1) * q] = synth.make_points_torch(q, L / 2 + 4, 45 + k, dev)
dsm = A.Dsm(A.DsmSettings(), m)
for rep in range(3):
    m.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    dsm.process(pts, m, sync=False); m.synchronize()
    dt = time.perf_counter() - t0
    print("DSM 400 M points -> 1.6e9 cells: %.1f ms (%.0f Mcells/s, %.0f Mpts/s)" % (dt * 1e3, side * side / dt / 1e6, N / dt / 1e6))
print("free HBM now: %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
# (no peeking at the elevation layer here: handing out its device pointer would invalidate
# the height range the DSM tracked for the mosaic's coarse pre-cull)
frames = synth.make_frames_torch(B, H, W, 1, 46, dev)
poses = synth.make_lawnmower_poses(F, L / 2, 700.0, 46, tilt_deg=5.0)
ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for lo in range(0, F, B):
        hi = min(lo + B, F)
        mosaic.process(poses[lo:hi], frames[:hi - lo], m, sync=False)
    m.synchronize(); dt = time.perf_counter() - t0
    print("2000 frames in 64-frame batches onto the whole map: %.1f ms (%.2f ms per batch)" % (dt * 1e3, dt * 1e3 / 32))
print("coverage %.3f, NaN elevations %.2e" % (
    float((~torch.isnan(m.as_torch("observation_index"))).float().mean()),
    float(torch.isnan(m.as_torch("elevation")).float().mean())))
