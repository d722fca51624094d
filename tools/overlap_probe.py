#!/usr/bin/env python3
"""Probe: does this GPU overlap the HBM-bound sort passes of one context with the
VALU-bound mosaic kernel of another (two HIP streams)?  Prints sequential vs
concurrent wall time per pair of calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, N, F, W, H = 10000, 0.25, 50_000_000, 249, 1920, 1080
L = side * res
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
m1 = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)); m1.set_stream(s1.cuda_stream)
m2 = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)); m2.set_stream(s2.cuda_stream)
pts = synth.make_points_torch(N, L / 2 + 4, 43, dev)
frames = synth.make_frames_torch(F, H, W, 1, 44, dev)
poses = synth.make_lawnmower_poses(F, L / 2, 700.0, 44, tilt_deg=5.0)
ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
d1 = A.Dsm(A.DsmSettings(), m1); d2 = A.Dsm(A.DsmSettings(), m2)
o2 = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m2)
torch.cuda.synchronize()
d2.process(pts, m2); m2.synchronize()          # elevation for the mosaic

def seq():
    m1.reset(); d1.process(pts, m1, sync=False); m1.synchronize()
    m2.set("elevation_angle", None) if False else None
    o2.process(poses, frames, m2, sync=False); m2.synchronize()

def conc():
    m1.reset(); d1.process(pts, m1, sync=False)
    o2.process(poses, frames, m2, sync=False)
    m1.synchronize(); m2.synchronize()

for name, fn in (("sequential", seq), ("concurrent", conc), ("sequential", seq), ("concurrent", conc)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(name, "ms per (dsm on ctx1 + mosaic on ctx2):", round((time.perf_counter() - t0) * 100, 3))
