#!/usr/bin/env python3
"""Probe: where the time of one pass through the host-buffer entry points goes (DSM call,
mosaic call, layer downloads), cold and warm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth
dev = torch.device("cuda", 0)
side, res, N, F, W, H = 10000, 0.25, 50_000_000, 249, 1920, 1080
L = side * res
m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
pts = synth.make_points_torch(N, L / 2 + 4, 43, dev).cpu().numpy()
frames = [f for f in synth.make_frames_torch(F, H, W, 1, 44, dev).cpu().numpy()]
poses = synth.make_lawnmower_poses(F, L / 2, 700.0, 44, tilt_deg=5.0)
ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
dsm = A.Dsm(A.DsmSettings(), m); mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
names = ["elevation", "elevation_angle", "observation_index", "ortho"]
h_layers = {n: np.zeros((m.cols, m.rows), np.float32) for n in names}
for rep in range(3):
    m.reset(); m.synchronize()
    t = [time.perf_counter()]
    dsm.process(pts, m); t.append(time.perf_counter())
    mosaic.process(poses, frames, m); t.append(time.perf_counter())
    for n in names:
        m.get(n, out=h_layers[n]); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print("rep", rep, "dsm %.1f  mosaic %.1f  gets %s  total %.1f" % (d[0], d[1], [round(x, 1) for x in d[2:]], (t[-1] - t[0]) * 1e3))
