#!/usr/bin/env python3
"""Probe: mosaic time across grid resolutions, flight altitudes and camera tilts (the
frame-list pruning needs frames that see a whole 64 x 64-cell tile)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
W, H, F = 1920, 1080, 249
frames = synth.make_frames_torch(F, H, W, 1, 44, dev)
ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
for res, side, alt, tilt in ((0.25, 10000, 700.0, 5.0), (1.0, 2500, 700.0, 5.0), (0.1, 10000, 520.0, 5.0),
                             (0.05, 10000, 460.0, 5.0), (0.25, 10000, 700.0, 30.0), (0.25, 10000, 1400.0, 5.0),
                             (2.0, 1250, 700.0, 5.0)):
    L = side * res
    elev = (400.0 + 10.0 * torch.rand((side, side), device=dev)).float().cpu().numpy()
    poses = synth.make_lawnmower_poses(F, L / 2, alt, 44, tilt_deg=tilt)
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        ts = []
        for rep in range(4):
            m.reset(); m.set("elevation", elev); m.synchronize()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mosaic.process(poses, frames, m, sync=False); m.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        cover = float((~np.isnan(m.get("observation_index"))).mean())
    print("res %.2f  %5d^2 cells  alt %4.0f  tilt %2.0f  %.2f ms  %.0f Mcells/s  coverage %.2f" % (
        res, side, alt, tilt, min(ts), side * side / min(ts) / 1e3, cover))
