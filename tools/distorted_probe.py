#!/usr/bin/env python3
"""Probe: mosaic time for a camera with a distortion model (k_ortho_backward, every pair in
the reference's arithmetic, conservative view-cone cull) on 36 M cells x 249 frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth, hip_lib as L

dev = torch.device("cuda", 0)
side, res, F, W, H = 6000, 0.25, 249, 1920, 1080
Lm = side * res
frames = synth.make_frames_torch(F, H, W, 1, 44, dev)
poses = synth.make_lawnmower_poses(F, Lm / 2, 700.0, 44, tilt_deg=5.0)
elev = (400.0 + 10.0 * torch.rand((side, side), device=dev)).float().cpu().numpy()
for name, model, dist in (("pinhole", L.DIST_NONE, (0, 0, 0, 0)),
                          ("radtan", L.DIST_RADTAN, (-0.28, 0.07, 2e-4, -1e-4)),
                          ("equidistant", L.DIST_EQUIDISTANT, (-0.01, 0.02, -0.005, 0.001))):
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H, model, dist)
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, Lm, Lm, res)) as m:
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        ts = []
        for rep in range(5):
            m.reset(); m.set("elevation", elev); m.synchronize()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mosaic.process(poses, frames, m, sync=False); m.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        cover = float((~np.isnan(m.get("observation_index"))).mean())
    print("%-12s %.2f ms per batch (min of 5), coverage %.3f" % (name, min(ts), cover))
