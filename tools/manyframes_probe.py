#!/usr/bin/env python3
"""Probe: one mosaic batch with more frames than a cull chunk holds (1024): the frame list
is then built per slab and per chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, W, H = 10000, 0.25, 1920, 1080
L = side * res
elev = (400.0 + 10.0 * torch.rand((side, side), device=dev)).float().cpu().numpy()
ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
base = synth.make_frames_torch(64, H, W, 1, 44, dev)
for F in (249, 1000, 1024, 1025, 2000, 4000):
    poses = synth.make_lawnmower_poses(F, L / 2, 700.0, 44, tilt_deg=5.0)
    frames = base[torch.arange(F, device=dev) % 64]      # (F, H, W) gathered view -> contiguous copy
    frames = frames.contiguous()
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        ts = []
        for rep in range(4):
            m.reset(); m.set("elevation", elev); m.synchronize()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mosaic.process(poses, frames, m, sync=False); m.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    print("F = %4d  %.2f ms per batch (min of 4)" % (F, min(ts)))
    del frames
