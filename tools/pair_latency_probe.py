#!/usr/bin/env python3
"""Probe: what ONE stereo pair of the incremental demo costs (main-ortho-backward-grid-incremental.cc:
143-166: Dsm::process of the pair's cloud + OrthoBackwardGrid::process of its one frame onto a
large resident map) -- small clouds, one frame: launch- and dispatch-bound territory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
W, H = 752, 480
for side in (10000, 40000):
    res = 0.25
    L = side * res
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        m.set_stream(stream.cuda_stream)
        ncam = A.NCamera(450.0, 450.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
        mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
        dsm = A.Dsm(A.DsmSettings(), m)
        frames = synth.make_frames_torch(8, H, W, 1, 5, dev)
        poses = synth.make_lawnmower_poses(8, L / 8.0, 100.0 + 400.0, 5, tilt_deg=3.0)
        clouds = []
        for k in range(8):   # a 170 x 110 m patch under each pose, 360 K points
            cx, cy = float(poses[k][0]), float(poses[k][1])
            clouds.append(synth.make_points_torch(360_000, (85.0, 55.0), 60 + k, dev, center=(cx, cy)))
        m.reset()
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(8):
                dsm.process(clouds[k], m, sync=False)
                mosaic.process(poses[k:k + 1], frames[k:k + 1], m, sync=False)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        m.enable_timing(True); m.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(8):
            dsm.process(clouds[k], m, sync=False)
            mosaic.process(poses[k:k + 1], frames[k:k + 1], m, sync=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        m.synchronize()
        kt = {k: round(v[0] / 8, 4) for k, v in m.kernel_times().items() if v[1]}
        print("map %d^2: %.3f ms per stereo pair (DSM of 360 K points + 1 frame)" % (side, dt * 1e3), kt)
