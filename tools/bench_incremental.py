#!/usr/bin/env python3
"""Supplementary benchmark of BASELINE.json configs[4] at ONE rank's share (NOT the
headline metric -- that is bench.py):

    python tools/bench_incremental.py [--frames 2000] [--batch 64] [--window k]

The 40 000 x 40 000 @ 0.25 m survey map is split into 2 x 4 windows of 20 000 x 10 000
cells; this process owns window k, its layers resident in HBM.
The DSM of the window's 50 M points is built once (untimed).  Then the 2000 frames of a
lawn-mower flight over the WHOLE 10 km map arrive in batches of 64 (31 x 64 + 16) -- every
rank sees every batch (frames are broadcast, culling is per tile) and folds it into its
window with OrthoBackwardGrid::process; nothing is reset in between.  Prints one JSON
line: time per batch, batches / frames per second, and the window's final coverage.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--window", type=int, default=5, help="0..7: which of the 2 x 4 windows")
    ap.add_argument("--points", type=int, default=50_000_000)
    args = ap.parse_args()

    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth, tiling

    if not torch.cuda.is_available():
        raise SystemExit("bench_incremental.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    side, res, W, H = 40000, 0.25, 1920, 1080
    L = side * res
    layout = tiling.TileLayout(side, side, 2, 4)
    win = layout.window(args.window)
    m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res), window=win)
    m.set_stream(stream.cuda_stream)
    # the window in map coordinates (x decreases with i, y with j)
    x_hi = L / 2.0 - win[0] * res
    y_hi = L / 2.0 - win[1] * res
    wx, wy = win[2] * res, win[3] * res
    n = args.points
    g = torch.Generator(device=dev)
    g.manual_seed(45 + args.window)
    pts = torch.empty((n, 3), dtype=torch.float64, device=dev)
    pts[:, 0] = x_hi + 4.0 - torch.rand(n, dtype=torch.float64, device=dev, generator=g) * (wx + 8.0)
    pts[:, 1] = y_hi + 4.0 - torch.rand(n, dtype=torch.float64, device=dev, generator=g) * (wy + 8.0)
    pts[:, 2] = 400.0 + 10.0 * torch.sin(0.01 * pts[:, 0]) * torch.cos(0.01 * pts[:, 1])
    pts[:, 2] += (torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2.0 - 1.0) * 0.05
    F, B = args.frames, args.batch
    poses = synth.make_lawnmower_poses(F, L / 2.0, 700.0, 46, tilt_deg=5.0)
    # (the pixel content does not matter for the timing: one resident batch is re-used)
    frames = synth.make_frames_torch(B, H, W, 1, 46, dev)
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    dsm = A.Dsm(A.DsmSettings(), m)
    mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
    dsm.process(pts, m)
    m.synchronize()

    def flight():
        for lo in range(0, F, B):
            hi = min(lo + B, F)
            mosaic.process(poses[lo:hi], frames[:hi - lo], m, sync=False)

    flight()                       # warm-up (also materializes the layers)
    m.synchronize()
    m.reset()
    dsm.process(pts, m)
    m.synchronize()
    m.enable_timing(True)
    m.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flight()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.synchronize()
    kt = m.kernel_times()
    m.enable_timing(False)
    nb = (F + B - 1) // B
    idx = m.as_torch("observation_index")
    cover = float((~torch.isnan(idx)).float().mean())
    ms, launches = kt.get("k_ortho_backward", (0.0, 0))
    print(json.dumps({
        "metric": "64-frame batches appended per second onto one rank's window (cfg5 share)",
        "value": round(nb / dt, 1), "unit": "batches/s", "frames_per_s": round(F / dt, 1),
        "ms_per_batch": round(dt / nb * 1e3, 3), "kernel_ms_per_batch": round(ms / max(launches, 1), 3),
        "batches": nb, "frames": F, "n_gpus": 1, "data": "synthetic", "dtype": "f64",
        "config": {"workload": "cfg5 share: window %d of the 2 x 4 tiling of 40000 x 40000 @0.25 m "
                               "(%d x %d cells, %d points), %d frames 1920x1080 in batches of %d, "
                               "layers resident" % (args.window, win[2], win[3], n, F, B)},
        "window_coverage": round(cover, 4),
        "cells_per_s_over_the_flight": round(win[2] * win[3] / dt / 1e6, 1)}))


if __name__ == "__main__":
    main()
