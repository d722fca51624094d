#!/usr/bin/env python3
"""Supplementary benchmark of ortho::OrthoForwardHomography::batch on one MI355X
(NOT the headline metric -- that is bench.py).

    python tools/bench_forward.py [--steps K] [--warmup W] [--frames F] [--mosaic N]

One step = reset + batch() of F synthetic 1920x1080 8UC1 frames (the cfg3
flight: lawn-mower at 700 m over a 400 m ground plane, +-5 deg tilt) into an
N x N mosaic at 1 m/pixel, frames resident in HBM.  Prints one JSON line:
imagery throughput (source pixels per second), the algorithmic bytes the
kernels move, the CPU oracle timed on a bounded sample of the same frames and a
bit-exact parity check on that sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=249)
    ap.add_argument("--mosaic", type=int, default=2500)
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench_forward.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    W, H, f, alt, ground = 1920, 1080, 1400.0, 700.0, 400.0
    F, N = args.frames, args.mosaic
    ncam = A.NCamera(f, f, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    st = A.OrthoForwardHomographySettings(ground_plane_elevation_m=ground, width_mosaic_pixels=N,
                                          height_mosaic_pixels=N)
    frames = synth.make_frames_torch(F, H, W, 1, 44, dev)
    poses = synth.make_lawnmower_poses(F, N / 2.0, alt, 44, tilt_deg=5.0)
    mosaic = A.OrthoForwardHomography(ncam, st)
    mosaic.set_stream(stream.cuda_stream)

    def step():
        mosaic.reset()
        mosaic.batch(poses, frames, sync=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps

    # algorithmic bytes: every frame's footprint region is written (u8 + mask),
    # distance-transformed (2 passes over 1 B) and fed once; the mosaic
    # accumulators (6 + 4 B) and the result (6 + 1 B) are touched once per pass
    foot = (W / f * (alt - ground)) * (H / f * (alt - ground))  # m^2 = pixels at 1 m/px
    region_px = F * foot
    out = {
        "metric": "Mpixels/s (OrthoForwardHomography::batch, source imagery)",
        "value": round(F * W * H / dt / 1e6, 1), "unit": "Mpixels/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
        "higher_is_better": True, "dtype": "f64 homography / u8+i16+f32 raster", "data": "synthetic",
        "config": {"workload": "%d frames %dx%d 8UC1 -> %dx%d mosaic @1 m/px, feather blend" %
                               (F, W, H, N, N),
                   "step": "mosaic reset + batch(): warp (nearest), exact L1 feather weights, "
                           "feed, blend; frames resident in HBM"},
        "frames_per_s": round(F / dt, 1),
        "footprint_pixel_frames": int(region_px),
    }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as O
        k = min(args.cpu_frames, F)
        cam = O.Camera()
        cam.fu = cam.fv = f
        cam.cu, cam.cv = (W - 1) / 2.0, (H - 1) / 2.0
        cam.width, cam.height = W, H
        desc = O.mosaic_desc(N, N, ground)
        host = frames[:k].cpu().numpy()
        fm = O.ForwardMosaic(cam, desc)
        t0 = time.perf_counter()
        assert fm.batch(poses[:k], [x for x in host]) == O.OK
        tc = time.perf_counter() - t0
        mosaic.reset()
        mosaic.batch(poses[:k], frames[:k])
        res, mask = mosaic.result()
        out["cpu_baseline"] = {
            "value": round(k * W * H / tc / 1e6, 2), "unit": "Mpixels/s", "cores": 1,
            "kind": "port",
            "sample": "oracle/amo_forward.cc batch() of the first %d frames into the same mosaic "
                      "(%.2f s; like OpenCV it warps and distance-transforms the WHOLE mosaic per "
                      "frame)" % (k, tc)}
        out["parity_sample"] = {"frames": k,
                                "result_mismatch_values": int((res != fm.result).sum()),
                                "mask_mismatch_pixels": int((mask != fm.mask).sum()),
                                "covered_fraction": round(float((fm.mask > 0).mean()), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
