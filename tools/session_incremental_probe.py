"""What the incremental demo's calls cost through the HOST-matrix route on a large map: one stereo
pair's cloud and one frame per call onto a `side`^2 map (default 20 000^2 cells: 1.6 GB per layer),
with whole-window downloads (tuning knob session_no_partial) and with the dirty rectangle only.  The
host-side content sums (O(map) per call) stay in both.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=20000)
    ap.add_argument("--calls", type=int, default=4)
    args = ap.parse_args()
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import hip_lib
    from aerial_mapper_amd import synth
    side, res = args.side, 0.25
    L = side * res
    dev = torch.device("cuda", 0)
    pts = synth.make_points_torch(int(side * side * 0.5), L / 2.0 + 3.0, 5, dev).cpu().numpy()
    W, H = 1920, 1080
    F = 2 + 2 * args.calls
    frames = synth.make_frames_torch(F, H, W, 1, 6, dev).cpu().numpy()
    poses = synth.make_lawnmower_poses(F, L / 5.0, 400.0 + 150.0, 6, tilt_deg=4.0)
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    rng = np.random.default_rng(7)
    zmed = float(np.median(pts[::1000, 2]))
    out = {"side": side, "layer_MB": side * side * 4 / 1e6}
    for mode in ("window", "rectangle"):
        if mode == "window":
            hip_lib.set_tuning("session_no_partial", 1)
        else:
            hip_lib.set_tuning("session_no_partial", None)
        with A.HostSession(A.GridMapSettings(0.0, 0.0, L, L, res)) as hs:
            hs.dsm_process(A.DsmSettings(1), pts)
            hs.ortho_process(ncam, A.OrthoSettings(), poses[:2], frames[:2])
            t_dsm, t_mos = [], []
            d0 = hs.transfer_stats()[1]
            for k in range(args.calls):
                cx, cy = rng.uniform(-L / 3, L / 3, 2)
                pair = np.c_[rng.uniform(cx - 60.0, cx + 60.0, 360000), rng.uniform(cy - 40.0, cy + 40.0, 360000),
                             zmed + rng.uniform(-1.0, 1.0, 360000)]
                t0 = time.perf_counter()
                hs.dsm_process(A.DsmSettings(1), pair)
                t1 = time.perf_counter()
                hs.ortho_process(ncam, A.OrthoSettings(), poses[2 + k:3 + k], frames[2 + k:3 + k])
                t2 = time.perf_counter()
                t_dsm.append((t1 - t0) * 1e3)
                t_mos.append((t2 - t1) * 1e3)
            out[mode] = {"dsm_call_ms": round(float(np.median(t_dsm)), 2),
                         "mosaic_call_ms": round(float(np.median(t_mos)), 2),
                         "downloaded_MB_per_pair": round((hs.transfer_stats()[1] - d0) / args.calls / 1e6, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
