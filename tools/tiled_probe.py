#!/usr/bin/env python3
"""What the multi-GPU step costs ON the GPU besides the RCCL call itself: rank 0 of a
2 x 1 tiling of bench.py's cfg3 map runs alone, its neighbour replaced by a stand-in that
echoes rank 0's own send rows back (same volume as a real neighbour's strip).
  plain   reset + Dsm::process + mosaic            (bench.py at N = 1)
  routed  reset + tiling.route_points + Dsm + mosaic   (selection pass of its own, counts
          read on the host, two exchanges)
  tiled   reset + tiling.TiledDsm + mosaic         (selection inside the binning pass, one
          exchange of equal splits, no host synchronisation)
    python tools/tiled_probe.py [--steps 10]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Echo(object):
    """One-process stand-in for the two collectives: what was selected for the neighbour
    comes back as if the neighbour had sent it."""

    def exchange_counts(self, sc):
        return sc.clone()

    def exchange_rows(self, out_rows, in_rows, recv_counts, send_counts):
        out_rows.copy_(in_rows)

    def exchange_equal(self, out_rows, in_rows):
        out_rows.copy_(in_rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import synth, tiling
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    side, res, n_pts, F, W, H = 10000, 0.25, 50_000_000, 249, 1920, 1080
    L = side * res
    world = 2
    layout = tiling.TileLayout(world * side, side, world, 1)
    win = layout.window(0)
    m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, world * L, L, res), window=win)
    m.set_stream(stream.cuda_stream)
    center = (world * L / 2.0 - (win[0] + win[2] / 2.0) * res, 0.0)
    cap = tiling.halo_strip_rows(n_pts / (L * L), L, 1, res)
    buf = torch.empty((n_pts + world * cap, 3), dtype=torch.float64, device=dev)
    buf[:n_pts] = synth.make_points_torch(n_pts, (win[2] * res / 2.0, L / 2.0), 43, dev, center=center)
    cxx, cyy = tiling.cell_coords(buf[:n_pts], m.grid)
    kept = buf[:n_pts][tiling.owner_mask(cxx, cyy, win)]
    n = int(kept.shape[0])
    buf[:n] = kept
    del cxx, cyy, kept
    pts = buf[:n]
    frames = synth.make_frames_torch(F, H, W, 1, 44, dev)
    poses = synth.make_lawnmower_poses(F, L / 2.0, 700.0, 44, tilt_deg=5.0, center=center)
    ncam = A.NCamera(1400.0, 1400.0, (W - 1) / 2.0, (H - 1) / 2.0, W, H)
    dsm = A.Dsm(A.DsmSettings(), m)
    mosaic = A.OrthoBackwardGrid(ncam, A.OrthoSettings(), m)
    tiled = tiling.TiledDsm(dsm.settings, m, layout, 0, cap, comm=Echo())

    def plain():
        dsm.process(pts, m, sync=False)

    def routed():
        cloud = tiling.route_points(pts, m.grid, layout, 0, radius_sq=1, map_=m, assume_owned=True,
                                    cap=cap, workspace=buf, comm=Echo())
        dsm.process(cloud, m, sync=False)

    def run_tiled():
        tiled.process(buf, n, sync=False)

    out = {}
    for name, fn in (("plain", plain), ("routed", routed), ("tiled", run_tiled)):
        for k in range(3 + args.steps):
            if k == 3:
                m.enable_timing(True)
                m.timing_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            m.reset()
            fn()
            mosaic.process(poses, frames, m, sync=False)
        torch.cuda.synchronize()
        out[name + "_ms_per_step"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
        m.synchronize()
        out[name + "_kernels"] = {k: round(v[0] / args.steps, 3) for k, v in m.kernel_times().items() if v[1]}
        m.enable_timing(False)
    tiled.check_overflow()
    out["halo_rows_selected"] = int(tiled.counts.sum().item())
    out["rows_per_pair"] = cap
    out["workload"] = "rank 0 of 2 x 1 windows of cfg3 (%d own points), neighbour echoed" % n
    print(json.dumps(out))


if __name__ == "__main__":
    main()
