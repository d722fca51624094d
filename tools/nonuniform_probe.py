#!/usr/bin/env python3
"""Probe: DSM time for clouds whose density is not uniform over the map (the gather's LDS
capacity is chosen from a density estimate; over-full tiles fall back to the global path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, N = 10000, 0.25, 50_000_000
L = side * res
m = A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res))
dsm = A.Dsm(A.DsmSettings(), m)

def run(name, pts):
    for _ in range(2):
        m.reset(); dsm.process(pts, m)
    m.enable_timing(True); m.timing_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.reset(); dsm.process(pts, m, sync=False)
    m.synchronize(); dt = (time.perf_counter() - t0) / 5
    kt = {k: round(v[0] / 5, 3) for k, v in m.kernel_times().items() if v[1]}
    m.enable_timing(False)
    print("%-34s %.2f ms  %s" % (name, dt * 1e3, kt))

uni = synth.make_points_torch(N, L / 2 + 4, 43, dev)
run("uniform 0.5 pts/cell", uni)
half = N // 2
mix = torch.empty_like(uni)
mix[:half] = synth.make_points_torch(half, L / 2 + 4, 44, dev)
mix[half:] = synth.make_points_torch(N - half, L / 4, 45, dev, center=(L / 4, L / 4))
run("half of the points in one quarter", mix)
strip = torch.empty_like(uni)
strip[:] = synth.make_points_torch(N, L / 2 + 4, 46, dev)
strip[:, 1] = strip[:, 1] * 0.5           # everything squeezed into the middle half: 1 pt/cell there
run("all points in half of the map", strip)

# spatially ORDERED clouds (scan lines, raster order of a stereo densifier, pre-tiled data):
# every chunk of the sort then holds one or two partitions
order = torch.argsort(uni[:, 1])
run("uniform, sorted by northing", uni[order].contiguous())
key = (torch.floor(uni[:, 1] / 4.0) * 100000.0 + uni[:, 0])
order = torch.argsort(key)
run("uniform, sorted in 4 m strips", uni[order].contiguous())
del order, key
