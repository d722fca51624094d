#!/usr/bin/env python3
"""Probe: BASELINE configs[3] on ONE GPU (400 M points -> 40 000 x 40 000 cells) in both arithmetic
modes -- the sort's placement in rounds (sub-partitions of 13 K points: amhip_sort.hip place_rounds,
doubles and records) at full size: the two modes' maps must agree within 1e-4 m with one NaN pattern."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, n = 40000, 0.25, 400_000_000
L = side * res
pts = synth.make_points_torch(n, L / 2.0 + 4.0, 45, dev)
with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, L, L, res)) as m:
    dsm = A.Dsm(A.DsmSettings(), m)
    maps = {}
    for mode in ("exact", "fast"):
        m.set_dsm_precision(mode == "exact")
        m.reset(); dsm.process(pts, m)
        m.enable_timing(True); m.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.reset(); dsm.process(pts, m)
        dt = time.perf_counter() - t0
        kt = {k: round(v[0], 3) for k, v in m.kernel_times().items() if v[1]}
        m.enable_timing(False)
        maps[mode] = m.as_torch("elevation").clone()
        torch.cuda.synchronize()   # (torch's stream: the map's own stream must not refill the layer under the copy)
        print(mode, "%.2f ms" % (dt * 1e3), kt, m.dsm_stats())
    a, b = maps["exact"], maps["fast"]
    nan_equal = bool(torch.equal(torch.isnan(a), torch.isnan(b)))
    d = (a - b).abs()
    d = d[~torch.isnan(d)]
    na, nb = torch.isnan(a), torch.isnan(b)
    print("NaNs exact %d fast %d, only-exact %d only-fast %d" % (int(na.sum()), int(nb.sum()), int((na & ~nb).sum()), int((nb & ~na).sum())))
    idx = torch.nonzero(na != nb)[:5]
    print("first mismatches (j, i):", idx.tolist())
    dd = (a - b)
    ok = ~(na | nb)
    print("mean diff %.3g, frac a>b %.4f, frac a<b %.4f" % (float(dd[ok].double().mean()), float((dd[ok] > 0).float().mean()), float((dd[ok] < 0).float().mean())))
    print("NaN pattern equal:", nan_equal, " max |dh| %.3g m" % float(d.max()), " bit-identical %.6f" % float((a.view(torch.int32) == b.view(torch.int32)).float().mean()))
    assert nan_equal and float(d.max()) <= 1e-4
