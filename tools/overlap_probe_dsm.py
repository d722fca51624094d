#!/usr/bin/env python3
"""Do the memory-bound sort and the FP64-bound gather of the DSM overlap when two DSM calls run on
two streams?  Two maps of cfg2's size, two clouds: sequential on one stream against one call per
stream (the second started half a call late, so that one's sort meets the other's gather).
    python tools/overlap_probe.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import synth

dev = torch.device("cuda", 0)
side, res, n = 10000, 0.25, 50_000_000
L = side * res
st = A.GridMapSettings(0.0, 0.0, L, L, res)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
maps, dsms, clouds = [], [], []
for k, s in enumerate((s1, s2)):
    m = A.AerialGridMap(st, device=0)
    m.set_stream(s.cuda_stream)
    m.set_dsm_precision(True)
    maps.append(m)
    dsms.append(A.Dsm(A.DsmSettings(1), m))
    clouds.append(synth.make_points_torch(n, L / 2.0 + 4.0, 43 + k, dev))
torch.cuda.synchronize()


def run(parallel, reps=10, lag=True):
    for r in range(2 + reps):
        if r == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if parallel:
            for k in (0, 1):
                with torch.cuda.stream((s1, s2)[k]):
                    maps[k].reset()
                    dsms[k].process(clouds[k], maps[k], sync=False)
        else:
            for k in (0, 1):
                with torch.cuda.stream(s1):
                    maps[k].set_stream(s1.cuda_stream)
                    maps[k].reset()
                    dsms[k].process(clouds[k], maps[k], sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    for k, s in enumerate((s1, s2)):
        maps[k].set_stream(s.cuda_stream)
    return dt * 1e3


out = {"two DSM calls, one stream (ms)": round(run(False), 3), "two DSM calls, two streams (ms)": round(run(True), 3)}
out["again, one stream"] = round(run(False), 3)
out["again, two streams"] = round(run(True), 3)
print(json.dumps(out))
