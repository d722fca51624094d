#!/usr/bin/env python3
"""Probe: host <-> device copy rates for pageable and pinned buffers of the host path's sizes."""
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
n = 1_200_000_000
a = np.ones(n // 8, np.float64)
t = torch.from_numpy(a)
d = torch.empty(n // 8, dtype=torch.float64, device=dev)
p = torch.empty(n // 8, dtype=torch.float64).pin_memory()
for name, src in (("pageable", t), ("pinned", p)):
    for _ in range(2):
        d.copy_(src); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d.copy_(src); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("H2D 1.2 GB %-8s %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
    t0 = time.perf_counter()
    for _ in range(3):
        src.copy_(d); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("D2H 1.2 GB %-8s %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
t0 = time.perf_counter()
for _ in range(3):
    p.copy_(t)
dt = (time.perf_counter() - t0) / 3
print("host memcpy pageable -> pinned (torch, %d threads) %.1f ms  %.1f GB/s" % (torch.get_num_threads(), dt * 1e3, n / dt / 1e9))
