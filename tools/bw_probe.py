#!/usr/bin/env python3
"""Probe: what a plain device copy / read / write of the sort's payload achieves on this
GPU (the yardstick for the scatter passes: 5.2 TB/s copy, 5.9 read, 6.7 write on MI355X)."""
import time

import torch
dev=torch.device("cuda",0)
for n in (1_200_000_000, 2_400_000_000):
    torch.cuda.synchronize()
    a=torch.empty(n//8,dtype=torch.float64,device=dev).normal_()
    b=torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print("copy %.1f GB: %.3f ms -> %.2f TB/s (read+write)"%(n/1e9, dt*1e3, 2*n/dt/1e12))
    t0=time.perf_counter()
    for _ in range(20): s=a.sum()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print("read-only sum %.1f GB: %.3f ms -> %.2f TB/s"%(n/1e9, dt*1e3, n/dt/1e12))
    t0=time.perf_counter()
    for _ in range(20): b.fill_(1.0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print("write-only fill %.1f GB: %.3f ms -> %.2f TB/s"%(n/1e9, dt*1e3, n/dt/1e12))
