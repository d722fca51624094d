#!/usr/bin/env python3
"""Soak: the randomized DSM / mosaic parity tests of tests/test_gpu_parity.py over many more
seeds than the suite runs (GPU vs oracle, bit-exact mosaic layers, DSM <= 1e-4 m).
    python tools/soak.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
t0 = time.time()
for seed in range(first, first + count):
    for name, fn in (("dsm", T.test_dsm_random_configurations), ("ortho", T.test_ortho_random_configurations),
                     ("fwd", getattr(T, "test_forward_random_configurations", None))):
        if fn is None:
            continue
        try:
            fn(seed)
        except Exception as e:  # keep going: report every failing seed
            bad.append((name, seed, repr(e)[:200]))
            traceback.print_exc()
print("soak: seeds %d..%d, %d failures, %.0f s" % (first, first + count - 1, len(bad), time.time() - t0))
for b in bad:
    print("FAILED", b)
sys.exit(1 if bad else 0)
