#!/bin/bash
# round 3, GPU call 1: the test suite, the real-cycle micro-benchmarks, the bench line with its
# new objects (exact_mode, rough_terrain, traffic checks).  Usage: gpurun -- 'bash tools/r3_run1.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_run1
mkdir -p "$OUT"
cd "$R"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > "$OUT/gpu_tests.log"
( timeout 120 tools/ubench/ubench3 A; timeout 120 tools/ubench/ubench3 B; timeout 120 tools/ubench/ubench3 C ) > "$OUT/ubench3.jsonl" 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -5 "$OUT/gpu_tests.log"; tail -4 "$OUT/ubench3.jsonl"; head -c 1500 "$OUT/bench_cfg3.json"; tail -3 "$OUT/bench_cfg3.err"
