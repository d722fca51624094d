#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_run4; mkdir -p "$OUT"; cd "$R"
timeout 1200 python -m pytest tests/test_gpu_dsm_fast.py tests/test_gpu_tiling.py tests/test_gpu_bench_multirank.py -q -rf --no-header > "$OUT/pytest.txt" 2>&1
tail -30 "$OUT/pytest.txt"
timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 > "$OUT/bench_cfg4_n1.json" 2> "$OUT/bench_cfg4_n1.err"; tail -c 1500 "$OUT/bench_cfg4_n1.json"; tail -3 "$OUT/bench_cfg4_n1.err"
timeout 600 python bench.py --workload cfg5 --steps 31 --warmup 2 > "$OUT/bench_cfg5_n1.json" 2> "$OUT/bench_cfg5_n1.err"; tail -c 1500 "$OUT/bench_cfg5_n1.json"; tail -3 "$OUT/bench_cfg5_n1.err"
