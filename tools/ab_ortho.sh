#!/bin/bash
# A/B of the backward-grid kernel builds on the GPU box (one gpurun call):
#   gpurun --timeout 1200 -- 'bash tools/ab_ortho.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_ortho
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_ortho_fold.py tests/test_gpu_golden.py -x -q -m gpu > "$OUT/pytest_fold.log" 2>&1
echo "pytest fold rc=$?" | tee -a "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k ortho > "$OUT/pytest_parity_ortho.log" 2>&1
echo "pytest parity-ortho rc=$?" | tee -a "$OUT/summary.txt"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path"
AMHIP_ORTHO_EXACT_FOLD=1 AMHIP_ORTHO_NO_PRUNE=1 timeout 300 $B > "$OUT/bench_exact_noprune.json" 2> "$OUT/bench_exact_noprune.err"
AMHIP_ORTHO_EXACT_FOLD=1 timeout 300 $B > "$OUT/bench_exact.json" 2> "$OUT/bench_exact.err"
AMHIP_ORTHO_NO_PRUNE=1 timeout 300 $B > "$OUT/bench_fast4_noprune.json" 2> "$OUT/bench_fast4_noprune.err"
AMHIP_ORTHO_FAST_WAVES=3 timeout 300 $B > "$OUT/bench_fast3.json" 2> "$OUT/bench_fast3.err"
AMHIP_ORTHO_FAST_WAVES=4 timeout 300 $B > "$OUT/bench_fast4.json" 2> "$OUT/bench_fast4.err"
for v in exact_noprune exact fast4_noprune fast3 fast4; do
  python - "$OUT/bench_$v.json" "$v" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "ortho", d["kernels"]["k_ortho_backward"]["ms_per_step"],
          "gather", d["kernels"]["k_dsm_gather"]["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
tail -n 3 "$OUT/pytest_fold.log"; tail -n 3 "$OUT/pytest_parity_ortho.log"
