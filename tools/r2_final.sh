#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_final; mkdir -p "$OUT"; cd "$R"
bash tools/collect_profiles.sh v2 r02 > "$OUT/collect.txt" 2>&1
tail -4 "$OUT/collect.txt"
timeout 600 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-host-path > "$OUT/r02_bench_cfg2_dsm_only.json" 2>/dev/null
AMHIP_DSM_EXACT=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-host-path > "$OUT/r02_bench_cfg3_fp64_mode.json" 2>/dev/null
timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 > "$OUT/r02_bench_cfg4_n1.json" 2>/dev/null
timeout 600 python bench.py --workload cfg5 --steps 31 --warmup 2 > "$OUT/r02_bench_cfg5_n1.json" 2>/dev/null
timeout 600 python bench.py --workload cfg2 --knn 4 --steps 3 --warmup 1 > "$OUT/r02_bench_cfg2_knn4_optional.json" 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-sample-side 10000 --no-host-path > "$OUT/r02_bench_cfg3_full_parity.json" 2>/dev/null
for f in "$OUT"/r02_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("parity_sample"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
