#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_run5; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_session.py -q -x --no-header > "$OUT/pytest.txt" 2>&1
tail -30 "$OUT/pytest.txt"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
python - "$OUT/bench_cfg3.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["ms_per_step"], json.dumps(d.get("pcie_inclusive"), indent=1))
PY
tail -3 "$OUT/bench_cfg3.err"
