#!/bin/bash
# One GPU call per change: the whole GPU suite, then the default bench line (both gather modes, rough-terrain
# extra, host path) and a one-screen digest.   gpurun --timeout 1500 -- 'bash tools/gpu_round_check.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_check
mkdir -p "$OUT"
cd "$R"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > "$OUT/gpu_tests.log"
timeout 600 python bench.py --steps 10 --warmup 3 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -6 "$OUT/gpu_tests.log"; python - <<'P'
import json,os
o=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/round_check/"
d=json.load(open(o+"bench_cfg3.json"))
print(d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
print(json.dumps(d.get("rough_terrain")))
print(d.get("exact_mode",{}).get("ms_per_step"), d.get("pcie_inclusive",{}).get("ms"), d.get("pcie_inclusive",{}).get("second_pass_ms"))
P
