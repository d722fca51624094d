#!/bin/bash
# Bench lines under different environments on ONE box.  Usage:
#   gpurun -- bash tools/gpu_env_ab.sh "" "AMHIP_TUNING=ortho_no_prune=1" ...   (one run per argument)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
n=0
for e in "$@"; do
  n=$((n+1))
  env $e timeout 300 python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-host-path > "$R/gpurun_out/env_$n.json" 2> "$R/gpurun_out/env_$n.err"
  python - <<P
import json
try:
    d = json.load(open("$R/gpurun_out/env_$n.json"))
    print("[$e]:", d["ms_per_step"], {k: x["ms_per_step"] for k, x in d["kernels"].items()})
except Exception as ex:
    print("[$e] failed", ex)
P
done
