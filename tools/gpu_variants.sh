#!/bin/bash
# Timing probes of the single-precision gather (AMHIP_F32_VARIANT; wrong heights, timing only):
# prints the gather's HIP-event time per variant.  Usage: gpurun -- bash tools/gpu_variants.sh 0 2 5 6
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
for v in "$@"; do
  AMHIP_F32_VARIANT=$v timeout 300 python "$R/bench.py" --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > "$R/gpurun_out/var_$v.json" 2> "$R/gpurun_out/var_$v.err"
  python - <<P
import json
try:
    d = json.load(open("$R/gpurun_out/var_$v.json"))
    print("variant $v:", d["ms_per_step"], {k: x["ms_per_step"] for k, x in d["kernels"].items()})
except Exception as e:
    print("variant $v failed", e)
P
done
