#!/bin/bash
# The other bench lines quoted in DESIGN.md, one GPU call: single-precision mode, cfg2 in both modes, colour, UTM origin, cfg4 / cfg5 at N = 1.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
run() { # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs timeout 600 python "$R/bench.py" --no-cpu-baseline --no-host-path "$@" > "$R/gpurun_out/wl_$name.json" 2> "$R/gpurun_out/wl_$name.err"
  python - <<P
import json
try:
    d = json.load(open("$R/gpurun_out/wl_$name.json"))
    print("$name:", d["value"], "Mcells/s", d["ms_per_step"], "ms", {k: x["ms_per_step"] for k, x in d["kernels"].items()})
except Exception as e:
    print("$name failed", e)
P
}
# (bench.py's --dsm-mode decides the arithmetic, not the environment: default exact = FP64)
run fast A=1 --steps 10 --warmup 3 --dsm-mode fast --no-second-mode --no-rough-terrain
run cfg2 A=1 --steps 10 --warmup 3 --workload cfg2 --no-second-mode --no-rough-terrain
run cfg2_fast A=1 --steps 10 --warmup 3 --workload cfg2 --dsm-mode fast --no-second-mode --no-rough-terrain
run colored A=1 --steps 10 --warmup 3 --colored
run utm A=1 --steps 10 --warmup 3 --map-origin 464980.25,5272690.5
run cfg4 A=1 --steps 5 --warmup 2 --workload cfg4
run cfg5 A=1 --steps 10 --warmup 2 --workload cfg5
