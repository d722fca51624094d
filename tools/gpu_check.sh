#!/bin/bash
# One GPU call: the whole GPU suite, then the cfg3 bench line without the CPU legs (per-kernel HIP-event
# times).  Usage (through gpurun): bash tools/gpu_check.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
timeout 900 python -m pytest "$R/tests" -m gpu -x -q 2>&1 | tail -6 > "$R/gpurun_out/gputests.log"
timeout 300 python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-host-path "$@" > "$R/gpurun_out/bench_quick.json" 2> "$R/gpurun_out/bench_quick.err"
tail -3 "$R/gpurun_out/gputests.log"
python - <<P
import json
d = json.load(open("$R/gpurun_out/bench_quick.json"))
print(d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
P
