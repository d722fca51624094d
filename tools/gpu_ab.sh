#!/bin/bash
# A-B timing of several builds of libaerial_mapper_hip.so on ONE box (boxes differ by a few percent):
# every build/ab/<name>.so named on the command line plus the in-tree library ("tree"), interleaved,
# ROUNDS times each.  Usage:
#   cp aerial_mapper_amd/lib/libaerial_mapper_hip.so build/ab/base.so   (before the change)
#   gpurun -- bash tools/gpu_ab.sh 2 base [other ...] [-- bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-2}; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
LIBS+=("tree")
mkdir -p "$R/gpurun_out"
for r in $(seq 1 $ROUNDS); do
  for which in "${LIBS[@]}"; do
    if [ $which = tree ]; then unset AMHIP_LIB_PATH; else export AMHIP_LIB_PATH="$R/build/ab/$which.so"; fi
    timeout 300 python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-host-path "$@" > "$R/gpurun_out/ab_${which}_$r.json" 2> "$R/gpurun_out/ab_${which}_$r.err"
    python - <<P
import json
try:
    d = json.load(open("$R/gpurun_out/ab_${which}_$r.json"))
    print("$which $r:", d["ms_per_step"], {k: x["ms_per_step"] for k, x in d["kernels"].items()})
except Exception as e:
    print("$which $r failed", e)
P
  done
done
