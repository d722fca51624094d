#!/bin/bash
# SQ instruction counters of the backward-grid kernel builds (one gpurun call):
#   gpurun --timeout 1200 -- 'bash tools/sq_ortho.sh'   -> gpurun_out/sq_ortho/sq_<variant>.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_ortho
O=/tmp/sq_ortho_$$
rm -rf "$OUT" "$O"; mkdir -p "$OUT" "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
for v in exact exact_noprune fast fast_noprune; do
  unset AMHIP_ORTHO_EXACT_FOLD AMHIP_ORTHO_FAST_WAVES AMHIP_ORTHO_NO_PRUNE
  case $v in
    exact) export AMHIP_ORTHO_EXACT_FOLD=1;;
    exact_noprune) export AMHIP_ORTHO_EXACT_FOLD=1 AMHIP_ORTHO_NO_PRUNE=1;;
    fast) ;;
    fast_noprune) export AMHIP_ORTHO_NO_PRUNE=1;;
  esac
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$O/$v-a" -o s -- $B > /dev/null 2> "$O/$v-a.err"
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/$v-b" -o s -- $B > /dev/null 2> "$O/$v-b.err"
  python "$R/tools/rocprof_summary.py" --sq "$O/$v-a/s_results.db" "$O/$v-b/s_results.db" --sq-json "$OUT/sq_$v.json" --tag "$v" > "$OUT/sq_$v.log" 2>&1
done
ls -la "$OUT"
