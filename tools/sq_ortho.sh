#!/bin/bash
# SQ instruction counters of the backward-grid kernel builds (one gpurun call).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_ortho
O=/tmp/sq_ortho_$$
rm -rf "$OUT" "$O"; mkdir -p "$OUT" "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
for v in stop3 stop4 stop5; do
  unset AMHIP_ORTHO_EXACT_FOLD AMHIP_ORTHO_FAST_WAVES AMHIP_ORTHO_NO_PRUNE AMHIP_ORTHO_STOP
  case $v in
    stop3) export AMHIP_ORTHO_STOP=3;;
    stop4) export AMHIP_ORTHO_STOP=4;;
    stop5) export AMHIP_ORTHO_STOP=5;;
  esac
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$O/$v-a" -o s -- $B > /dev/null 2> "$O/$v-a.err"
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/$v-b" -o s -- $B > /dev/null 2> "$O/$v-b.err"
  python "$R/tools/rocprof_summary.py" --sq "$O/$v-a/s_results.db" "$O/$v-b/s_results.db" --sq-json "$OUT/sq_$v.json" --tag "$v" > "$OUT/sq_$v.log" 2>&1
done
ls -la "$OUT"
