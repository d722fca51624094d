#!/bin/bash
# Collect the round's rocprofv3 evidence ON the GPU box, for BOTH arithmetic modes of the DSM
# gather, and leave only the small summaries in gpurun_out/profiles_out/ (the rocpd databases
# exceed what gpurun copies back).  Usage (from the repo root, through gpurun):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r05 [modes]'
# modes: "exact fast" (default; exact = the library's default = bench.py's headline) or one of them.
# Per mode M:  <round>_M_cfg3_rocprofv3.md          kernel trace + FETCH_SIZE / WRITE_SIZE passes
#              <round>_M_pmc_traffic.json           HBM bytes per bench step and kernel slot
#              <round>_M_cfg3_pmc_sq.json           SQ / GRBM counters per kernel
#              <round>_M_cfg3_bench_under_rocprof.json   the bench line of the traced command
# bench.py reads the two JSONs of ITS OWN mode only (file name + "dsm_mode" key).
# Counter passes run on their own, with --kernel-trace only (gpurun refuses --pmc next to
# --sys-trace / hip / hsa traces).
set -u
RND=${1:-r05}
MODES=${2:-"exact fast"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=/tmp/amhip_prof_$$
OUT=$R/gpurun_out/profiles_out
mkdir -p "$O" "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 python "$R/bench.py" --steps 10 --warmup 3 > "$OUT/${RND}_bench_cfg3_n1.json" 2> "$O/bench.err"
for MODE in $MODES; do
  B="python $R/bench.py --steps 5 --warmup 2 --dsm-mode $MODE --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain"
  P="${RND}_${MODE}"
  # (the trace pass over more steps: the first launches of a process run 10 - 20 % slower, and the
  # average of 7 would not agree with bench.py's live HIP-event time of the same kernel)
  BT="python $R/bench.py --steps 30 --warmup 5 --dsm-mode $MODE --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- $BT > "$OUT/${P}_cfg3_bench_under_rocprof.json" 2> "$O/trace.err"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" -o f -- $B > /dev/null 2> "$O/fetch.err"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/write" -o w -- $B > /dev/null 2> "$O/write.err"
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$O/sq1" -o s -- $B > /dev/null 2> "$O/sq1.err"
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/sq2" -o s -- $B > /dev/null 2> "$O/sq2.err"
  python "$R/tools/rocprof_summary.py" --trace "$O/trace/t_results.db" --fetch "$O/fetch/f_results.db" \
    --write "$O/write/w_results.db" --dsm-mode "$MODE" \
    --title "$RND, --dsm-mode $MODE: python bench.py --steps 30 --warmup 5 (kernel trace; the counter passes: --steps 5 --warmup 2) --dsm-mode $MODE --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain under rocprofv3, cfg3" \
    -o "$OUT/${P}_cfg3_rocprofv3.md" --traffic-json "$OUT/${P}_pmc_traffic.json" \
    --note "$RND kernels, --dsm-mode $MODE; see profiles/${P}_cfg3_rocprofv3.md." > /dev/null
  python "$R/tools/rocprof_summary.py" --sq "$O/sq1/s_results.db" "$O/sq2/s_results.db" --dsm-mode "$MODE" \
    --sq-json "$OUT/${P}_cfg3_pmc_sq.json" --tag "$RND $MODE"
  rm -rf "$O/trace" "$O/fetch" "$O/write" "$O/sq1" "$O/sq2"
done
tail -3 "$O"/*.err 2>/dev/null | tail -20
rm -rf "$O"
ls -la "$OUT"
