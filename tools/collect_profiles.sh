#!/bin/bash
# Collect the round's rocprofv3 evidence ON the GPU box and leave only the small
# summaries in gpurun_out/profiles_out/ (the rocpd databases exceed what gpurun
# copies back).  Usage (from the repo root, through gpurun):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh v1 r02'
# A third argument "exact" profiles the library's DEFAULT mode (FP64 gather) instead of the
# bench's headline mode: kernel trace + traffic only, files tagged <round>_<tag>_exact_*.
set -u
TAG=${1:-vX}
RND=${2:-r03}
MODE=${3:-fast}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=/tmp/amhip_prof_$$
OUT=$R/gpurun_out/profiles_out
rm -rf "$O" "$OUT"; mkdir -p "$O" "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = exact ]; then
  B="python $R/bench.py --steps 5 --warmup 2 --dsm-mode exact --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- $B > "$OUT/${RND}_${TAG}_exact_cfg3_bench_under_rocprof.json" 2> "$O/trace.err"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" -o f -- $B > /dev/null 2> "$O/fetch.err"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/write" -o w -- $B > /dev/null 2> "$O/write.err"
  python "$R/tools/rocprof_summary.py" --trace "$O/trace/t_results.db" --fetch "$O/fetch/f_results.db" \
    --write "$O/write/w_results.db" \
    --title "$RND $TAG, DEFAULT mode (FP64 gather): python bench.py --dsm-mode exact --steps 5 --warmup 2 under rocprofv3, cfg3" \
    -o "$OUT/${RND}_${TAG}_exact_cfg3_rocprofv3.md" --traffic-json "$OUT/${RND}_${TAG}_exact_pmc_traffic_not_for_bench.json" \
    --note "$RND $TAG kernels, FP64 mode." > /dev/null
  rm -rf "$O"; ls -la "$OUT"; exit 0
fi
timeout 900 python "$R/bench.py" --steps 10 --warmup 3 > "$OUT/${RND}_bench_cfg3_n1.json" 2> "$O/bench.err"
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- $B > "$OUT/${RND}_${TAG}_cfg3_bench_under_rocprof.json" 2> "$O/trace.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" -o f -- $B > /dev/null 2> "$O/fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/write" -o w -- $B > /dev/null 2> "$O/write.err"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$O/sq1" -o s -- $B > /dev/null 2> "$O/sq1.err"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/sq2" -o s -- $B > /dev/null 2> "$O/sq2.err"
python "$R/tools/rocprof_summary.py" --trace "$O/trace/t_results.db" --fetch "$O/fetch/f_results.db" \
  --write "$O/write/w_results.db" \
  --title "$RND $TAG: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path under rocprofv3, cfg3" \
  -o "$OUT/${RND}_${TAG}_cfg3_rocprofv3.md" --traffic-json "$OUT/${RND}_pmc_traffic.json" \
  --note "$RND $TAG kernels; see profiles/${RND}_${TAG}_cfg3_rocprofv3.md." > /dev/null
python "$R/tools/rocprof_summary.py" --sq "$O/sq1/s_results.db" "$O/sq2/s_results.db" \
  --sq-json "$OUT/${RND}_${TAG}_cfg3_pmc_sq.json" --tag "$TAG"
rm -rf "$O"
ls -la "$OUT"
