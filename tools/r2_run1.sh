#!/bin/bash
# round 2, first GPU call: instruction rates, DSM parity in both modes, cfg2 fast vs exact
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2_run1
mkdir -p "$OUT"
cd "$R"
timeout 120 tools/ubench/ubench > "$OUT/ubench.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_reference_loops.py tests/test_gpu_fullsize.py -q -k "dsm or golden or fullsize" -rf --no-header > "$OUT/pytest_dsm.txt" 2>&1
tail -15 "$OUT/pytest_dsm.txt"
for mode in fast exact; do
  if [ $mode = exact ]; then export AMHIP_DSM_EXACT=1; else unset AMHIP_DSM_EXACT; fi
  timeout 600 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-host-path --cpu-sample-side 3000 > "$OUT/bench_cfg2_$mode.json" 2> "$OUT/bench_cfg2_$mode.err"
  python - "$OUT/bench_cfg2_$mode.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()}, d.get("parity_sample"))
PY
done
unset AMHIP_DSM_EXACT
cd /tmp && export TMPDIR=/tmp
O=/tmp/prof_$$; mkdir -p $O
B="python $R/bench.py --workload cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- $B > "$OUT/bench_under_rocprof.json" 2> "$O/trace.err"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$O/sq1" -o s -- $B > /dev/null 2> "$O/sq1.err"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/sq2" -o s -- $B > /dev/null 2> "$O/sq2.err"
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d "$O/sq3" -o s -- $B > /dev/null 2> "$O/sq3.err"
python "$R/tools/rocprof_summary.py" --trace "$O/trace/t_results.db" --title "r02 run1 cfg2 fast" -o "$OUT/cfg2_rocprofv3.md" > /dev/null 2> "$OUT/summary.err"
python "$R/tools/rocprof_summary.py" --sq "$O/sq1/s_results.db" "$O/sq2/s_results.db" "$O/sq3/s_results.db" --sq-json "$OUT/cfg2_pmc_sq.json" --tag run1 >> "$OUT/summary.err" 2>&1
cat "$OUT/cfg2_rocprofv3.md" | head -40
ls -la "$OUT"
