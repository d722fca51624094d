#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_run4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o r -- python "$R/tools/rough_probe.py" > "$OUT/rough.log" 2>&1
python "$R/tools/rocprof_summary.py" --trace /tmp/rp/r_results.db -o "$OUT/rough_rocprof.md" > /dev/null 2>&1
cat "$OUT/rough.log" | tail -3; head -30 "$OUT/rough_rocprof.md"
