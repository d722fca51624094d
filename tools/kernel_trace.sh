#!/bin/bash
# Per-kernel times of one bench.py run under rocprofv3 --kernel-trace (summary on stdout).
#   gpurun -- 'bash tools/kernel_trace.sh [bench.py args]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=/tmp/amhip_kt_$$
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain "$@" > "$O/bench.json" 2> "$O/err.txt"
python "$R/tools/rocprof_summary.py" --trace "$O/trace/t_results.db" --title "kernel trace" -o "$O/out.md" > /dev/null 2> "$O/sum.err" || { tail -5 "$O/err.txt" "$O/sum.err"; ls -R "$O" | head -20; }
grep -E "^\|" "$O/out.md" | head -24 | cut -c1-200
rm -rf "$O"
