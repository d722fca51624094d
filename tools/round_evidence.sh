#!/bin/bash
# One GPU call for a round's evidence on the FINAL build (copy gpurun_out/profiles_out/* and the logs
# named below into profiles/ afterwards):
#   gpurun --timeout 3000 -- 'bash tools/round_evidence.sh r06'
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_out
mkdir -p "$OUT"
cd "$R"
python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$OUT/${RND}_gpu_tests.log"
python -c "import __graft_entry__ as g; g.smoke()" >> "$OUT/${RND}_gpu_tests.log" 2>&1
bash tools/collect_profiles.sh "$RND" > "$OUT/${RND}_collect.log" 2>&1
timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 3 > "$OUT/${RND}_bench_cfg1.json" 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --map-origin 464980.25,5272690.5 --no-cpu-baseline --no-host-path > "$OUT/${RND}_bench_cfg3_utm.json" 2>/dev/null
timeout 300 python tools/bench_io.py > "$OUT/${RND}_bench_io.json" 2>/dev/null
timeout 300 python tools/bench_forward.py > "$OUT/${RND}_bench_forward.json" 2>/dev/null
timeout 1500 python tools/soak.py 5000 400 > "$OUT/${RND}_soak.log" 2>&1
tail -3 "$OUT/${RND}_soak.log"
cat "$OUT/${RND}_gpu_tests.log"
ls -la "$OUT"
