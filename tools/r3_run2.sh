#!/bin/bash
# round 3, GPU call 2: stream-overlap probe, the rough-terrain extra, the whole-map parity record
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_run2
mkdir -p "$OUT"
cd "$R"
timeout 120 tools/ubench/overlap_probe > "$OUT/overlap_probe.jsonl" 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-host-path > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
timeout 1200 python bench.py --steps 5 --warmup 2 --no-host-path --no-rough-terrain --cpu-sample-side 10000 > "$OUT/bench_cfg3_full_parity.json" 2> "$OUT/full_parity.err"
cat "$OUT/overlap_probe.jsonl"; python - <<'P'
import json,os
o=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_run2/"
for f in ("bench_cfg3.json","bench_cfg3_full_parity.json"):
    try:
        d=json.load(open(o+f))
        print(f, d["ms_per_step"], json.dumps(d.get("rough_terrain")), json.dumps(d.get("parity_sample")), json.dumps(d.get("exact_mode",{}).get("parity_sample")), d.get("cpu_baseline",{}).get("sample"))
    except Exception as e: print(f, "ERR", e)
P
