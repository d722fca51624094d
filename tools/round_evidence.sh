#!/bin/bash
# a round's final evidence in one GPU call: the GPU suite, the default bench line (the file the
# driver's BENCH run should reproduce) and the whole-map parity record (every one of the 1e8 cells
# against the reference's own compiled process() calls, both gather modes).  Usage: round_evidence.sh r04
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_evidence
RND=${1:-r04}
export RND
mkdir -p "$OUT"
cd "$R"
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > "$OUT/gpu_tests.log"
tail -4 "$OUT/gpu_tests.log"
timeout 600 python bench.py > "$OUT/${RND}_bench_cfg3_n1_evidence.json" 2> "$OUT/bench.err"
timeout 1200 python bench.py --steps 10 --warmup 3 --cpu-sample-side 10000 --no-host-path --no-rough-terrain \
  > "$OUT/${RND}_bench_cfg3_full_parity.json" 2> "$OUT/full.err"
python - <<'P'
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/round_evidence/"
rnd = os.environ.get("RND", "r04")
for f in (rnd + "_bench_cfg3_n1_evidence.json", rnd + "_bench_cfg3_full_parity.json"):
    try:
        d = json.loads(open(o + f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
        print("  fast", d.get("fast_mode", {}).get("ms_per_step"), "traffic", d["roofline"].get("traffic"),
              d["roofline"].get("traffic_rejected"))
        for k in ("parity_sample", "parity_full", "cpu_baseline"):
            if k in d:
                print("  ", k, json.dumps(d[k])[:700])
        if "fast_mode" in d:
            print("   fast parity", json.dumps(d["fast_mode"].get("parity_sample"))[:500])
    except Exception as e:
        print(f, "unreadable:", e)
P
