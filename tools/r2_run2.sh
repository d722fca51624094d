#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_run2; mkdir -p "$OUT"; cd "$R"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-host-path --no-cpu-baseline > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("%-22s step %.3f  gather %.4f" % (sys.argv[2], d["ms_per_step"], d["kernels"]["k_dsm_gather"]["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for v in "$@"; do
  case $v in
    base) run base X=1;;
    var*) run $v AMHIP_F32_VARIANT=${v#var};;
    exact) run exact AMHIP_DSM_EXACT=1;;
  esac
done
