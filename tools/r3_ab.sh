#!/bin/bash
# same-box A-B: the doubles pipeline (lab build of the previous commit, near-centre guard
# distance 0.02 / 0.005 through its knob) against the working tree's library
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
}
for rep in 1 2; do
  AMHIP_LIB_PATH=$R/aerial_mapper_amd/lib/lab/libold.so AMHIP_FX_THETA=0.02 run "old theta 0.02"
  AMHIP_LIB_PATH=$R/aerial_mapper_amd/lib/lab/libold.so AMHIP_FX_THETA=0.005 run "old theta 0.005"
  run "tree"
done
