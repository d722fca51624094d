#!/bin/bash
# A-B of the near-centre guard's distance (lab build with the knob) on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for th in 0.02 0.01 0.005; do
  AMHIP_LIB_PATH=$R/aerial_mapper_amd/lib/lab/libaerial_mapper_hip.so AMHIP_FX_THETA=$th timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('theta $th', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done
