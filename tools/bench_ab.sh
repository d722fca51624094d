#!/bin/bash
# same-box A-B of library builds on the bench's workload (kernel times of the headline mode)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
for lib in "$@"; do
  AMHIP_LIB_PATH=$R/aerial_mapper_amd/lib/$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-second-mode --no-rough-terrain 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done; done
