#!/bin/bash
# same-box A-B of library builds on the density probe (single-precision rows only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export AMHIP_PROBE_DENSITIES=${1:-0.5,1,2,4,8,16}
shift
for lib in "$@"; do
  echo "== $lib"
  AMHIP_LIB_PATH=$R/aerial_mapper_amd/lib/$lib timeout 600 python tools/density_probe.py 2>&1 | grep "pts/cell"
done
