#!/bin/bash
# A/B of a compile-time constant on the GPU box: rebuilds the library there.
#   gpurun -- 'bash tools/ab_tile.sh FILE "PATTERN" VALUE...'   (PATTERN contains @V@)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
F=$1; PAT=$2; shift 2
cp "$F" /tmp/ab_tile_orig
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path"
for v in "$@"; do
  cp /tmp/ab_tile_orig "$F"
  python - "$F" "$PAT" "$v" <<'PY'
import re, sys
f, pat, v = sys.argv[1:4]
s = open(f).read()
rx = re.escape(pat).replace("@V@", r"[-\w.]+")
new, n = re.subn(rx, pat.replace("@V@", v), s)
assert n == 1, (n, rx)
open(f, "w").write(new)
PY
  python -m aerial_mapper_amd.build > /tmp/ab_build.log 2>&1 || { echo "build failed for $v"; tail -5 /tmp/ab_build.log; continue; }
  timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms/step', d['ms_per_step'], {k:round(x['ms_per_step'],3) for k,x in d['kernels'].items()})"
done
cp /tmp/ab_tile_orig "$F"
