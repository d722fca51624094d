cd /tmp && export TMPDIR=/tmp
export AMHIP_PROBE_DENSITIES=$1
for lib in lab/libold.so libaerial_mapper_hip.so; do
  rm -rf /tmp/pp
  AMHIP_LIB_PATH=$GRAFT_REPO_ROOT/aerial_mapper_amd/lib/$lib timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/density_probe.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py --trace /tmp/pp/t_results.db --title t -o /tmp/pp/s.md > /dev/null 2>&1
  echo "== $lib density $1"; grep "k_dsm_gather\|k_dsm_tile" /tmp/pp/s.md | head -12
done
