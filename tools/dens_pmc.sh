#!/bin/bash
# SQ counters of the gather kernels on the density probe, for two builds of the library
cd /tmp && export TMPDIR=/tmp
export AMHIP_PROBE_DENSITIES=$1
shift
for lib in "$@"; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
    rm -rf /tmp/pq
    AMHIP_LIB_PATH=$GRAFT_REPO_ROOT/aerial_mapper_amd/lib/$lib timeout 400 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pq -o s -- python $GRAFT_REPO_ROOT/tools/density_probe.py > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocprof_summary.py --sq /tmp/pq/s_results.db --sq-json /tmp/pq/o.json --tag x > /dev/null 2>&1
    python - "$lib" <<'P'
import json, sys
d = json.load(open("/tmp/pq/o.json"))
for k, v in d["kernels"].items():
    if "f32_wide" in k or "gather_f32<" in k:
        print(sys.argv[1], k[:40], {a: round(b / 1e6, 2) for a, b in v.items()})
P
  done
done
