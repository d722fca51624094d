#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_run3; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_dsm_fast.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_reference_loops.py tests/test_gpu_fullsize.py tests/test_gpu_tiling.py tests/test_gpu_cpp_shim.py -q -rf --no-header -x > "$OUT/pytest.txt" 2>&1
tail -25 "$OUT/pytest.txt"
bash tools/collect_profiles.sh v1 r02 > "$OUT/collect.txt" 2>&1
tail -5 "$OUT/collect.txt"
