#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -14) > gpurun_out/r2d_pytest.log
timeout 300 python scripts/head_bench.py 1 8 64 > gpurun_out/r2d_head.log 2>&1
THMR_LEGACY_HEAD=1 timeout 300 python scripts/head_bench.py 1 8 64 >> gpurun_out/r2d_head.log 2>&1
rm -rf gpurun_out/r2d_trace; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r2d_trace" -o t -- python "$R/scripts/head_bench.py" 64) > gpurun_out/r2d_trace.log 2>&1
tail -6 gpurun_out/r2d_pytest.log; grep "B=" gpurun_out/r2d_head.log
f=$(find gpurun_out/r2d_trace -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-150
