#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -x 2>&1 | tail -12 > gpurun_out/tests.log
timeout 600 python scripts/gemm_bench.py 9 > gpurun_out/gemm_bench.log 2>&1
tail -5 gpurun_out/tests.log; grep -v amdgpu gpurun_out/gemm_bench.log | head -5
