#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu --tb=short -s 2>&1 | tail -30 > gpurun_out/tests.log
timeout 600 python scripts/gemm_bench.py 5 > gpurun_out/gemm_bench.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
tail -8 gpurun_out/tests.log; grep -v amdgpu gpurun_out/gemm_bench.log | head -5; grep '^{' gpurun_out/bench.log | cut -c1-1500
