#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -k "ring or gelu" 2>&1 | tail -25 > gpurun_out/tests_ring.log
SKIP_PATH=1 timeout 900 python scripts/small_batch_prof.py 1 2 4 8 16 > gpurun_out/small_gemm.log 2>&1
tail -8 gpurun_out/tests_ring.log; cat gpurun_out/small_gemm.log
