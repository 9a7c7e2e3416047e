#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/tests.log
GEMM_VARIANTS=auto timeout 600 python scripts/small_batch_prof.py 1 2 3 4 5 8 2>&1 | grep "^B " > gpurun_out/small_batch.log
timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -8 gpurun_out/tests.log; cat gpurun_out/small_batch.log; grep '^{' gpurun_out/bench.log | cut -c1-300
