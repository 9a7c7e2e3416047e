#!/bin/bash
# End-of-round verification: full GPU test suite, smoke(), the default bench line, then the committed profiles
# (rocprofv3 kernel stats of the bench + PMC passes on the product GEMM).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -15 > gpurun_out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/bench.log 2>&1
bash scripts/gpu_profile.sh > gpurun_out/profile_driver.log 2>&1
tail -5 gpurun_out/tests.log; tail -3 gpurun_out/smoke.log; grep '^{' gpurun_out/bench.log | cut -c1-700
