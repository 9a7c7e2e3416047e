#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/tests.log
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/bench.log 2>&1
tail -6 gpurun_out/tests.log; grep '^{' gpurun_out/bench.log | cut -c1-1400
