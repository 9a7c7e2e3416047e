#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/tests.log
timeout 600 python scripts/gelu_ab.py 9 > gpurun_out/gelu_ab.log 2>&1
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/bench.log 2>&1
timeout 600 python scripts/graph_latency.py 1 2 4 8 16 > gpurun_out/graph_latency.log 2>&1
tail -6 gpurun_out/tests.log; cat gpurun_out/gelu_ab.log; grep '^{' gpurun_out/bench.log | cut -c1-1400; tail -7 gpurun_out/graph_latency.log
