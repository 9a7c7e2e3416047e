#!/bin/bash
# Round-2 run M: validate the resolved-once head weight pointers (HotW) — full GPU suite + bench line + B=1 latency.
set -x
mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2m/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest_gpu.log
tail -3 gpurun_out/r2m/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err; tail -c 1500 gpurun_out/r2m/bench.json
timeout 300 python scripts/head_bench.py > gpurun_out/r2m/head_bench.log 2>&1; tail -15 gpurun_out/r2m/head_bench.log
