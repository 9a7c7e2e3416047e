#!/bin/bash
# Round-2 run U: tiny-M GEMM kernel for the VQ decoder in the small-batch regime — full GPU suite, head A/B, call latency
mkdir -p gpurun_out/r2u
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2u/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2u/pytest_gpu.log
tail -6 gpurun_out/r2u/pytest_gpu.log
timeout 300 python scripts/head_bench.py 1 2 4 6 7 8 > gpurun_out/r2u/head_bench.log 2>&1; grep "B=" gpurun_out/r2u/head_bench.log
THMR_TINY_GEMM=0 timeout 300 python scripts/head_bench.py 1 2 4 6 > gpurun_out/r2u/head_bench_notiny.log 2>&1; grep "B=" gpurun_out/r2u/head_bench_notiny.log
timeout 300 python scripts/graph_latency.py 1 2 4 6 > gpurun_out/r2u/graph_latency.log 2>&1; tail -1 gpurun_out/r2u/graph_latency.log
