#!/bin/bash
# Round-2 run Q: GEMM co-resident block de-phasing sweep; B = 1/2/4 call latency with the current build
mkdir -p gpurun_out/r2q
timeout 400 build_ab/gemm_dephase > gpurun_out/r2q/gemm_dephase.log 2>&1; echo "rc=$?" >> gpurun_out/r2q/gemm_dephase.log
cat gpurun_out/r2q/gemm_dephase.log
timeout 300 python scripts/graph_latency.py > gpurun_out/r2q/graph_latency.log 2>&1; tail -5 gpurun_out/r2q/graph_latency.log
