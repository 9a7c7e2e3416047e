#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-$PWD}"
GEMM_EPI_NONE=1 GEMM_VARIANTS=128x128,abl11,abl12,abl13,abl17,128x160,abl21,abl23,abl27 timeout 600 python scripts/gemm_bench.py 5 > gpurun_out/gemm_abl.log 2>&1
cat gpurun_out/gemm_abl.log | tail -6
