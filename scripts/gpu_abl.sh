#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-$PWD}"
GEMM_EPI_NONE=1 GEMM_VARIANTS=128x160,ds1,ds4,bar2,bar10,bar20,abl37 timeout 600 python scripts/gemm_bench.py 7 > gpurun_out/gemm_abl2.log 2>&1
grep -v amdgpu gpurun_out/gemm_abl2.log | head -5
