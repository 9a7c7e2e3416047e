#!/bin/bash
# Committed-profile run: kernel stats of the default bench + PMC passes (SQ, FETCH, WRITE) on the product GEMM.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"
cd "$R"
rm -rf gpurun_out/prof3 gpurun_out/pmc3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof3" -o r1 -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/prof3.log 2>&1
find gpurun_out/prof3 -type f -name "*kernel_trace*" -delete
V=128x160
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/gpurun_out/pmc3/sq" -o p -- python "$R/scripts/gemm_bench.py" 2) > gpurun_out/pmc3.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc3/fetch" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/pmc3.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc3/write" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/pmc3.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/gpurun_out/pmc3/l2" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/pmc3.log 2>&1
find gpurun_out -type f -size +6M -delete
ls gpurun_out/prof3 gpurun_out/pmc3/*
