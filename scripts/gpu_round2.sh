#!/bin/bash
# GPU round 2: tests, GEMM variant sweep, bench, kernel stats (csv), PMC counters for the GEMM.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu --tb=short -s 2>&1 | tail -40 > gpurun_out/tests.log
timeout 600 python scripts/gemm_bench.py 5 > gpurun_out/gemm_bench.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof2 gpurun_out/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof2" -o r1 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/prof2.log 2>&1
find gpurun_out/prof2 -type f -name "*kernel_trace*" -delete
for V in 128x128 128x160s3; do
  (cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/gpurun_out/pmc/$V/sq" -o p -- python "$R/scripts/gemm_bench.py" 1) > gpurun_out/pmc_$V.log 2>&1
  (cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc/$V/fetch" -o p -- python "$R/scripts/gemm_bench.py" 1) >> gpurun_out/pmc_$V.log 2>&1
  (cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc/$V/write" -o p -- python "$R/scripts/gemm_bench.py" 1) >> gpurun_out/pmc_$V.log 2>&1
done
find gpurun_out -type f -size +6M -delete
ls -R gpurun_out | head -50
