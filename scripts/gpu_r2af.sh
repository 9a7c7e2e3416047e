#!/bin/bash
# Round-2 run AF: W fragments of the next GEMV stage requested under the grid barrier (persistent decoder kernel)
mkdir -p gpurun_out/r2af
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2af/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2af/pytest_gpu.log
tail -3 gpurun_out/r2af/pytest_gpu.log
THMR_DEC_TIMELINE=1 timeout 300 python scripts/head_bench.py 1 8 16 32 64 > gpurun_out/r2af/head_bench.log 2>&1; grep "B=" gpurun_out/r2af/head_bench.log | cut -c1-700
