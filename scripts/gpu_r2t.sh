#!/bin/bash
# Round-2 run T: the MLP-Mixer stack distributed over ten workgroups per crop inside the persistent decoder kernel (B <= 25)
mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2t/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2t/pytest_gpu.log
tail -5 gpurun_out/r2t/pytest_gpu.log
timeout 300 python scripts/head_bench.py 1 2 4 8 16 25 26 64 > gpurun_out/r2t/head_bench.log 2>&1; grep "B=" gpurun_out/r2t/head_bench.log
THMR_MIXER_CLUSTER=0 timeout 300 python scripts/head_bench.py 1 2 4 8 16 25 > gpurun_out/r2t/head_bench_nocluster.log 2>&1; grep "B=" gpurun_out/r2t/head_bench_nocluster.log
timeout 300 python scripts/graph_latency.py 1 2 4 8 > gpurun_out/r2t/graph_latency.log 2>&1; tail -1 gpurun_out/r2t/graph_latency.log
