#!/bin/bash
# Round-2 run O: persistent attention kernel + XCD-aware block order — micro A/B, attention / model parity tests, bench line
mkdir -p gpurun_out/r2o
timeout 300 build_ab/attn_timeline > gpurun_out/r2o/attn_timeline.log 2>&1; echo "rc=$?" >> gpurun_out/r2o/attn_timeline.log
grep -v "^dephase" gpurun_out/r2o/attn_timeline.log | tail -25
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2o/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o/pytest_gpu.log
tail -5 gpurun_out/r2o/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r2o/bench.json 2> gpurun_out/r2o/bench.err; tail -c 2500 gpurun_out/r2o/bench.json; tail -3 gpurun_out/r2o/bench.err
THMR_ATTN_VARIANT=3 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2o/bench_plain_attn.json 2> gpurun_out/r2o/bench_plain_attn.err; tail -c 600 gpurun_out/r2o/bench_plain_attn.json
