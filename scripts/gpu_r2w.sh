#!/bin/bash
# Round-2 run W: distributed mixer at 10 / 5 / 2 workgroups per crop — pipeline + model tests, head A/B over the batch sizes, bench line
mkdir -p gpurun_out/r2w
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2w/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2w/pytest_gpu.log
tail -4 gpurun_out/r2w/pytest_gpu.log
timeout 300 python scripts/head_bench.py 16 25 26 32 51 52 64 128 > gpurun_out/r2w/head_bench.log 2>&1; grep "B=" gpurun_out/r2w/head_bench.log
THMR_MIXER_CLUSTER=0 timeout 300 python scripts/head_bench.py 32 64 128 > gpurun_out/r2w/head_bench_nocluster.log 2>&1; grep "B=" gpurun_out/r2w/head_bench_nocluster.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2w/bench.json 2> gpurun_out/r2w/bench.err; python - <<EOF
import json
d=json.loads(open("gpurun_out/r2w/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["classes_ms_per_step"], d["parity"]["mismatches"])
EOF
timeout 600 python bench.py --no-cpu-baseline --batch 32 > gpurun_out/r2w/bench_b32.json 2> gpurun_out/r2w/bench_b32.err; python - <<EOF
import json
d=json.loads(open("gpurun_out/r2w/bench_b32.json").read().strip().split("\n")[-1])
print("B=32:", d["value"], d["ms_per_step"], d["roofline"]["classes_ms_per_step"])
EOF
