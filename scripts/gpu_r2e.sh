#!/bin/bash
# round 2, run E: final fused head vs the round-1 launch chain (A/B), launch counts, pipeline tests with two engines
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_tokenizer.py -m gpu -q -x 2>&1 | tail -4) > gpurun_out/r2e_pytest.log
timeout 300 python scripts/head_bench.py 1 2 4 8 16 32 64 > gpurun_out/r2e_head.log 2>&1
THMR_LEGACY_HEAD=1 timeout 300 python scripts/head_bench.py 1 2 4 8 16 32 64 >> gpurun_out/r2e_head.log 2>&1
timeout 300 python scripts/head_bench.py 512 >> gpurun_out/r2e_head.log 2>&1
THMR_LEGACY_HEAD=1 timeout 300 python scripts/head_bench.py 512 >> gpurun_out/r2e_head.log 2>&1
rm -rf gpurun_out/r2e_trace; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r2e_trace" -o t -- python "$R/scripts/head_bench.py" 64) > gpurun_out/r2e_trace.log 2>&1
timeout 600 python scripts/graph_latency.py 1 2 4 > gpurun_out/r2e_latency.log 2>&1
tail -4 gpurun_out/r2e_pytest.log; grep "B=" gpurun_out/r2e_head.log | grep -v timeline; tail -1 gpurun_out/r2e_latency.log
f=$(find gpurun_out/r2e_trace -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-160
