#!/bin/bash
# Round 3, session Z: hardware bf16 conversion in the split3 producers: parity (bit-identical to the numpy restatement), both modes at 64 crops, bench.py --vit-gemm split3
set -u
O=gpurun_out/r3z; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -s -k "split3 or b64_tokens" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "golden\|split3 B\|passed\|failed\|Error\|error\|assert" $O/tests.log | tail -20
timeout 300 python scripts/mode_bench.py 64 2>/dev/null | grep -a '"mode"\|max_abs' | cut -c1-420 > $O/mode_bench.log; cat $O/mode_bench.log
timeout 600 python bench.py --vit-gemm split3 --no-cpu-baseline > $O/bench_split3.json 2> $O/bench_split3.err; echo "bench rc=$?"; cut -c1-700 $O/bench_split3.json
echo "total t=$(( $(date +%s) - t0 ))"
