#!/bin/bash
# Round 3, session X: attention writes the proj GEMM's split3 operand directly: op parity, engine parity, both modes at 64 / 32 / 16 crops
set -u
O=gpurun_out/r3x; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -s -k "split3 or b64_tokens" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "golden\|split3 B\|passed\|failed\|Error\|error" $O/tests.log | tail -20
for B in 64 32 16; do timeout 300 python scripts/mode_bench.py $B 2>/dev/null | grep -a '"mode"\|max_abs' | cut -c1-420 >> $O/mode_bench.log; done
cat $O/mode_bench.log
echo "total t=$(( $(date +%s) - t0 ))"
