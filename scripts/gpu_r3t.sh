#!/bin/bash
# Round 3, session T: fused fc1 -> split3 epilogue: op parity, engine parity, both modes at 64 crops, smaller batches with the mode forced down
set -u
O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -s -k "split3 or b64_tokens" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "golden\|split3 B\|passed\|failed\|Error\|error" $O/tests.log | tail -20
timeout 600 python scripts/mode_bench.py 64 > $O/mode_bench_b64.log 2>&1; echo "bench rc=$?"; grep -a mode $O/mode_bench_b64.log | cut -c1-400
for B in 32 24 17; do timeout 300 python scripts/mode_bench.py $B 2>/dev/null | grep -a '"mode"' | cut -c1-120 >> $O/mode_bench_mid.log; done
for B in 16 12 8; do THMR_SPLIT3_MIN_B=7 timeout 300 python scripts/mode_bench.py $B 2>/dev/null | grep -a '"mode"' | cut -c1-120 >> $O/mode_bench_mid.log; done
cat $O/mode_bench_mid.log
echo "total t=$(( $(date +%s) - t0 ))"
