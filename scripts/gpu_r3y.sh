#!/bin/bash
# Round 3, session Y: small-batch regime of the split3 mode (ring kernel on split3 operands): parity, then one / two / four / six crops in both modes
set -u
O=gpurun_out/r3y; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -s -k "split3" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "split3 B\|passed\|failed\|Error\|error\|assert" $O/tests.log | tail -20
for B in 1 2 4 6; do timeout 300 python scripts/mode_bench.py $B 30 2>/dev/null | grep -a '"mode"\|max_abs' | cut -c1-420 >> $O/mode_bench_small.log; done
cat $O/mode_bench_small.log
echo "total t=$(( $(date +%s) - t0 ))"
