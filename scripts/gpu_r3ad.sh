#!/bin/bash
# Round 3, session AD: mid regime of the split3 mode (7 ... 15 crops, proj / fc2 split K two ways): parity, then 7 / 8 / 10 / 12 / 15 crops in both modes
set -u
O=gpurun_out/r3ad; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -s -k "split3" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "split3 B\|passed\|failed\|Error\|error\|assert" $O/tests.log | tail -12
for B in 7 8 10 12 15; do timeout 300 python scripts/mode_bench.py $B 20 2>/dev/null | grep -a '"mode"\|max_abs' | cut -c1-330 >> $O/mode_bench_mid.log; done
cat $O/mode_bench_mid.log
echo "total t=$(( $(date +%s) - t0 ))"
