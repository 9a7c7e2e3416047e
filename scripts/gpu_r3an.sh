#!/bin/bash
# Round 3, session AN: low regime of the split3 mode (3 and 4 crops: proj / fc2 split K four ways), mid regime from 5 crops: parity, then 3 ... 6 crops
set -u
O=gpurun_out/r3an; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "split3" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "split3 B\|passed\|failed\|Error\|error\|assert" $O/tests.log | tail -8
for B in 3 4 5 6; do timeout 300 python scripts/mode_bench.py $B 20 2>/dev/null | grep -a '"mode"' | head -2 | cut -c1-330 >> $O/mode_bench_low.log; done
cat $O/mode_bench_low.log
echo "total t=$(( $(date +%s) - t0 ))"
