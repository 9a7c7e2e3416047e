#!/bin/bash
# Round 3, session AG: fc2 of the split3 mode split K two ways at every batch size: parity, then 16 / 32 / 48 / 64 crops
set -u
O=gpurun_out/r3ag; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "split3 or b64_tokens" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "golden\|split3 B\|passed\|failed\|Error\|error\|assert" $O/tests.log | tail -12
for B in 16 32 48 64; do timeout 300 python scripts/mode_bench.py $B 10 2>/dev/null | grep -a '"mode": "split3"' | cut -c1-330 >> $O/mode_bench.log; done
cat $O/mode_bench.log
echo "total t=$(( $(date +%s) - t0 ))"
