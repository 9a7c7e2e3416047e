#!/bin/bash
# Round 3, session L: 16x16x4 small-M GEMM for qkv at one and two crops — parity, A/B of the call, kernel stats
set -u
O=gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
for q in 1 0 1 0; do
  echo "== THMR_QKV_RING16=$q" >> $O/ring16_ab.log
  THMR_QKV_RING16=$q timeout 300 python scripts/mid_split_sweep.py 2 1 2 1 2>/dev/null | grep '^{' >> $O/ring16_ab.log
done
cat $O/ring16_ab.log | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_b1" -o p -- python "$R/scripts/b1_trace.py" 1) > $O/prof_b1.log 2>&1
find $O/prof_b1 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
grep -i "ring" $O/prof_b1/p_kernel_stats.csv 2>/dev/null | cut -c1-200
echo "total t=$(( $(date +%s) - t0 ))"
