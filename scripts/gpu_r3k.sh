#!/bin/bash
# Round 3, session K: all-to-all grid barrier of the persistent decoder kernel (A/B), key-split regime {1, 2} verification, small-batch latency
set -u
O=gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
for bm in 1 0 1 0; do
  echo "== THMR_DEC_BARRIER=$bm" >> $O/barrier_ab.log
  THMR_DEC_BARRIER=$bm timeout 200 python scripts/head_bench.py 1 2 8 32 64 128 2>/dev/null | grep "head_forward" >> $O/barrier_ab.log
done
cat $O/barrier_ab.log | cut -c1-120
for bm in 1 0; do
  echo "== THMR_DEC_BARRIER=$bm" >> $O/latency.log
  THMR_DEC_BARRIER=$bm timeout 300 python scripts/mid_split_sweep.py 1 2 3 4 6 8 16 2>/dev/null | grep '^{' >> $O/latency.log
done
cat $O/latency.log | cut -c1-500
THMR_DEC_TIMELINE=1 timeout 100 python scripts/head_bench.py 1 8 2>/dev/null | grep timeline | cut -c1-900 > $O/decoder_timeline.log; cat $O/decoder_timeline.log
echo "total t=$(( $(date +%s) - t0 ))"
