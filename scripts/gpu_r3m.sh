#!/bin/bash
# Round 3, session M: ring depth 4 vs 8 of the 16x16x4 small-M GEMM
set -u
O=gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ring16" > $O/pytest_ring16.log 2>&1; tail -2 $O/pytest_ring16.log
for env in "THMR_RING16_DEPTH=8" "THMR_RING16_DEPTH=4" "THMR_QKV_RING16=0" "THMR_RING16_DEPTH=8" "THMR_RING16_DEPTH=4" "THMR_QKV_RING16=0"; do
  echo "== $env" >> $O/ring16_depth.log
  env $env timeout 300 python scripts/mid_split_sweep.py 2 1 2 1 2>/dev/null | grep '^{' >> $O/ring16_depth.log
done
cat $O/ring16_depth.log | cut -c1-300
echo "total t=$(( $(date +%s) - t0 ))"
