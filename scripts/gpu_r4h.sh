#!/bin/bash
# Round 4, session H: same-box A/B of the row-blocked fc1 -> fc2 operand (experiments build, THMR_SPLIT3_BS_BLK=0 switches it off), interleaved
set -u
O=gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for rep in 1 2 3; do
  for blk in 1 0; do
    echo "== rep $rep THMR_SPLIT3_BS_BLK=$blk"
    THMR_LIB=exp THMR_SPLIT3_BS_BLK=$blk timeout 300 python scripts/mode_bench.py 64 8 2> $O/err_${rep}_${blk}.log | grep -E '"mode": "split3"' | cut -c1-420 | tee -a $O/ab_blocked_b64.log
  done
done
echo "total t=$(( $(date +%s) - t0 ))"
