#!/bin/bash
# Round 4, session L: dispatch rule of the 16x16x32 split3 kernels (persistent only for K >= min_k): A/B of min_k on one box, model parity with
# the fp64-derived bound, engine timing at 64/32/16/8
set -u
O=gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for mk in 2560 0 100000 2560; do
  THMR_LIB=exp THMR_SPLIT3_PERSIST_MIN_K=$mk timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64_mk$mk.err | grep -E '"mode": "split3"' | cut -c1-600 | sed "s/^/min_k=$mk /" | tee -a $O/mode_b64_mink.log
done
echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode"|max_abs' | cut -c1-700 | tee $O/mode_b64.log
for b in 32 16 8; do timeout 300 python scripts/mode_bench.py $b 10 2> $O/mode_b$b.err | grep -E '"mode": "split3"' | cut -c1-500 | tee $O/mode_b$b.log; done
echo "t=$(( $(date +%s) - t0 ))"
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
grep -E "golden full|passed|failed|rc=" $O/pytest_model.log | cut -c1-400 | tail -14
echo "total t=$(( $(date +%s) - t0 ))"
