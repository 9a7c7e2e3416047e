#!/bin/bash
# Round 3, session C: big-tile split-K for proj / fc2 at mid-size batches (crossover sweep), hybrid LBS skin kernel, improved hand-off micro
set -u
O=gpurun_out/r3c; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert" $O/pytest_gpu.log | head -10; echo "t=$(( $(date +%s) - t0 ))"
timeout 200 python scripts/lbs_bench.py 1 64 512 2>/dev/null | grep LBS > $O/lbs_bench.log; cat $O/lbs_bench.log
mkdir -p build_ab
hipcc --offload-arch=gfx950 -O3 -o build_ab/xcd_handoff scripts/micro/xcd_handoff.hip 2>/dev/null
timeout 120 build_ab/xcd_handoff 200 > $O/xcd_handoff.log 2>&1; cat $O/xcd_handoff.log
echo "t=$(( $(date +%s) - t0 ))"
: > $O/mid_split_sweep.jsonl
for ms in 0 2 4; do
  THMR_MID_SPLIT=$ms timeout 300 python scripts/mid_split_sweep.py 2>/dev/null | grep '^{' >> $O/mid_split_sweep.jsonl
done
timeout 300 python scripts/mid_split_sweep.py 2>/dev/null | grep '^{' >> $O/mid_split_sweep.jsonl
cat $O/mid_split_sweep.jsonl
echo "total t=$(( $(date +%s) - t0 ))"
