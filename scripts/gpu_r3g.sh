#!/bin/bash
# Round 3, session G: verification of the 64x128 / 128x64 tiles in the cost model, the two-rank real-engine run on one GPU, default bench
set -u
O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert" $O/pytest_gpu.log | head -10; echo "t=$(( $(date +%s) - t0 ))"
for ms in "" "00"; do
  echo "== THMR_MID_SPLIT=$ms" >> $O/mid.log
  if [ -z "$ms" ]; then timeout 300 python scripts/mid_split_sweep.py 6 7 8 9 10 12 16 17 24 32 2>/dev/null | grep '^{' >> $O/mid.log
  else THMR_MID_SPLIT=$ms timeout 300 python scripts/mid_split_sweep.py 6 7 8 9 10 12 16 17 24 32 2>/dev/null | grep '^{' >> $O/mid.log; fi
done
cat $O/mid.log | cut -c1-700
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
timeout 300 python bench.py --gpus 2 --backend gloo --single-device --vit-depth 4 --batch 8 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err; python -c "
import json; d=json.load(open('$O/bench_two_ranks_one_gpu.json')); print(json.dumps(d['multi_gpu'])[:1500])"
timeout 200 python scripts/lbs_bench.py 512 64 512 2>/dev/null | grep LBS; 
echo "total t=$(( $(date +%s) - t0 ))"
