#!/bin/bash
# Round 3, session AO: last verification (after the 3 ... 15 crop ranges of the split3 mode, the tokenizer drop-ins, the 256x256 variant): GPU suite, smoke, both bench lines
set -u
O=gpurun_out/r3ao; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
timeout 600 python bench.py --vit-gemm split3 --batch 8 --no-cpu-baseline > $O/bench_split3_b8.json 2> $O/bench_split3_b8.err; cut -c1-200 $O/bench_split3_b8.json
echo "total t=$(( $(date +%s) - t0 ))"
