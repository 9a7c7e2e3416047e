#!/bin/bash
# Round 4, session F: the row-blocked split3 operand between fc1 and fc2: bit-identity tests, op-level and engine timing
set -u
O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split3 or persistent" > $O/pytest_ops.log 2>&1; echo "rc=$?" >> $O/pytest_ops.log
tail -6 $O/pytest_ops.log | cut -c1-400; echo "t=$(( $(date +%s) - t0 ))"
timeout 400 python scripts/split3_bench.py --crops 64 --persist --no-error > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; grep -E '"fc1"|"fc2"' $O/split3_bench_b64.jsonl | cut -c1-1400
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"|max_abs' | cut -c1-600 | tee $O/mode_b64.log
timeout 300 python scripts/mode_bench.py 32 10 2> $O/mode_b32.err | grep -E '"mode": "split3"' | cut -c1-600 | tee $O/mode_b32.log
timeout 300 python scripts/mode_bench.py 16 10 2> $O/mode_b16.err | grep -E '"mode": "split3"' | cut -c1-600 | tee $O/mode_b16.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "split3_mode or small_trained or b64_tokens" > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
tail -3 $O/pytest_model.log | cut -c1-300
echo "total t=$(( $(date +%s) - t0 ))"
