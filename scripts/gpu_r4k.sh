#!/bin/bash
# Round 4, session K: first run of the 16x16x32 split3 GEMM (gemm_split16.hip): tests, op-level A/B against the 32x32x16 kernels, engine timing
set -u
O=gpurun_out/r4k; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split3 or persistent" > $O/pytest_ops.log 2>&1; echo "rc=$?" >> $O/pytest_ops.log
tail -8 $O/pytest_ops.log | cut -c1-500; echo "t=$(( $(date +%s) - t0 ))"
timeout 400 python scripts/split3_bench.py --crops 64 --persist --no-error > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; cut -c1-1500 $O/split3_bench_b64.jsonl; tail -2 $O/split3_bench_b64.err
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"|max_abs' | cut -c1-700 | tee $O/mode_b64.log
timeout 300 python scripts/mode_bench.py 32 10 2> $O/mode_b32.err | grep -E '"mode": "split3"' | cut -c1-500 | tee $O/mode_b32.log
timeout 300 python scripts/mode_bench.py 16 10 2> $O/mode_b16.err | grep -E '"mode": "split3"' | cut -c1-500 | tee $O/mode_b16.log
timeout 300 python scripts/mode_bench.py 8 10 2> $O/mode_b8.err | grep -E '"mode": "split3"' | cut -c1-500 | tee $O/mode_b8.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "split3_mode or small_trained or b64_tokens" -s > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
grep -E "golden full|passed|failed|rc=" $O/pytest_model.log | cut -c1-330 | tail -12
echo "total t=$(( $(date +%s) - t0 ))"
