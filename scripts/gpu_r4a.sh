#!/bin/bash
# Round 4, session A: first run of the persistent split3 GEMM: bit-identity tests, op-level timing beside the per-tile kernel, engine A/B
set -u
O=gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "persistent" > $O/pytest_persist.log 2>&1; echo "rc=$?" >> $O/pytest_persist.log
tail -15 $O/pytest_persist.log; echo "t=$(( $(date +%s) - t0 ))"
timeout 400 python scripts/split3_bench.py --crops 64 --persist --no-error > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; cat $O/split3_bench_b64.jsonl | cut -c1-700
timeout 300 python scripts/split3_bench.py --crops 32 --persist --no-error > $O/split3_bench_b32.jsonl 2> $O/split3_bench_b32.err; cat $O/split3_bench_b32.jsonl | cut -c1-700
echo "t=$(( $(date +%s) - t0 ))"
for cfg in "0 0" "1 0" "1 1" "1 2"; do
  set -- $cfg
  echo "== THMR_SPLIT3_PERSIST=$1 THMR_SPLIT3_FC1_MODE=$2"
  THMR_SPLIT3_PERSIST=$1 THMR_SPLIT3_FC1_MODE=$2 timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64_p$1_f$2.err | grep -E '"mode": "split3"|max_abs' | cut -c1-600 | tee -a $O/mode_b64.log
done
echo "t=$(( $(date +%s) - t0 ))"
THMR_SPLIT3_PERSIST=1 THMR_SPLIT3_FC1_MODE=2 timeout 300 python scripts/mode_bench.py 32 10 2> $O/mode_b32.err | grep -E '"mode": "split3"' | cut -c1-400 | tee $O/mode_b32.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "b64_tokens or split3_mode or trained" -s > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
grep -E "golden|passed|failed|rc=" $O/pytest_model.log | cut -c1-400 | tail -20
echo "total t=$(( $(date +%s) - t0 ))"
