#!/bin/bash
# Round 4, session E: the cheaper integer split3 rounding (pieces kept in the high half, one NaN test, v_perm packing): bit-identity tests
# of every producer, op-level and engine timing; v_cvt_pk_bf16_f32 against the integer rounding on every input class
set -u
O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
build_ab/bf16_cvt_classes | tee $O/bf16_cvt_classes.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split3 or attention or persistent or layernorm" > $O/pytest_ops.log 2>&1; echo "rc=$?" >> $O/pytest_ops.log
tail -4 $O/pytest_ops.log | cut -c1-300; echo "t=$(( $(date +%s) - t0 ))"
timeout 400 python scripts/split3_bench.py --crops 64 --persist --no-error > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; cut -c1-900 $O/split3_bench_b64.jsonl | grep fc1
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"|max_abs' | cut -c1-600 | tee $O/mode_b64.log
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "split3_mode or small_trained or b64_tokens" > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
tail -3 $O/pytest_model.log | cut -c1-300
echo "total t=$(( $(date +%s) - t0 ))"
