#!/bin/bash
# Round 4, session M: split3 pieces through v_cvt_pk_bf16_f32 in every producer + the leaner split3-output epilogue of the 16x16x32 kernel;
# per-class A/B of the persistent decomposition (THMR_SPLIT3_PERSIST_MASK: 1 qkv, 2 proj, 4 fc1, 8 fc2)
set -u
O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "split3 or persistent or layernorm or ln_" > $O/pytest_ops.log 2>&1; echo "rc=$?" >> $O/pytest_ops.log
tail -5 $O/pytest_ops.log | cut -c1-400; echo "t=$(( $(date +%s) - t0 ))"
for mk in 8 12 8; do
  THMR_LIB=exp THMR_SPLIT3_PERSIST_MASK=$mk timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64_mask$mk.err | grep -E '"mode": "split3"' | cut -c1-600 | sed "s/^/mask=$mk /" | tee -a $O/mode_b64_mask.log
done
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode"|max_abs' | cut -c1-700 | tee $O/mode_b64.log
echo "t=$(( $(date +%s) - t0 ))"
timeout 400 python scripts/split3_bench.py --crops 64 --persist --no-error > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; cut -c1-1200 $O/split3_bench_b64.jsonl; tail -2 $O/split3_bench_b64.err
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "b64 or small_trained or split3_mode" > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
grep -E "golden full|passed|failed|rc=" $O/pytest_model.log | cut -c1-300 | tail -12
echo "total t=$(( $(date +%s) - t0 ))"
