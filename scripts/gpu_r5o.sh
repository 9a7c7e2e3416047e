#!/bin/bash
# Round 5, session O: the bias + residual epilogue of the per-tile split3 GEMM loads the residual of row tile mi + 1 before it stores row tile
# mi (C and the residual are one buffer: written per element the compiler keeps 16 load round trips in a row): op tests, A/B vs the committed build.
set -u
O=gpurun_out/r5o; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "split3" -p no:cacheprovider > $O/pytest_ops.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.log
grep -E "passed|failed|rc=" $O/pytest_ops.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_ops.log | head; grep -E "^E  " $O/pytest_ops.log | head; echo "t=$(( $(date +%s) - t0 ))"
for b in 64 32 16; do
  timeout 300 python scripts/ab_same_box.py --a build_ab/head/libtokenhmr_hip.so --b current --out $O/ab_head_vs_current_b$b.json --batch $b --reps 5 --iters 10 > $O/ab_$b.log 2> $O/ab_$b.err
  python - $b <<'PY'
import json, sys
try:
    j = json.load(open(f"gpurun_out/r5o/ab_head_vs_current_b{sys.argv[1]}.json"))
    d = j["classes_B_minus_A_ms"]
    print("B", sys.argv[1], "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "|", {k: d[k] for k in ("gemm_proj", "gemm_fc2", "gemm_qkv", "gemm_fc1", "layernorm")}, "| bit-identical", all(j["outputs_bit_identical"].values()))
except Exception as e:
    print(sys.argv[1], "parse failed", e)
PY
done
echo "total t=$(( $(date +%s) - t0 ))"
