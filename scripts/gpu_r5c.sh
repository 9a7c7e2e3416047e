#!/bin/bash
# Round 5, session C: the persistent fc2 with the epoch in a register again and its arrival under the slab loads; full suite + same-box
# A/B against the round-4 library at 64 crops (reps 7).
set -u
O=gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
timeout 600 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_r5_b64.json --batch 64 --reps 7 --iters 10 > $O/ab_b64.log 2> $O/ab_b64.err; tail -2 $O/ab_b64.err | cut -c1-300
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5c/ab_r4_vs_r5_b64.json"))
    print("A", j["A"]["ms_per_call_windows"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_windows"], j["B"]["crops_per_s_median"], "B/A", j["B_over_A_time"])
    print("   classes B-A", j["classes_B_minus_A_ms"]); print("   B classes", j["B"]["classes_ms_mean"])
    print("   token_idx identical", j["outputs_bit_identical"]["token_idx"], "max vert diff", j["max_abs_diff"]["pred_vertices"])
except Exception as e:
    print("ab parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
