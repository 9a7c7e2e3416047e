#!/bin/bash
# Round 5, session E: pricing the epoch protocol and the folded LDS bases: builds with -DTHMR_S16_EPOCH=0 (round-4 flags) and -DTHMR_S16_FOLD=0 against the round-4 library
# tree with -DTHMR_S16_PUBLISH / -DTHMR_S16_ARRIVE (scripts/build_ab_lib.py WORKTREE ...), same-box interleaved against the round-4 library.
set -u
O=gpurun_out/r5e; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "persistent or handover or hip_graph" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
grep -E "passed|failed|rc=" $O/pytest_subset.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_subset.log | head; grep -E "^E  " $O/pytest_subset.log | head -10; echo "t=$(( $(date +%s) - t0 ))"
for v in current e0 e0f0 f0; do
  lib=$([ $v = current ] && echo current || echo build_ab/$v/libtokenhmr_hip.so)
  timeout 400 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so $lib $O/ab_r4_vs_$v.json --batch 64 --reps 5 --iters 10 > $O/ab_$v.log 2> $O/ab_$v.err; tail -1 $O/ab_$v.err | cut -c1-200
done
python - <<'PY'
import json
for v in ("current", "e0", "e0f0", "f0"):
    try:
        j = json.load(open(f"gpurun_out/r5e/ab_r4_vs_{v}.json"))
        d = j["classes_B_minus_A_ms"]
        print(v, "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "fc2 B-A", d["gemm_fc2"], "fc1", d["gemm_fc1"], "| A fc2", j["A"]["classes_ms_mean"]["gemm_fc2"], "B fc2", j["B"]["classes_ms_mean"]["gemm_fc2"], "tok", j["outputs_bit_identical"]["token_idx"])
    except Exception as e:
        print(v, "parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
