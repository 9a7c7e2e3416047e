#!/bin/bash
# Round 5, session B: after the first session's findings — the persistent fc2 publishes its slab outside the K loop (the epoch flag had
# put a scratch reload + vmcnt(0) into the loop: +12 % on fc2), per-lane LDS bases with the stage offsets folded in; the head-regime test
# in both modes; the 8-rank single-device rehearsal with the ranks' forwards in turns.  Full suite, same-box A/B at 64 / 32 / 8 crops.
set -u
O=gpurun_out/r5b; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
for b in 64 32 8; do
  timeout 600 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_r5_b$b.json --batch $b --reps 5 --iters $(( b >= 32 ? 10 : 30 )) > $O/ab_b$b.log 2> $O/ab_b$b.err; tail -2 $O/ab_b$b.err | cut -c1-300
done
python - <<'PY'
import json
for b in (64, 32, 8):
    try:
        j = json.load(open(f"gpurun_out/r5b/ab_r4_vs_r5_b{b}.json"))
        print(b, "A", j["A"]["ms_per_call_windows"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_windows"], j["B"]["crops_per_s_median"], "B/A", j["B_over_A_time"])
        print("   classes B-A", j["classes_B_minus_A_ms"]); print("   B classes", j["B"]["classes_ms_mean"])
        print("   token_idx identical", j["outputs_bit_identical"]["token_idx"], "max vert diff", j["max_abs_diff"]["pred_vertices"])
    except Exception as e:
        print(b, "ab parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_512.json 2> $O/rank8_512.err; cut -c1-160 $O/rank8_512.json; grep -E "Error|error" $O/rank8_512.err | head -5 | cut -c1-300
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --global-batch 509 --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_509.json 2> $O/rank8_509.err; cut -c1-160 $O/rank8_509.json; grep -E "Error|error" $O/rank8_509.err | head -5 | cut -c1-300
python - <<'PY'
import json
for f in ("rank8_512", "rank8_509"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5b/{f}.json") if l.startswith("{")][-1])
        m = j["multi_gpu"]
        print(f, "gathered_ok", j.get("gathered_records_ok"), "cross", m["cross_rank_check"], "crops", [r["crops"] for r in m["per_rank"]], "bcast_ms", m["bcast_ms"], j["vit_gemm"])
    except Exception as e:
        print(f, "parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
