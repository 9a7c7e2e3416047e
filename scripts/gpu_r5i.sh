#!/bin/bash
# Round 5, session I: the persistent split3 GEMM publishing from inside its K loop with the branch's operands parked in LDS (session H's arm
# `vd`, now the source): full GPU suite, smoke, the default bench line, same-box A/B against the round-4 library.
set -u
O=gpurun_out/r5i; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep smoke $O/smoke.log | tail -13 | cut -c1-120
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-160 $O/bench_full.json; tail -2 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5i/bench_full.json"))
    o = j.get("exact_f32_mode", {})
    print("value", j["value"], j["vit_gemm"], "frac", j["roofline"]["frac"], "fc1 ms", j["roofline"]["avg_launch_ms"], "| f32", o.get("value"))
    print("parity total", json.dumps(j["parity"]["set"]["total"]))
    print("sweep", [(r["batch"], r["crops_per_s"], r["vs_timed_batch"]) for r in j["batch_sweep"]["rows"]])
    print(json.dumps(j["roofline"].get("classes_ms_per_step")))
except Exception as e:
    print("bench parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
for b in 64 32 8; do
timeout 400 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_current_b$b.json --batch $b --reps 5 --iters 10 > $O/ab_$b.log 2> $O/ab_$b.err
python - $b <<'PY'
import json, sys
try:
    j = json.load(open(f"gpurun_out/r5i/ab_r4_vs_current_b{sys.argv[1]}.json"))
    print("B", sys.argv[1], "A", j["A"]["ms_per_call_median"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_median"], j["B"]["crops_per_s_median"], "B/A", j["B_over_A_time"])
    print("   classes A", j["A"]["classes_ms_mean"]); print("   classes B-A", j["classes_B_minus_A_ms"])
except Exception as e:
    print("ab parse failed", e)
PY
done
echo "total t=$(( $(date +%s) - t0 ))"
