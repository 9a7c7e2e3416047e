#!/bin/bash
# Round 4, session D: full GPU suite on the two-library build (shipped + experiments), smoke, the bench line with the pipeline extra
set -u
O=gpurun_out/r4d; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-300 $O/bench_full.json; tail -3 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4d/bench_full.json"))
    print("value", j["value"], "split3", j.get("split3_mode", {}).get("value"), "frac", j["roofline"]["frac"])
    print(json.dumps(j.get("pipeline"))[:1800])
    print(json.dumps(j.get("split3_mode", {}).get("classes_ms_per_step")))
except Exception as e:
    print("bench parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
