#!/bin/bash
# Round 4, session AG: the final build once more on another fresh box — full GPU suite + the default bench line (reproducibility of r4z)
set -u
O=gpurun_out/r4ag; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; echo "t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4ag/bench_full.json"))
print("value", j["value"], j["vit_gemm"], "frac", j["roofline"]["frac"], "| f32", j["exact_f32_mode"]["value"], j["exact_f32_mode"]["roofline"]["frac"], "build", j["build"][:48])
print("parity", {k: j["parity"][k] for k in ("mismatches", "max_joint_err_m", "max_vertex_err_m")}, "set", {k: v["mismatches"] for k, v in j["parity"]["set"].items()})
print(json.dumps(j["roofline"]["classes_ms_per_step"]))
PY
echo "total t=$(( $(date +%s) - t0 ))"
