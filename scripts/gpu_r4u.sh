#!/bin/bash
# Round 4, session U: the LBS skin kernel with every load of a pass issued up front (posed vertices of all CG crops in registers, bone
# matrices in LDS), a barrier-free skinning loop on MFMA tiles and the J19 partial sums afterwards: SMPL tests, stand-alone timing,
# crops-per-pass A/B (experiments build), rocprofv3 stats, bench parity numbers
set -u
O=gpurun_out/r4u; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_smpl_bounds.py tests/test_gpu_model.py tests/test_gpu_pipeline.py -q -m gpu -x -k "lbs or smpl or small or eval_loop" > $O/pytest_lbs.log 2>&1; echo "rc=$?" >> $O/pytest_lbs.log
grep -E "passed|failed|rc=|Error|assert" $O/pytest_lbs.log | cut -c1-300 | tail -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/lbs_bench.py 1 8 64 128 512 2> $O/lbs_bench.err | tee $O/lbs_bench.log
for cg in 1 2 4 8; do THMR_LIB=exp THMR_LBS_CG=$cg timeout 200 python scripts/lbs_bench.py 64 512 2>/dev/null | sed "s/^/cg=$cg /" | tee -a $O/lbs_cg_ab.log; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_lbs" -o p -- python "$R/scripts/lbs_bench.py" 512) > $O/prof_lbs.log 2>&1
find $O/prof_lbs -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4u/prof_lbs/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lbs" in r["Name"] or "gemm" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4u/bench.json"))
    r = j["roofline"]
    print("value", j["value"], "lbs_hbm", r.get("lbs_hbm"), "b512", r.get("lbs_hbm_b512"))
    print("parity", {k: j["parity"][k] for k in ("mismatches", "max_joint_err_m", "max_vertex_err_m")})
except Exception as e:
    print("bench parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
