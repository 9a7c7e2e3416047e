#!/bin/bash
# Round 4, session T: the LBS skinning as MFMA tiles (csrc/lbs.hip: T^T = A_b^T . W^T on v_mfma_f32_16x16x4_f32 instead of 72 broadcast
# ds_read_b128 per thread and crop) — SMPL tests, stand-alone timing at 64 / 512 crops, rocprofv3 stats of the stage; and the hipGraph
# capture test of thmr_forward in both modes
set -u
O=gpurun_out/r4t; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_smpl_bounds.py tests/test_gpu_model.py tests/test_gpu_pipeline.py -q -m gpu -x -k "lbs or smpl or hip_graph or small or eval_loop" > $O/pytest_lbs_graph.log 2>&1; echo "rc=$?" >> $O/pytest_lbs_graph.log
grep -E "passed|failed|rc=|Error|assert" $O/pytest_lbs_graph.log | cut -c1-300 | tail -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/lbs_bench.py 1 64 512 2> $O/lbs_bench.err | tee $O/lbs_bench.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_lbs" -o p -- python "$R/scripts/lbs_bench.py" 512) > $O/prof_lbs.log 2>&1
find $O/prof_lbs -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -6 $(find $O/prof_lbs -name '*kernel_stats.csv' | head -1) | cut -c1-200
echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4t/bench.json"))
    r = j["roofline"]
    print("value", j["value"], "lbs_hbm", r.get("lbs_hbm"), "b512", r.get("lbs_hbm_b512"))
    print("parity", {k: j["parity"][k] for k in ("mismatches", "max_joint_err_m", "max_vertex_err_m")})
except Exception as e:
    print("bench parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
