#!/bin/bash
# Round 4, session G: verification + committed profiles of the round's FINAL build: full GPU suite, smoke, the default bench line,
# rocprofv3 kernel stats of the bench in both modes, PMC passes on the persistent split3 GEMMs as the engine runs them now (fc1 with
# row-blocked output, fc2 with row-blocked A), batch sweep in both modes
set -u
O=gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json; tail -2 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4g/bench_full.json"))
    s3 = j.get("split3_mode", {})
    print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"].get("traffic"), "| split3", s3.get("value"), s3.get("roofline", {}).get("frac"), s3.get("parity"))
    print("pipeline", {k: (v.get("crops_per_s"), v.get("vs_forward_only")) for k, v in j.get("pipeline", {}).items() if isinstance(v, dict)})
    print(json.dumps(s3.get("classes_ms_per_step")))
except Exception as e:
    print("bench parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32" -o p -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_f32.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split3" -o p -- python "$R/bench.py" --vit-gemm split3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_split3.log 2>&1
find $O/prof_f32 $O/prof_split3 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -7 $O/prof_split3/p_kernel_stats.csv | cut -c1-160
grep -h '"value"' $O/prof_f32.log $O/prof_split3.log | cut -c1-170
for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r4_pmc_workload.py gemm) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
find $O/pmc -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_final.json 4 "gemm_split3_persist_kernel<5, 0, false>" "gemm_split3_persist_kernel<2, 2, false>" "gemm_split3_persist_kernel<4, 0, true>" "gemm_f32_kernel" 2>&1 | tail -2
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4g/pmc_final.json"))
for k, e in j.items():
    print(k, {x: e.get(x) for x in ("profiled_dur_us", "mfma_util_profiled", "traffic_bytes", "SQ_INSTS_VALU")})
PY
find $O/pmc -type f -delete 2>/dev/null
echo "t=$(( $(date +%s) - t0 ))"
for B in 1 2 4 8 16 32 128; do
  timeout 200 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(json.dumps({'mode':'f32','batch':$B,'crops_per_s':j['value'],'ms_per_step':j['ms_per_step']}))" | tee -a $O/batch_sweep.jsonl
  if [ $B -ge 3 ]; then timeout 200 python bench.py --batch $B --vit-gemm split3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(json.dumps({'mode':'split3','batch':$B,'crops_per_s':j['value'],'ms_per_step':j['ms_per_step']}))" | tee -a $O/batch_sweep.jsonl; fi
done
find $O -type f -size +12M -delete
echo "total t=$(( $(date +%s) - t0 ))"
