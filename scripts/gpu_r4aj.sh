#!/bin/bash
# Round 4, session AJ: verification + committed profiles of the round's FINAL build (16x16x32 split3 GEMMs, hardware bf16 conversion,
# the attention of the split3 mode on the bf16 matrix pipe too, split3 as the timed mode of bench.py): full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the bench in both
# modes, PMC passes on the GEMMs as the engine runs them, batch sweep
set -u
O=gpurun_out/r4aj; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json; tail -2 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4aj/bench_full.json"))
    o = j.get("exact_f32_mode", {})
    print("value", j["value"], j["vit_gemm"], "frac", j["roofline"]["frac"], "traffic", j["roofline"].get("traffic"), "| f32", o.get("value"), o.get("roofline", {}).get("frac"))
    print("parity", json.dumps(j.get("parity"))[:1500])
    print("pipeline", {k: (v.get("crops_per_s"), v.get("vs_forward_only")) for k, v in j.get("pipeline", {}).items() if isinstance(v, dict)})
    print(json.dumps(j["roofline"].get("classes_ms_per_step")))
except Exception as e:
    print("bench parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split3" -o p -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_split3.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32" -o p -- python "$R/bench.py" --vit-gemm f32 --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_f32.log 2>&1
find $O/prof_f32 $O/prof_split3 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -8 $O/prof_split3/*/p_kernel_stats.csv $O/prof_split3/p_kernel_stats.csv 2>/dev/null | cut -c1-170
grep -h '"value"' $O/prof_f32.log $O/prof_split3.log | cut -c1-170
echo "t=$(( $(date +%s) - t0 ))"
for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r4_pmc_workload.py gemm) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
find $O/pmc -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
for d in $O/pmc/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_final.json 4 "gemm_split16_kernel<4, 5, false, false>" "gemm_split16_kernel<4, 4, false, false>" "gemm_split16_kernel<4, 2, false, false>" "gemm_split16_kernel<4, 4, true, true>" "gemm_f32_kernel" "vit_attention_b16_kernel<3, true>" "vit_attention_persistent_kernel<0, true>" > /dev/null 2>$O/pmc_json.err; tail -2 $O/pmc_json.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4aj/pmc_final.json"))
    for k, e in j.items():
        print(k, {x: e.get(x) for x in ("profiled_dur_us", "mfma_util_profiled", "traffic_bytes", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")})
except Exception as ex:
    print("pmc parse failed", ex)
PY
rm -rf $O/pmc
echo "t=$(( $(date +%s) - t0 ))"
for b in 32 8; do timeout 300 python scripts/mode_bench.py $b 10 2> $O/mode_b$b.err | grep -E '"mode"' | head -2 | cut -c1-420 | tee $O/mode_b$b.log; done
echo "total t=$(( $(date +%s) - t0 ))"
