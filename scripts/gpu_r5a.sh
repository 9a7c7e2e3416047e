#!/bin/bash
# Round 5, session A: first run of the round's build — split3 as the creation default (ABI 4), attention_b16's conflict-free LDS images,
# epoch hand-over flags, the half-tile tail of the split3 GEMM, the patch embed on the split3 pipe: full GPU suite, smoke (both modes),
# same-box interleaved A/B against the round-4 final library, LDS counters of the attention, the default bench line, the 8-rank rehearsal.
set -u
O=gpurun_out/r5a; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -14 $O/smoke.log | cut -c1-160
echo "t=$(( $(date +%s) - t0 ))"
timeout 600 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_r5_b64.json --reps 5 --iters 10 > $O/ab_b64.log 2> $O/ab_b64.err; tail -3 $O/ab_b64.err | cut -c1-300
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5a/ab_r4_vs_r5_b64.json"))
    for k in ("A", "B"):
        print(k, j[k]["build"][:60], j[k]["ms_per_call_windows"], j[k]["ms_per_call_median"], j[k]["crops_per_s_median"])
    print("B/A", j["B_over_A_time"], "classes B-A", j["classes_B_minus_A_ms"])
    print("A classes", j["A"]["classes_ms_mean"]); print("B classes", j["B"]["classes_ms_mean"])
    print("bit-identical", j["outputs_bit_identical"]); print("maxdiff", {k: v for k, v in j["max_abs_diff"].items() if v})
except Exception as e:
    print("ab parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
for p in "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r5_pmc_workload.py attn 5) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
find $O/pmc -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
for d in $O/pmc/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_lds_attention.json 4 "vit_attention_b16_kernel<3, true>" > /dev/null 2> $O/pmc_json.err; tail -2 $O/pmc_json.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5a/pmc_lds_attention.json"))
    for k, e in j.items():
        print(k, {x: (round(v, 1) if isinstance(v, float) else v) for x, v in e.items() if x.startswith("SQ_") or x == "profiled_dur_us"})
except Exception as ex:
    print("pmc parse failed", ex)
PY
rm -rf $O/pmc
echo "t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json; tail -2 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5a/bench_full.json"))
    o = j.get("exact_f32_mode", {})
    print("value", j["value"], j["vit_gemm"], j.get("vit_gemm_set_by"), "frac", j["roofline"]["frac"], "fc1 ms", j["roofline"]["avg_launch_ms"], "| f32", o.get("value"), o.get("roofline", {}).get("frac"))
    print("facade", j.get("facade"))
    print("parity", json.dumps(j.get("parity"))[:1200])
    print("sweep", json.dumps(j.get("batch_sweep")))
    print(json.dumps(j["roofline"].get("classes_ms_per_step")))
    print({k: j["roofline"].get(k) for k in ("attention", "patch_embed_hbm", "lbs_hbm")})
except Exception as e:
    print("bench parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_512.json 2> $O/rank8_512.err; cut -c1-160 $O/rank8_512.json; tail -3 $O/rank8_512.err | cut -c1-300
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --global-batch 509 --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_509.json 2> $O/rank8_509.err; cut -c1-160 $O/rank8_509.json; tail -3 $O/rank8_509.err | cut -c1-300
python - <<'PY'
import json
for f in ("rank8_512", "rank8_509"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5a/{f}.json") if l.startswith("{")][-1])
        m = j["multi_gpu"]
        print(f, "gathered_ok", j.get("gathered_records_ok"), "cross", m["cross_rank_check"], "crops", [r["crops"] for r in m["per_rank"]], "bcast_ms", m["bcast_ms"], j["vit_gemm"])
    except Exception as e:
        print(f, "parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
