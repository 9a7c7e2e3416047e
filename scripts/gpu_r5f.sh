#!/bin/bash
# Round 5, session F: verification + committed profiles of the round's FINAL build: full GPU suite, smoke (both modes), the default bench line
# (with the CPU baseline), rocprofv3 kernel stats of the bench in both modes, PMC passes (SQ / LDS / FETCH / WRITE / L2, separate runs) on the
# kernels as the engine runs them, same-box A/B against the round-4 library, the driver's launcher form at world size 1 over RCCL.
set -u
O=gpurun_out/r5f; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep smoke $O/smoke.log | tail -13 | cut -c1-120
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-160 $O/bench_full.json; tail -2 $O/bench_full.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5f/bench_full.json"))
    o = j.get("exact_f32_mode", {})
    print("value", j["value"], j["vit_gemm"], j.get("vit_gemm_set_by"), "frac", j["roofline"]["frac"], "fc1 ms", j["roofline"]["avg_launch_ms"], "traffic", j["roofline"].get("traffic"), "| f32", o.get("value"), o.get("roofline", {}).get("frac"))
    print("facade", {k: j["facade"][k] for k in ("crops_per_s", "vs_engine_forward", "vit_gemm", "mode_set_by")})
    print("parity total", json.dumps(j["parity"]["set"]["total"]), "cpu", {k: j["cpu_baseline"][k] for k in ("value", "cores", "kind")}, "gpu/cpu", j.get("gpu_over_cpu"))
    print("sweep", [(r["batch"], r["crops_per_s"], r["vs_timed_batch"]) for r in j["batch_sweep"]["rows"]])
    print("pipeline", {k: (v.get("crops_per_s"), v.get("vs_forward_only")) for k, v in j.get("pipeline", {}).items() if isinstance(v, dict)})
    print(json.dumps(j["roofline"].get("classes_ms_per_step")))
    print({k: j["roofline"].get(k) for k in ("attention", "patch_embed_hbm", "lbs_hbm", "lbs_hbm_b512")})
except Exception as e:
    print("bench parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split3" -o p -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_split3.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32" -o p -- python "$R/bench.py" --vit-gemm f32 --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_f32.log 2>&1
find $O/prof_f32 $O/prof_split3 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -9 $O/prof_split3/*/p_kernel_stats.csv $O/prof_split3/p_kernel_stats.csv 2>/dev/null | cut -c1-150
grep -h '"value"' $O/prof_f32.log $O/prof_split3.log | cut -c1-120
echo "t=$(( $(date +%s) - t0 ))"
for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r5_pmc_workload.py gemm 5) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$? t=$(( $(date +%s) - t0 ))"
done
find $O/pmc -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
for d in $O/pmc/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_final.json 4 "gemm_split16_kernel<4, 5, false, false>" "gemm_split16_kernel<4, 4, false, false>" "gemm_split16_tail_kernel<2, false>" "gemm_split16_kernel<4, 2, false, false>" "gemm_split16_kernel<4, 4, true, true>" "gemm_f32_kernel" "vit_attention_b16_kernel<3, true>" > /dev/null 2>$O/pmc_json.err; tail -2 $O/pmc_json.err
python - <<'PY'
import json
try:
    from tokenhmr_amd import _cabi
    j = json.load(open("gpurun_out/r5f/pmc_final.json"))
    j["_build"] = _cabi.load().thmr_build_info().decode()
    j["_note"] = "every pass (sq, lds, fetch, write, l2) is of THIS build, separate rocprofv3 --pmc runs of scripts/r5_pmc_workload.py; FETCH_SIZE doubled per the gfx950 correction"
    json.dump(j, open("gpurun_out/r5f/pmc_final.json", "w"), indent=1)
    for k, e in j.items():
        if isinstance(e, dict):
            print(k, {x: e.get(x) for x in ("profiled_dur_us", "mfma_util_profiled", "traffic_bytes", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "sq_wait_inst_any_frac_of_wave_cycles")})
except Exception as ex:
    print("pmc parse failed", ex)
PY
rm -rf $O/pmc
echo "t=$(( $(date +%s) - t0 ))"
timeout 400 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_r5_final_b64.json --batch 64 --reps 5 --iters 10 > $O/ab.log 2> $O/ab.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r5f/ab_r4_vs_r5_final_b64.json"))
    print("A", j["A"]["ms_per_call_windows"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_windows"], j["B"]["crops_per_s_median"], "B/A", j["B_over_A_time"])
    print("   classes B-A", j["classes_B_minus_A_ms"])
except Exception as e:
    print("ab parse failed", e)
PY
echo "t=$(( $(date +%s) - t0 ))"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_launcher_world1.json 2> $O/bench_launcher_world1.err; grep -h '^{' $O/bench_launcher_world1.json | cut -c1-200
echo "total t=$(( $(date +%s) - t0 ))"
