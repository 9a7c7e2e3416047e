#!/bin/bash
# ONE parametrised GPU-session runner (round 6; replaces the per-session scripts/gpu_r*.sh files of rounds 1-5, whose index — what each
# session ran and which files under profiles/ it produced — is scripts/SESSIONS.md).
#
#   gpurun --timeout 1500 -- 'scripts/session.sh <name> <step> [<step> ...]'        # outputs under gpurun_out/<name>/
#
# Steps (each prints a short summary; the full logs stay in the output directory):
#   tests[=<pytest -k expression>]      the -m gpu suite (or the selected tests)
#   smoke                               __graft_entry__.smoke()
#   bench[=<bench.py arguments>]        one bench line -> bench<n>.json (+ a parsed summary)
#   classprof=<B,B,...>                 per-class ms of one call at each batch size (default mode), scripts/small_batch_prof.py
#   ab=<libA>:<batch>[:<extra args>]    scripts/ab_same_box.py <libA> vs the in-tree library at that batch size
#   abexp=<batch>:<A-env>:<B-env>       two knob settings of the experiments build against each other (env lists comma-separated K=V)
#   rocprof=<bench.py arguments>        rocprofv3 --kernel-trace --stats of bench.py --no-extras --no-cpu-baseline <arguments> -> kernel_stats<n>.csv
#   pmc=<workload>:<n>                  the PMC passes of MI355X_MICROARCH.md (separate runs, --kernel-trace only) over
#                                       `python scripts/r5_pmc_workload.py <workload> <n>` -> pmc_<workload>.json (kernels: $PMC_KERNELS, ';'-separated)
#   run=<command>                       anything else, logged to run<n>.log
set -u
NAME="$1"; shift
O=gpurun_out/$NAME; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
n=0
stamp() { echo "[$NAME] $1 done, t=$(( $(date +%s) - t0 )) s"; }

PMC_PASSES=${PMC_PASSES:-"sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE|sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS|lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC|fetch:FETCH_SIZE|write:WRITE_SIZE|l2:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"}

for step in "$@"; do
  n=$((n + 1))
  kind="${step%%=*}"; arg=""; [ "$kind" != "$step" ] && arg="${step#*=}"
  case "$kind" in
    tests)
      if [ -n "$arg" ]; then timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "$arg" > $O/pytest$n.log 2>&1
      else timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest$n.log 2>&1; fi
      echo "pytest rc=$?" >> $O/pytest$n.log
      grep -E "passed|failed|rc=" $O/pytest$n.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest$n.log | head -12; grep -E "^E  " $O/pytest$n.log | head -30
      grep -E "^\[golden" $O/pytest$n.log | grep -v "mismatches: 0 of" | cut -c1-260 | head -40 ;;
    smoke)
      timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke$n.log 2>&1; echo "smoke rc=$?"; grep smoke $O/smoke$n.log | tail -13 | cut -c1-140 ;;
    bench)
      timeout 1700 python bench.py $arg > $O/bench$n.json 2> $O/bench$n.err; echo "bench rc=$?"; tail -2 $O/bench$n.err
      python - $O/bench$n.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = j.get("roofline") or {}
    print("value", j["value"], j.get("vit_gemm"), "ms", j["ms_per_step"], "frac", r.get("frac"), "fc1 ms", r.get("avg_launch_ms"), "traffic", r.get("traffic"), "build", j.get("build"))
    o = j.get("exact_f32_mode") or {}
    print("f32 mode", o.get("value"), (o.get("roofline") or {}).get("frac"))
    if j.get("parity"): print("parity", {k: v for k, v in j["parity"].items() if k != "set"}, "set total", (j["parity"].get("set") or {}).get("total"))
    if j.get("batch_sweep"): print("sweep", [(x["batch"], x["ms_per_call"], x["crops_per_s"], x["vs_timed_batch"], (x.get("parity") or {}).get("mismatches")) for x in j["batch_sweep"].get("rows", [])], j["batch_sweep"].get("timed_batch_row_vs_value"), j["batch_sweep"].get("error"))
    if j.get("cpu_baseline"): print("cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "value_b1")}, "gpu/cpu", j.get("gpu_over_cpu"))
    if j.get("facade"): print("facade", j["facade"].get("crops_per_s"), "pipeline", {k: (v.get("crops_per_s"), v.get("vs_forward_only")) for k, v in (j.get("pipeline") or {}).items() if isinstance(v, dict)})
    print("classes", json.dumps(r.get("classes_ms_per_step")))
    if j.get("multi_gpu"): print("multi_gpu", {k: j["multi_gpu"].get(k) for k in ("bcast_ms", "gather_ms_exposed", "rank_step_ms", "expected", "efficiency_vs_expected")}, (j["multi_gpu"].get("cross_rank_check") or {}).get("bit_identical"))
except Exception as e:
    print("bench parse failed:", e)
PY
      ;;
    classprof)
      SKIP_GEMM=1 timeout 900 python scripts/small_batch_prof.py ${arg//,/ } > $O/classprof$n.log 2>&1; grep "^B " $O/classprof$n.log ;;
    ab)
      IFS=: read -r liba batch extra <<< "$arg"
      timeout 600 python scripts/ab_same_box.py --a "$liba" --b current --out $O/ab${n}_b$batch.json --batch $batch --reps 5 --iters 10 $extra > $O/ab$n.log 2> $O/ab$n.err
      python - $O/ab${n}_b$batch.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print("A", j["A"]["ms_per_call_median"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_median"], j["B"]["crops_per_s_median"], "B/A time", j["B_over_A_time"], "bit-identical", j.get("outputs_bit_identical"))
    print("   classes A", j["A"]["classes_ms_mean"]); print("   classes B-A", j["classes_B_minus_A_ms"])
except Exception as e:
    print("ab parse failed:", e)
PY
      ;;
    abexp)
      IFS=: read -r batch aenv benv <<< "$arg"
      timeout 600 python scripts/ab_same_box.py --a exp --b exp --a-env ${aenv//,/ } --b-env ${benv//,/ } --out $O/abexp${n}_b$batch.json --batch $batch --reps 5 --iters 10 > $O/abexp$n.log 2> $O/abexp$n.err
      python - $O/abexp${n}_b$batch.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print("A", j["A"]["ms_per_call_median"], j["A"]["crops_per_s_median"], "| B", j["B"]["ms_per_call_median"], j["B"]["crops_per_s_median"], "B/A time", j["B_over_A_time"], "bit-identical", j.get("outputs_bit_identical"))
    print("   classes A", j["A"]["classes_ms_mean"]); print("   classes B-A", j["classes_B_minus_A_ms"])
except Exception as e:
    print("abexp parse failed:", e)
PY
      ;;
    rocprof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof$n" -o p -- python "$R/bench.py" --no-cpu-baseline --no-extras $arg) > $O/prof$n.log 2>&1
      f=$(find $O/prof$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats$n.csv; rm -rf $O/prof$n
      head -12 $O/kernel_stats$n.csv 2>/dev/null | cut -c1-170; grep -h '"value"' $O/prof$n.log | cut -c1-140 ;;
    pmc)
      IFS=: read -r wl cnt <<< "$arg"; cnt=${cnt:-5}
      rm -rf $O/pmc_$wl
      IFS='|' read -ra passes <<< "$PMC_PASSES"
      for p in "${passes[@]}"; do
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc_$wl/${p%%:*}" -o p -- python $R/scripts/r5_pmc_workload.py $wl $cnt) >> $O/pmc_$wl.log 2>&1
        echo "pmc pass ${p%%:*} rc=$? t=$(( $(date +%s) - t0 ))"
      done
      find $O/pmc_$wl -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
      for d in $O/pmc_$wl/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
      IFS=';' read -ra kern <<< "${PMC_KERNELS:-vit_attention_b16_kernel<3, true>}"
      python scripts/pmc_kernels_to_json.py $O/pmc_$wl $O/pmc_$wl.json 4 "${kern[@]}" > /dev/null 2> $O/pmc_${wl}_json.err; tail -2 $O/pmc_${wl}_json.err
      python - $O/pmc_$wl.json <<'PY'
import json, sys
try:
    from tokenhmr_amd import _cabi
    j = json.load(open(sys.argv[1]))
    j["_build"] = _cabi.load().thmr_build_info().decode()
    j["_note"] = "separate rocprofv3 --pmc passes (--kernel-trace only) of scripts/r5_pmc_workload.py; FETCH_SIZE doubled per the gfx950 correction"
    json.dump(j, open(sys.argv[1], "w"), indent=1)
    for k, e in j.items():
        if isinstance(e, dict): print(k, json.dumps(e)[:1500])
except Exception as ex:
    print("pmc parse failed:", ex)
PY
      rm -rf $O/pmc_$wl ;;
    run)
      timeout 1500 bash -c "$arg" > $O/run$n.log 2>&1; echo "run rc=$?"; tail -25 $O/run$n.log | cut -c1-300 ;;
    *) echo "unknown step $step" ;;
  esac
  stamp "$step"
done
echo "[$NAME] total t=$(( $(date +%s) - t0 )) s"
