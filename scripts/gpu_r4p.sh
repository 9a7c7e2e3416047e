#!/bin/bash
# Round 4, session P: the committed profiles of the round's build — rocprofv3 kernel stats of the bench in both modes, PMC passes
# (separate passes, --kernel-trace only) on the persistent split3 GEMMs, the exact-fp32 fc1 GEMM, LayerNorm / attention with split3
# output, and the half-chip round of the per-tile split3 kernel (shader clock = GRBM_GUI_ACTIVE / duration)
set -u
O=gpurun_out/r4p; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32" -o p -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_f32.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split3" -o p -- python "$R/bench.py" --vit-gemm split3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $O/prof_split3.log 2>&1
find $O/prof_f32 $O/prof_split3 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -8 $O/prof_split3/*/*kernel_stats.csv 2>/dev/null | cut -c1-220 || head -8 $O/prof_split3/*kernel_stats.csv | cut -c1-220
grep -h '"value"' $O/prof_f32.log $O/prof_split3.log | cut -c1-160
echo "t=$(( $(date +%s) - t0 ))"
for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "lds:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r4_pmc_workload.py all) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
python - <<'PY'
import csv, glob, collections
# shader clock of the per-tile split3 kernel with 128 / 256 tiles: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration
for f in glob.glob("gpurun_out/r4p/pmc/sq/**/*counter_collection.csv", recursive=True):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "gemm_split3_kernel" in r["Kernel_Name"] or "gemm_split3_persist_kernel" in r["Kernel_Name"] or "gemm_f32_kernel" in r["Kernel_Name"]:
            d = rows[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            d["grid"] = int(r["Grid_Size"]); d["wg"] = int(r["Workgroup_Size"]); d["name"] = r["Kernel_Name"][:60]
    agg = collections.defaultdict(list)
    for i, d in sorted(rows.items()):
        if "GRBM_GUI_ACTIVE" in d:
            agg[(d["name"], d["grid"] // d["wg"])].append((d["dur"], d["GRBM_GUI_ACTIVE"] / 8 / d["dur"] * 1e-3, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    out = open("gpurun_out/r4p/clock_by_kernel.log", "w")
    for k, v in agg.items():
        v = v[-4:]
        line = f"{k[0]:60s} workgroups {k[1]:5d}  dur_us {sum(x[0] for x in v)/len(v):8.1f}  shader_clock_GHz {sum(x[1] for x in v)/len(v):.3f}  mfma_util {sum(x[2] for x in v)/len(v):.3f}"
        print(line); out.write(line + "\n")
PY
find $O -type f ! -name '*counter_collection.csv' ! -name '*.log' ! -name '*.json' ! -name '*kernel_stats.csv' -delete 2>/dev/null
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_r4.json 4 "gemm_split3_persist_kernel<5, 0>" "gemm_split3_persist_kernel<2, 2>" "gemm_split3_persist_kernel<4, 0>" "gemm_f32_kernel" "ln_split3_kernel" "ln_wave_kernel" "vit_attention_persistent_kernel<0, true>" "vit_attention_persistent_kernel<0, false>" 2>&1 | tail -3
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4p/pmc_r4.json"))
for k, e in j.items():
    print(k, {x: e.get(x) for x in ("profiled_dur_us", "mfma_util_profiled", "traffic_bytes", "sq_wait_any_frac_of_wave_cycles", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum")})
PY
find $O -type f -size +12M -delete
tail -3 $O/pmc.log
echo "total t=$(( $(date +%s) - t0 ))"
