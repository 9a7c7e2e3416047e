#!/bin/bash
# Round 4, session AH: LDS counters of the bf16-pipe attention kernel (bank conflicts, LDS instruction count / active time) — one --pmc pass
set -u
O=gpurun_out/r4ah; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for p in "lds:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" "lds2:SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/r4_pmc_workload.py gemm) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
find $O/pmc -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
for d in $O/pmc/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_lds.json 4 "vit_attention_b16_kernel<3, true>" "vit_attention_persistent_kernel<0, true>" "gemm_split16_kernel<4, 2, false, false>" > /dev/null 2> $O/pmc_json.err; tail -2 $O/pmc_json.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4ah/pmc_lds.json"))
    for k, e in j.items():
        print(k, {x: (round(v, 1) if isinstance(v, float) else v) for x, v in e.items() if x.startswith("SQ_") or x == "profiled_dur_us"})
except Exception as ex:
    print("parse failed", ex); print(open("gpurun_out/r4ah/pmc.log").read()[-1500:])
PY
rm -rf $O/pmc
