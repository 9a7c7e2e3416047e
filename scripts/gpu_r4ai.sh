#!/bin/bash
# Round 4, session AI: the split3 GEMM's stage-image rotation matched to gfx950's ds_read_b128 lane groups (rot = (0, 2, 0, 2) per row quad
# instead of (0, 1, 2, 3)): op tests, LDS counters of the fc1 kernel, op-level and engine timing
set -u
O=gpurun_out/r4ai; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "split3" 2>&1 | tail -2 | tee $O/pytest_split3.log
echo "t=$(( $(date +%s) - t0 ))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$R/$O/pmc/lds" -o p -- python $R/scripts/r4_pmc_workload.py gemm) > $O/pmc.log 2>&1
find $O/pmc -type f ! -name '*counter_collection.csv' -exec rm -f {} + 2>/dev/null
for d in $O/pmc/*; do f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && mv "$f" "$d/p_counter_collection.csv"; done
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_lds.json 4 "gemm_split16_kernel<4, 2, false, false>" "gemm_split16_kernel<4, 5, false, false>" "gemm_split16_kernel<4, 4, true, true>" > /dev/null 2>&1
python - <<'PY'
import json
try:
    for k, e in json.load(open("gpurun_out/r4ai/pmc_lds.json")).items():
        print(k, {x: round(v, 1) for x, v in e.items() if x.startswith("SQ_")}, e.get("profiled_dur_us"))
except Exception as ex:
    print("pmc parse failed", ex)
PY
rm -rf $O/pmc
for rep in 1 2; do timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"' | cut -c1-600 | tee -a $O/mode_b64.log; done
echo "total t=$(( $(date +%s) - t0 ))"
