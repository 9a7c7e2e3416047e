#!/bin/bash
# Round 3, session S: counters for the split3 GEMM (fc1 / fc2 shapes at 64 crops)
set -u
O=gpurun_out/r3s; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "lds:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc/${p%%:*}" -o p -- python $R/scripts/split3_pmc_workload.py) >> $O/pmc.log 2>&1
  echo "pass ${p%%:*} rc=$?"
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
python scripts/pmc_kernels_to_json.py $O/pmc $O/pmc_split3.json 4 "gemm_split3_kernel<2, 4, 2, 2, 2>" "gemm_split3_kernel<2, 4, 2, 2, 4>" 2>&1 | tail -3
cat $O/pmc_split3.json | head -80
find $O -type f -size +12M -delete
tail -5 $O/pmc.log
echo "total t=$(( $(date +%s) - t0 ))"
