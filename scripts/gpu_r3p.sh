#!/bin/bash
# Round 3, session P: counters for the round's new kernels (key-split attention, 16x16x4 qkv GEMM at one crop; split-K big tiles at 8 crops) + kernel stats at 8 crops
set -u
O=gpurun_out/r3p; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for B in 1 8; do
  for p in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d "$R/$O/pmc_b$B/${p%%:*}" -o p -- python $R/scripts/b1_trace.py $B) >> $O/pmc_b$B.log 2>&1
  done
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_b8" -o p -- python "$R/scripts/b1_trace.py" 8) > $O/prof_b8.log 2>&1
find $O/prof_b8 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -8 $O/prof_b8/p_kernel_stats.csv | cut -c1-180
ls -la $O/pmc_b1/*/ | head; find $O -type f -size +12M -delete
echo "total t=$(( $(date +%s) - t0 ))"
