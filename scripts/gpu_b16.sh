#!/bin/bash
# kernel-level durations of the B = 16 step (rocprofv3 --kernel-trace --stats), to compare with the stand-alone GEMM timings
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for B in 16 32; do
  rm -rf gpurun_out/prof_b$B
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_b$B" -o r -- python "$R/bench.py" --batch $B --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/prof_b$B.log 2>&1
  find gpurun_out/prof_b$B -type f -name "*kernel_trace*" -delete
  echo "B=$B"; head -8 gpurun_out/prof_b$B/r_kernel_stats.csv | cut -c1-200
done
