#!/bin/bash
# the committed pair: default bench line + rocprofv3 kernel stats of the same command (no PMC passes)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 200 python bench.py --steps 8 --warmup 3 > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof3
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof3" -o r1 -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/prof3.log 2>&1
find gpurun_out/prof3 -type f -name "*kernel_trace*" -delete
grep '^{' gpurun_out/bench.log | cut -c1-120; head -3 gpurun_out/prof3/r1_kernel_stats.csv | cut -c60-160
