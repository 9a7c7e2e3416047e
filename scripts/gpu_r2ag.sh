#!/bin/bash
# Round-2 run AG: the committed pair with the FINAL build: default bench line + rocprofv3 kernel stats of the same command on the same box
set -u
mkdir -p gpurun_out/r2ag; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 900 python bench.py > gpurun_out/r2ag/bench_full.json 2> gpurun_out/r2ag/bench_full.err
rm -rf gpurun_out/r2ag/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r2ag/prof" -o k -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > gpurun_out/r2ag/prof.log 2>&1
f=$(find gpurun_out/r2ag/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2ag/kernel_stats.csv; rm -rf gpurun_out/r2ag/prof
cut -c1-200 gpurun_out/r2ag/bench_full.json; head -8 gpurun_out/r2ag/kernel_stats.csv | cut -c1-150
