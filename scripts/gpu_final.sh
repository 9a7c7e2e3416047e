#!/bin/bash
# Round-end evidence run: all gpu tests, smoke(), default bench, vit-only bench, kernel stats csv.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 1200 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -15 > gpurun_out/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
timeout 600 python bench.py --workload vit --no-cpu-baseline > gpurun_out/bench_vit.log 2>&1
rm -rf gpurun_out/prof4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof4" -o r1 -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/prof4.log 2>&1
find gpurun_out/prof4 -type f -name "*kernel_trace*" -delete
tail -4 gpurun_out/tests.log; tail -3 gpurun_out/smoke.log; grep '^{' gpurun_out/bench.log | cut -c1-300; grep '^{' gpurun_out/bench_vit.log | cut -c1-300
