#!/bin/bash
# Round 3, session AH: verification of the build with the split3 mode: full GPU suite, smoke, bench lines (default and --vit-gemm split3) each with rocprofv3 stats of the same command, ViT-only line
set -u
O=gpurun_out/r3ah; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_full" -o p -- python "$R/bench.py" --no-cpu-baseline) > $O/prof_full.log 2>&1
timeout 600 python bench.py --vit-gemm split3 --no-cpu-baseline > $O/bench_split3.json 2> $O/bench_split3.err; cut -c1-200 $O/bench_split3.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split3" -o p -- python "$R/bench.py" --vit-gemm split3 --no-cpu-baseline --no-extras) > $O/prof_split3.log 2>&1
timeout 600 python bench.py --workload vit --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err; cut -c1-200 $O/bench_vit.json
find $O/prof_split3 $O/prof_full -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
head -6 $O/prof_split3/*kernel_stats.csv | cut -c1-200
find $O -type f -size +8M -delete
echo "total t=$(( $(date +%s) - t0 ))"
