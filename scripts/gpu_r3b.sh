#!/bin/bash
# Round 3, session B: LBS skin kernel rewrite (scalar-load bone matrices, register-blocked regression, one arrival per pass),
# cross-XCD hand-off micro-benchmark (persistent-ViT feasibility), fc1-only event overhead
set -u
O=gpurun_out/r3b; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error|error" $O/pytest_gpu.log | tail -5; echo "t=$(( $(date +%s) - t0 ))"
timeout 200 python scripts/lbs_bench.py 1 8 64 512 2>/dev/null | grep LBS > $O/lbs_bench.log; cat $O/lbs_bench.log
mkdir -p build_ab
hipcc --offload-arch=gfx950 -O3 -o build_ab/xcd_handoff scripts/micro/xcd_handoff.hip 2>/dev/null
timeout 120 build_ab/xcd_handoff 200 > $O/xcd_handoff.log 2>&1; cat $O/xcd_handoff.log
echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
timeout 300 python bench.py --batch 1 --no-cpu-baseline --steps 40 --warmup 5 > $O/bench_b1.json 2> $O/bench_b1.err; cut -c1-200 $O/bench_b1.json
timeout 300 python bench.py --batch 8 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_b8.json 2> $O/bench_b8.err; cut -c1-200 $O/bench_b8.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_lbs" -o p -- python "$R/scripts/lbs_bench.py" 64 512) > $O/prof_lbs.log 2>&1
find $O/prof_lbs -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
cat $O/prof_lbs/*/p_kernel_stats.csv 2>/dev/null | cut -c1-200 | head -8
echo "total t=$(( $(date +%s) - t0 ))"
