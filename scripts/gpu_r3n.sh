#!/bin/bash
# Round 3, session N: verification of the final build + the committed artefacts (bench lines with rocprofv3 stats of the same commands, batch sweep)
set -u
O=gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert|Error" $O/pytest_gpu.log | head -12; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_full" -o p -- python "$R/bench.py" --no-cpu-baseline) > $O/prof_full.log 2>&1
timeout 600 python bench.py --workload vit --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err; cut -c1-200 $O/bench_vit.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_vit" -o p -- python "$R/bench.py" --workload vit --no-cpu-baseline) > $O/prof_vit.log 2>&1
find $O/prof_vit $O/prof_full -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
grep '^{' $O/prof_full.log | cut -c1-160; echo "t=$(( $(date +%s) - t0 ))"
: > $O/batch_sweep.jsonl
for B in 1 2 3 4 6 7 8 9 10 12 16 17 21 24 32 128; do
  timeout 300 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'batch': d['config']['batch_per_gpu'], 'crops_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'step_ms_median': d['step_ms']['median'], 'facade': d.get('facade'), 'all_gemm_tflops': r['all_gemm_achieved'], 'classes_ms': r['classes_ms_per_step']}))" >> $O/batch_sweep.jsonl
done
python -c "
import json
for l in open('$O/batch_sweep.jsonl'):
    d=json.loads(l); print(d['batch'], d['crops_per_s'], d['ms_per_step'], d['facade']['ms_per_call'] if d['facade'] else None)"
echo "t=$(( $(date +%s) - t0 ))"
timeout 200 python scripts/lbs_bench.py 1 64 512 2>/dev/null | grep LBS | tee $O/lbs_bench.log
timeout 200 python scripts/graph_latency.py 1 2 4 6 2>/dev/null | tail -1 > $O/latency.json; cut -c1-500 $O/latency.json
find $O -type f -size +8M -delete
echo "total t=$(( $(date +%s) - t0 ))"
