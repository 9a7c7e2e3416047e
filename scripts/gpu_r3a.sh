#!/bin/bash
# Round 3, session A: whole GPU suite + smoke + default bench + ViT-only bench (configs[1]) with rocprofv3 stats, PMC passes for the
# LBS kernels (B = 512) and the persistent decoder kernel (B = 64), batch sweep, B = 1 knobs (ring threshold, cooperative decoder launch)
set -u
O=gpurun_out/r3a; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -2; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-260 $O/bench_full.json; echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python bench.py --workload vit --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err; cut -c1-200 $O/bench_vit.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_vit" -o p -- python "$R/bench.py" --workload vit --steps 8 --warmup 3 --no-cpu-baseline --no-extras) > $O/prof_vit.log 2>&1
find $O/prof_vit $O/prof_full -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
echo "t=$(( $(date +%s) - t0 ))"
# PMC: separate passes (SQ / FETCH / WRITE), counters only with --kernel-trace
for what in "lbs:scripts/lbs_bench.py 512" "dec:scripts/head_bench.py 64"; do
  tag=${what%%:*}; cmd=${what#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_$tag/sq" -o p -- python $R/$cmd) > $O/pmc_$tag.log 2>&1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$R/$O/pmc_$tag/sq2" -o p -- python $R/$cmd) >> $O/pmc_$tag.log 2>&1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_$tag/fetch" -o p -- python $R/$cmd) >> $O/pmc_$tag.log 2>&1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/$O/pmc_$tag/write" -o p -- python $R/$cmd) >> $O/pmc_$tag.log 2>&1
done
find $O/pmc_lbs $O/pmc_dec -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
find $O -type f -size +8M -delete
echo "pmc done t=$(( $(date +%s) - t0 ))"
# batch sweep with the round-3 build
: > $O/batch_sweep.jsonl
for B in 1 8 16 32; do
  timeout 300 python bench.py --batch $B --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'batch': d['config']['batch_per_gpu'], 'crops_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'all_gemm_tflops': r['all_gemm_achieved'], 'classes_ms': r['classes_ms_per_step'], 'facade': d.get('facade')}))" >> $O/batch_sweep.jsonl
done
cut -c1-200 $O/batch_sweep.jsonl; echo "t=$(( $(date +%s) - t0 ))"
# B = 1..4 knobs
for env in "" "THMR_RING_MAX_TILES=320"; do
  echo "== $env" >> $O/b1_knobs.log
  env $env timeout 300 python scripts/graph_latency.py 1 2 4 2>/dev/null | tail -1 >> $O/b1_knobs.log
done
cat $O/b1_knobs.log | cut -c1-400
echo "== head default vs coop" > $O/head_coop.log
timeout 200 python scripts/head_bench.py 1 8 64 2>/dev/null | grep -i "head\|wall" >> $O/head_coop.log
THMR_DEC_COOP=1 timeout 200 python scripts/head_bench.py 1 8 64 2>/dev/null | grep -i "head\|wall" >> $O/head_coop.log
tail -8 $O/head_coop.log | cut -c1-300
echo "total t=$(( $(date +%s) - t0 ))"
