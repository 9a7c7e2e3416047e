#!/bin/bash
# Round 4, session R: csrc/attention_b16.hip as at most 512 workgroups walking the items (block pipeline across item boundaries):
# operator tests, stand-alone timing, engine A/B in the split3 mode (experiments build, THMR_ATTN_B16=0 / 1)
set -u
O=gpurun_out/r4r; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -k "attention" > $O/pytest_attention.log 2>&1; echo "rc=$?" >> $O/pytest_attention.log
grep -E "attention b16|passed|failed|rc=|Error|assert" $O/pytest_attention.log | cut -c1-300 | tail -30; echo "t=$(( $(date +%s) - t0 ))"
for b in 64 32 16; do timeout 200 python scripts/attn_b16_bench.py $b 50 2> $O/attn_bench_b$b.err | cut -c1-1500 | tee -a $O/attn_b16_bench.jsonl; done
echo "t=$(( $(date +%s) - t0 ))"
for ab in 0 1 0 1; do
  THMR_LIB=exp THMR_ATTN_B16=$ab timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64_ab$ab.err | grep -E '"mode": "split3"' | cut -c1-600 | sed "s/^/attn_b16=$ab /" | tee -a $O/mode_b64_attn_ab.log
done
echo "total t=$(( $(date +%s) - t0 ))"
