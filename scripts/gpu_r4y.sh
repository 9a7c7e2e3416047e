#!/bin/bash
# Round 4, session Y: attention_b16 with its split3 output assembled in the dead V^T image and streamed out as consecutive 16-byte chunks
set -u
O=gpurun_out/r4y; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -k "attention" > $O/pytest_attention.log 2>&1; echo "rc=$?" >> $O/pytest_attention.log
grep -E "passed|failed|rc=|Error|assert" $O/pytest_attention.log | cut -c1-300 | tail -8; echo "t=$(( $(date +%s) - t0 ))"
for b in 64 32 16 8; do timeout 200 python scripts/attn_b16_bench.py $b 50 2> $O/attn_bench_b$b.err | cut -c1-1500 | tee -a $O/attn_b16_bench.jsonl; done
for rep in 1 2; do timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"' | cut -c1-600 | tee -a $O/mode_b64.log; done
echo "total t=$(( $(date +%s) - t0 ))"
