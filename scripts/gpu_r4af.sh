#!/bin/bash
# Round 4, session AF: the bf16-pipe attention with a block's V loads issued one k step early (into the registers the Q prefetch leaves free)
set -u
O=gpurun_out/r4af; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention_b16" 2>&1 | tail -2
for b in 64 32; do timeout 200 python scripts/attn_b16_bench.py $b 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('B', j['B'], {k: j[k]['us'] for k in j if isinstance(j[k], dict)})" | tee -a $O/attn_b16_early_v.log; done
