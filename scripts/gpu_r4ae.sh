#!/bin/bash
# Round 4, session AE: start-up de-phasing of the two co-resident workgroups of the bf16-pipe attention (experiments build knob)
set -u
O=gpurun_out/r4ae; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for us in 0 5 10 20 40 0; do
  THMR_LIB=exp THMR_ATTN_B16_DEPHASE_US=$us timeout 200 python scripts/attn_b16_bench.py 64 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('dephase_us=$us', {k: j[k]['us'] for k in ('b16 qt=3','b16 qt=3 split3_out','b16 qt=1 split3_out')})" | tee -a $O/attn_b16_dephase.log
done
