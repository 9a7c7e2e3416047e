#!/bin/bash
# Round 4, session AB: stress / determinism test of the bf16-pipe attention (three runs of it), attention tests again
set -u
O=gpurun_out/r4ab; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for rep in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention_b16" 2>&1 | tail -2 | tee -a $O/pytest_attention_b16_stress.log; done
