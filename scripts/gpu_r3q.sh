#!/bin/bash
# Round 3, session Q: first run of the split3 GEMM (fp32 operands as three bf16 pieces on the bf16 matrix pipe): parity tests, then rate + error
set -u
O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "split3" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout 600 python scripts/split3_bench.py --crops 64 > $O/split3_bench_b64.jsonl 2> $O/split3_bench_b64.err; echo "bench rc=$?"; cat $O/split3_bench_b64.jsonl; tail -5 $O/split3_bench_b64.err
echo "total t=$(( $(date +%s) - t0 ))"
