#!/bin/bash
# Round 3, session R: the engine's split3 mode (ViT GEMMs on the bf16 matrix pipe): parity tests, then B = 64 timing in both modes
set -u
O=gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "split3 or b64_tokens" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -a "golden\|split3\|passed\|failed\|Error\|error" $O/tests.log | tail -20
timeout 600 python scripts/mode_bench.py > $O/mode_bench.log 2>&1; echo "bench rc=$?"; tail -12 $O/mode_bench.log
echo "total t=$(( $(date +%s) - t0 ))"
