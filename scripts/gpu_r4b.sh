#!/bin/bash
# Round 4, session B: persistent split3 GEMM bit-identity tests (threshold of the fp64 slice check fixed), attention split3 output through
# permlane16 swaps (tests + timing), power / clock probe of the GEMM kernels, engine timing at 64 / 32 crops
set -u
O=gpurun_out/r4b; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "persistent or attention" > $O/pytest_ops.log 2>&1; echo "rc=$?" >> $O/pytest_ops.log
tail -6 $O/pytest_ops.log | cut -c1-300; echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/power_probe.py --secs 2.5 > $O/power_probe.jsonl 2> $O/power_probe.err; cat $O/power_probe.jsonl; tail -3 $O/power_probe.err
echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/mode_bench.py 64 10 2> $O/mode_b64.err | grep -E '"mode": "split3"|max_abs' | cut -c1-600 | tee $O/mode_b64.log
timeout 300 python scripts/mode_bench.py 32 10 2> $O/mode_b32.err | grep -E '"mode": "split3"' | cut -c1-600 | tee $O/mode_b32.log
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "split3_mode or small_trained" > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
tail -4 $O/pytest_model.log | cut -c1-300
echo "total t=$(( $(date +%s) - t0 ))"
