#!/bin/bash
# Round 4, session V: duty-cycle probe — the split3 fc1 GEMM (and the exact-fp32 one as the control) back to back and with idle gaps of
# 1x / 3x its own duration between launches: does the SAME kernel run faster after a pause?
set -u
O=gpurun_out/r4v; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 300 python scripts/power_duty_probe.py 2> $O/probe.err | tee $O/power_duty_probe.log | cut -c1-600
tail -3 $O/probe.err
