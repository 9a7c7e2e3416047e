#!/bin/bash
# Round 4, session W: cold-weights probe — the four split3 GEMMs in layer order over 1 / 2 / 8 / 32 distinct weight sets
set -u
O=gpurun_out/r4w; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 400 python scripts/cold_weights_probe.py 2> $O/probe.err | tee $O/cold_weights_probe.log | cut -c1-400
tail -3 $O/probe.err
