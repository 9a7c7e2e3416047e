#!/bin/bash
# Round 4, session X: which neighbour makes fc1 7 % slower in layer order than in a loop of its own?
set -u
O=gpurun_out/r4x; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 400 python scripts/fc1_neighbour_probe.py 2> $O/probe.err | tee $O/fc1_neighbour_probe.log | cut -c1-300
tail -3 $O/probe.err
