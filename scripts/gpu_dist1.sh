#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1
echo "rc=$?" >> gpurun_out/bench_dist1.log
grep -v "amdgpu.ids" gpurun_out/bench_dist1.log | tail -5 | cut -c1-900
