#!/bin/bash
# Round-2 final verification: whole GPU suite, smoke(), the default bench line, the driver's launcher form at world size 1 over RCCL
set -u
mkdir -p gpurun_out/r2z; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2z/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2z/pytest_gpu.log
grep -E "passed|failed|rc=" gpurun_out/r2z/pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r2z/smoke.log 2>&1; tail -2 gpurun_out/r2z/smoke.log
timeout 900 python bench.py > gpurun_out/r2z/bench_full.json 2> gpurun_out/r2z/bench_full.err; cut -c1-300 gpurun_out/r2z/bench_full.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2z/bench_dist1.json 2> gpurun_out/r2z/bench_dist1.err
echo "rc=$?"; grep crops_per_sec gpurun_out/r2z/bench_dist1.json | cut -c1-700
