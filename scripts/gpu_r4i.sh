#!/bin/bash
# Round 4, session I: rocprofv3 kernel stats of the frames -> crops -> forward -> evaluator pipeline (both modes); the forced hand-over
# timeout recovery test and the evaluator tests
set -u
O=gpurun_out/r4i; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_evaluator.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_pipe_f32" -o p -- python "$R/bench.py" --workload pipeline --steps 10 --warmup 2 --no-cpu-baseline) > $O/pipe_f32.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_pipe_split3" -o p -- python "$R/bench.py" --workload pipeline --vit-gemm split3 --steps 10 --warmup 2 --no-cpu-baseline) > $O/pipe_split3.log 2>&1
find $O/prof_pipe_f32 $O/prof_pipe_split3 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
grep -h '"value"' $O/pipe_f32.log $O/pipe_split3.log | cut -c1-1200
grep -E "crop_|eval_|copyBuffer|regress" $O/prof_pipe_f32/p_kernel_stats.csv | cut -c1-200
echo "total t=$(( $(date +%s) - t0 ))"
