#!/bin/bash
# One gpurun call: parity tests, bench, rocprof kernel stats. Everything is logged under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
STAGE="${1:-all}"
echo "== rocminfo ==" > gpurun_out/env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; free -g | head -2; lscpu | grep "Model name") >> gpurun_out/env.log 2>&1
if [[ "$STAGE" == "all" || "$STAGE" == "ops" ]]; then
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -x 2>&1 | tail -40 > gpurun_out/test_ops.log
  echo "ops rc=${PIPESTATUS[0]}" >> gpurun_out/test_ops.log
fi
if [[ "$STAGE" == "all" || "$STAGE" == "model" ]]; then
  timeout 1200 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -s 2>&1 | tail -80 > gpurun_out/test_model.log
  echo "model rc=${PIPESTATUS[0]}" >> gpurun_out/test_model.log
fi
if [[ "$STAGE" == "all" || "$STAGE" == "bench" ]]; then
  timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
  echo "bench rc=$?" >> gpurun_out/bench.log
  timeout 600 python bench.py --steps 5 --warmup 2 --workload vit --no-cpu-baseline > gpurun_out/bench_vit.log 2>&1
fi
if [[ "$STAGE" == "all" || "$STAGE" == "prof" ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/prof.log 2>&1
  find gpurun_out/prof -name "*kernel_stats*" | head -3 >> gpurun_out/prof.log
  # keep only the small stats files (the trace itself can be large)
  find gpurun_out/prof -type f ! -name "*stats*" -size +8M -delete
fi
tail -5 gpurun_out/*.log
