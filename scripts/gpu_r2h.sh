#!/bin/bash
# round 2, run H: K-chunked / stream-K tiled GEMM: parity, B = 64 regression check, batch sweep
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
(timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm_property.py tests/test_gpu_model.py tests/test_tokenizer.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -8) > gpurun_out/r2h_pytest.log
timeout 300 python scripts/gemm_bench.py 5 > gpurun_out/r2h_gemm_bench.log 2>&1
: > gpurun_out/r2h_batch_sweep.jsonl
for B in 64 8 16 32 128; do
  timeout 600 python bench.py --batch $B --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'batch': d['config']['batch_per_gpu'], 'crops_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'all_gemm_tflops': r['all_gemm_achieved'], 'fc1_frac': r['frac'], 'classes_ms': r['classes_ms_per_step'], 'parity': d.get('parity')}))" >> gpurun_out/r2h_batch_sweep.jsonl
done
tail -6 gpurun_out/r2h_pytest.log; tail -12 gpurun_out/r2h_gemm_bench.log | cut -c1-200; cut -c1-420 gpurun_out/r2h_batch_sweep.jsonl
