#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
: > gpurun_out/batch_sweep.log
for B in ${SWEEP_B:-1 2 4 8 16 32 64 128 256}; do
  timeout 600 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'batch': d['config']['batch_per_gpu'], 'crops_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'all_gemm_tflops': r['all_gemm_achieved'], 'classes_ms': r['classes_ms_per_step'], 'patch_embed_hbm': r.get('patch_embed_hbm'), 'lbs_hbm': r.get('lbs_hbm')}))" >> gpurun_out/batch_sweep.log
done
cat gpurun_out/batch_sweep.log | cut -c1-400
