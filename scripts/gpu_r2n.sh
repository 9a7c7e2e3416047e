#!/bin/bash
# Round-2 run N: diagnostics — attention phase timeline + TG-slot de-phasing, ring-GEMM ablations at M = 192 * {1,2,4,6}
mkdir -p gpurun_out/r2n
timeout 300 build_ab/attn_timeline gpurun_out/r2n/attn_timeline.csv > gpurun_out/r2n/attn_timeline.log 2>&1; echo "rc=$?" >> gpurun_out/r2n/attn_timeline.log
cat gpurun_out/r2n/attn_timeline.log
timeout 400 build_ab/ring_ablation > gpurun_out/r2n/ring_ablation.log 2>&1; echo "rc=$?" >> gpurun_out/r2n/ring_ablation.log
cat gpurun_out/r2n/ring_ablation.log
