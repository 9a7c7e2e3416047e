#!/bin/bash
# Round 4, session AA: calibration — the vendor GEMM behind torch.mm (fp32 and ONE bf16 product) on the ViT shapes of a 64-crop batch
set -u
O=gpurun_out/r4aa; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 400 python scripts/library_gemm_reference.py 2> $O/err.log | tee $O/library_gemm_reference.log | cut -c1-300
tail -3 $O/err.log
