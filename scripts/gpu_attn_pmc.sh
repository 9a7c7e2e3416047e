#!/bin/bash
# PMC passes on the attention kernel (B = 64: 1024 workgroups) -> profiles/r1_pmc_attention.json via scripts/pmc_attn_to_json.py
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
rm -rf gpurun_out/pmc_attn
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/gpurun_out/pmc_attn/sq" -o p -- python "$R/scripts/attn_bench.py" 2) > gpurun_out/pmc_attn.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d "$R/gpurun_out/pmc_attn/sq2" -o p -- python "$R/scripts/attn_bench.py" 2) >> gpurun_out/pmc_attn.log 2>&1
find gpurun_out/pmc_attn -type f -size +6M -delete
ls gpurun_out/pmc_attn/*; tail -3 gpurun_out/pmc_attn.log
