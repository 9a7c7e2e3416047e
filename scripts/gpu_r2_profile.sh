#!/bin/bash
# Round-2 committed-profile run: the whole GPU suite, the default bench, its rocprofv3 kernel stats, PMC passes on the dominant
# GEMM (SQ / FETCH_SIZE / WRITE_SIZE / L2, separate passes) and on the attention kernel.  TAG names the outputs.
set -u
TAG="${1:-r2}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc gpurun_out/pmc_attn
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof" -o k -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > gpurun_out/${TAG}_prof.log 2>&1
find gpurun_out/${TAG}_prof -type f -name "*kernel_trace*" -delete
V=128x160
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/gpurun_out/${TAG}_pmc/sq" -o p -- python "$R/scripts/gemm_bench.py" 2) > gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/${TAG}_pmc/fetch" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/${TAG}_pmc/write" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && GEMM_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/gpurun_out/${TAG}_pmc/l2" -o p -- python "$R/scripts/gemm_bench.py" 2) >> gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/gpurun_out/pmc_attn/sq" -o p -- python "$R/scripts/attn_bench.py" 2) > gpurun_out/pmc_attn.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d "$R/gpurun_out/pmc_attn/sq2" -o p -- python "$R/scripts/attn_bench.py" 2) >> gpurun_out/pmc_attn.log 2>&1
find gpurun_out -type f -size +6M -delete
python scripts/pmc_to_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc_gemm.json > gpurun_out/${TAG}_pmc_summary.log 2>&1
python scripts/pmc_attn_to_json.py gpurun_out/${TAG}_pmc_attention.json >> gpurun_out/${TAG}_pmc_summary.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log; cut -c1-400 gpurun_out/${TAG}_bench_full.json; cat gpurun_out/${TAG}_pmc_summary.log | tail -30
