#!/bin/bash
# Round 5, session M: the attention kernel's q as split3 pieces written by the qkv GEMM's epilogue (GemmArgs::cs_cols, attention_b16.hip QSP):
# the GPU suite, then same-box interleaved A/B of the experiments build with THMR_ATTN_QSP=0 / 1 at 64 / 32 / 8 crops, then r4 vs current.
set -u
O=gpurun_out/r5m; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -12; grep -E "^E  " $O/pytest_gpu.log | head -20; echo "t=$(( $(date +%s) - t0 ))"
run() {  # name A B [extra...]
  n=$1; A=$2; B=$3; shift 3
  timeout 300 python scripts/ab_same_box.py --a $A --b $B --out $O/ab_$n.json --reps 5 --iters 10 "$@" > $O/ab_$n.log 2> $O/ab_$n.err
  python - $n <<'PY'
import json, sys
try:
    j = json.load(open(f"gpurun_out/r5m/ab_{sys.argv[1]}.json"))
    d = j["classes_B_minus_A_ms"]
    print(sys.argv[1], "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "|", {k: d[k] for k in ("gemm_qkv", "attention", "gemm_proj", "gemm_fc1", "gemm_fc2")}, "| bit-identical", all(j["outputs_bit_identical"].values()))
except Exception as e:
    print(sys.argv[1], "parse failed", e)
PY
}
run qsp0_vs_1_b64 exp exp --batch 64 --a-env THMR_ATTN_QSP=0 --b-env THMR_ATTN_QSP=1
run qsp0_vs_1_b32 exp exp --batch 32 --a-env THMR_ATTN_QSP=0 --b-env THMR_ATTN_QSP=1
run qsp0_vs_1_b8 exp exp --batch 8 --a-env THMR_ATTN_QSP=0 --b-env THMR_ATTN_QSP=1
run r4_vs_current_b64 build_ab/r4/libtokenhmr_hip.so current --batch 64
echo "total t=$(( $(date +%s) - t0 ))"
