#!/bin/bash
# Round 5, session J: (1) the persistent kernels' LDS-DMA destinations as {scalar} + {immediate} (frees ~18 SGPRs; the bias + residual
# instantiation on a row-major A loses its scratch reloads) against the committed build; (2) which GEMM classes run the persistent kernel:
# mask 8 (fc2 only, what ships) against 10 (+ proj) and 9 (+ qkv), experiments build, same box, interleaved.
set -u
O=gpurun_out/r5j; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
run() {  # name A B [extra...]
  n=$1; A=$2; B=$3; shift 3
  timeout 300 python scripts/ab_same_box.py --a $A --b $B --out $O/ab_$n.json --batch 64 --reps 5 --iters 10 "$@" > $O/ab_$n.log 2> $O/ab_$n.err
  python - $n <<'PY'
import json, sys
try:
    j = json.load(open(f"gpurun_out/r5j/ab_{sys.argv[1]}.json"))
    d = j["classes_B_minus_A_ms"]
    print(sys.argv[1], "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "|", {k: d[k] for k in ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2")}, "| bit-identical verts", j["outputs_bit_identical"].get("pred_vertices"))
except Exception as e:
    print(sys.argv[1], "parse failed", e)
PY
}
run head_vs_current build_ab/head/libtokenhmr_hip.so current
run mask8_vs_10 exp exp --a-env THMR_SPLIT3_PERSIST_MASK=8 --b-env THMR_SPLIT3_PERSIST_MASK=10
run mask8_vs_9 exp exp --a-env THMR_SPLIT3_PERSIST_MASK=8 --b-env THMR_SPLIT3_PERSIST_MASK=9
run mask8_vs_11 exp exp --a-env THMR_SPLIT3_PERSIST_MASK=8 --b-env THMR_SPLIT3_PERSIST_MASK=11
echo "total t=$(( $(date +%s) - t0 ))"
