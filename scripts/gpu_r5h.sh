#!/bin/bash
# Round 5, session H: two direct same-box comparisons.  (1) build_ab/va = the current source with ROUND 4's attention_b16.hip (208 / 144-byte
# image rows, 48 % of its LDS cycles bank conflicts) against the current build (conflict-free images): does the layout cost time?
# (2) build_ab/vd = the persistent fc2 publishing from inside the K loop again, its operands parked in LDS, against r4 and against current.
set -u
O=gpurun_out/r5h; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
run() { timeout 400 scripts/ab_same_box.sh "$1" "$2" $O/ab_$3.json --batch 64 --reps 7 --iters 10 > $O/ab_$3.log 2> $O/ab_$3.err; tail -1 $O/ab_$3.err | cut -c1-200; }
run build_ab/va/libtokenhmr_hip.so current oldattn_vs_current
run build_ab/r4/libtokenhmr_hip.so build_ab/vd/libtokenhmr_hip.so r4_vs_vd
run current build_ab/vd/libtokenhmr_hip.so current_vs_vd
run build_ab/va/libtokenhmr_hip.so current oldattn_vs_current_again
python - <<'PY'
import json
for n in ("oldattn_vs_current", "r4_vs_vd", "current_vs_vd", "oldattn_vs_current_again"):
    try:
        j = json.load(open(f"gpurun_out/r5h/ab_{n}.json")); d = j["classes_B_minus_A_ms"]
        print(n, "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "| attn", d["attention"], "fc2", d["gemm_fc2"], "fc1", d["gemm_fc1"], "ln", d["layernorm"], "| bit-identical verts", j["outputs_bit_identical"]["pred_vertices"])
    except Exception as e:
        print(n, "parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
