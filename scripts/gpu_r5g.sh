#!/bin/bash
# Round 5, session G: is the +0.5 ms of the persistent fc2 the epoch protocol's atomics or the code around them?  build_ab/vb = the shipped
# source with round 4's 0 / 1 flags (consumer clears, no arrival) but the SHIPPED publish point; build_ab/vc = epoch without atomics.
set -u
O=gpurun_out/r5g; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for v in $(ls build_ab | grep -v '^r4$'); do
  timeout 400 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so build_ab/$v/libtokenhmr_hip.so $O/ab_r4_vs_$v.json --batch 64 --reps 5 --iters 10 > $O/ab_$v.log 2> $O/ab_$v.err; tail -1 $O/ab_$v.err | cut -c1-200
done
timeout 400 scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current $O/ab_r4_vs_current.json --batch 64 --reps 5 --iters 10 > $O/ab_current.log 2> $O/ab_current.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5g/ab_r4_vs_*.json")):
    try:
        j = json.load(open(f)); d = j["classes_B_minus_A_ms"]
        print(f.split("vs_")[1][:-5], "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "fc2 B-A", d["gemm_fc2"], "fc1", d["gemm_fc1"], "| A fc2", j["A"]["classes_ms_mean"]["gemm_fc2"], "B fc2", j["B"]["classes_ms_mean"]["gemm_fc2"], "tok", j["outputs_bit_identical"]["token_idx"])
    except Exception as e:
        print(f, "parse failed", e)
PY
echo "total t=$(( $(date +%s) - t0 ))"
