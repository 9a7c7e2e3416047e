#!/bin/bash
# Round 5, session N: cache policy of the bf16-pipe attention kernel's memory accesses.  Its PMC pass shows 4.5 TB/s of memory-side traffic at
# 36 % L2 hit: q is read once per key block (3x) and K / V stream through the same L2.  Arms (build_ab_lib.py WORKTREE ntN -DTHMR_ATTN_NT=N):
# 1 = K / V loads non-temporal (evict first, so that q's lines survive for the re-reads), 2 = the split3 output stores non-temporal, 3 = both.
set -u
O=gpurun_out/r5n; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for n in 1 2 3; do
  timeout 300 python scripts/ab_same_box.py --a current --b build_ab/nt$n/libtokenhmr_hip.so --out $O/ab_current_vs_nt$n.json --batch 64 --reps 5 --iters 10 > $O/ab_nt$n.log 2> $O/ab_nt$n.err
  python - $n <<'PY'
import json, sys
try:
    j = json.load(open(f"gpurun_out/r5n/ab_current_vs_nt{sys.argv[1]}.json"))
    d = j["classes_B_minus_A_ms"]
    print("nt" + sys.argv[1], "A", j["A"]["ms_per_call_median"], "B", j["B"]["ms_per_call_median"], "B/A", j["B_over_A_time"], "|", {k: d[k] for k in ("attention", "gemm_qkv", "gemm_proj", "layernorm")}, "| bit-identical", all(j["outputs_bit_identical"].values()))
except Exception as e:
    print(sys.argv[1], "parse failed", e)
PY
done
echo "total t=$(( $(date +%s) - t0 ))"
