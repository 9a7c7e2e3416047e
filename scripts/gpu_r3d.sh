#!/bin/bash
# Round 3, session D: crossovers of the big-tile split-K (proj and fc2 separately) over 8 ... 22 crops
set -u
O=gpurun_out/r3d; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
: > $O/mid_split_sweep.jsonl
for ms in 00 02 20 22 04; do
  THMR_MID_SPLIT=$ms timeout 300 python scripts/mid_split_sweep.py 8 9 10 11 12 13 14 15 16 17 18 19 20 22 2>/dev/null | grep '^{' >> $O/mid_split_sweep.jsonl
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r3d/mid_split_sweep.jsonl")]
Bs=sorted(int(b) for b in rows[0]["results"])
print("B     "+"  ".join(f"{r['THMR_MID_SPLIT']:>7}" for r in rows))
for b in Bs:
    print(f"{b:<5} "+"  ".join(f"{r['results'][str(b)]['ms']:7.3f}" for r in rows))
PY
echo "total t=$(( $(date +%s) - t0 ))"
