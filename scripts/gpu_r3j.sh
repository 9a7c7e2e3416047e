#!/bin/bash
# Round 3, session J: key-split attention with 16 / 32 / 48 queries per workgroup over 1 ... 6 crops against the 64-query kernel
set -u
O=gpurun_out/r3j; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log
for env in "THMR_ATTN_KEYSPLIT=0" "THMR_ATTN_KEYSPLIT_QT=1" "THMR_ATTN_KEYSPLIT_QT=2" "THMR_ATTN_KEYSPLIT_QT=3" "THMR_ATTN_KEYSPLIT=0" "THMR_ATTN_KEYSPLIT_QT=2" "THMR_ATTN_KEYSPLIT_QT=3"; do
  echo "== $env" >> $O/keysplit_qt.log
  env $env timeout 300 python scripts/mid_split_sweep.py 1 2 3 4 5 6 2>/dev/null | grep '^{' >> $O/keysplit_qt.log
done
python - <<'PY'
import json
rows=[]; cur=None
for l in open("gpurun_out/r3j/keysplit_qt.log"):
    if l.startswith("=="): cur=l.strip()[3:]
    elif l.startswith("{"): rows.append((cur, json.loads(l)["results"]))
print("setting".ljust(28)+"".join(f"B={b:<8}" for b in range(1,7)))
for name, r in rows:
    print(name.ljust(28)+"".join(f"{r[str(b)]['ms']:<10.3f}" for b in range(1,7)))
PY
echo "total t=$(( $(date +%s) - t0 ))"
