#!/bin/bash
# Round 3, session O: penalty for one-block-per-CU grids in launch_gemm's cost model (3 ... 24 crops)
set -u
O=gpurun_out/r3o; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
for p in 1.1 1.0 1.25 1.1 1.0; do
  echo "== THMR_ALONE_PENALTY=$p" >> $O/alone.log
  THMR_ALONE_PENALTY=$p timeout 300 python scripts/mid_split_sweep.py 3 4 5 6 7 8 12 16 17 20 24 32 2>/dev/null | grep '^{' >> $O/alone.log
done
python - <<'PY'
import json
rows=[]; cur=None
for l in open("gpurun_out/r3o/alone.log"):
    if l.startswith("=="): cur=l.strip()[3:]
    elif l.startswith("{"): rows.append((cur, json.loads(l)["results"]))
Bs=sorted(int(b) for b in rows[0][1])
print("setting".ljust(26)+"".join(f"B={b:<7}" for b in Bs))
for name, r in rows:
    print(name.ljust(26)+"".join(f"{r[str(b)]['ms']:<9.3f}" for b in Bs))
PY
echo "total t=$(( $(date +%s) - t0 ))"
