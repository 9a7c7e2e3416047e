#!/bin/bash
# Round 3, session H: verification of the final split-K rule (7 ... 16 crops, 64x128 tile at 7-8) + same-box comparison against unsplit
set -u
O=gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)|assert" $O/pytest_gpu.log | head -10; echo "t=$(( $(date +%s) - t0 ))"
for ms in "" "00"; do
  echo "== THMR_MID_SPLIT=$ms" >> $O/mid.log
  if [ -z "$ms" ]; then timeout 300 python scripts/mid_split_sweep.py 6 7 8 9 10 11 12 13 14 15 16 17 2>/dev/null | grep '^{' >> $O/mid.log
  else THMR_MID_SPLIT=$ms timeout 300 python scripts/mid_split_sweep.py 6 7 8 9 10 11 12 13 14 15 16 17 2>/dev/null | grep '^{' >> $O/mid.log; fi
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r3h/mid.log") if l.startswith("{")]
Bs=sorted(int(b) for b in rows[0]["results"])
print("B      rule(ms)   unsplit(ms)   rule crops/s   gain")
for b in Bs:
    a, u = rows[0]["results"][str(b)], rows[1]["results"][str(b)]
    print(f"{b:<5} {a['ms']:9.3f} {u['ms']:12.3f} {a['crops_per_s']:12.1f} {100*(u['ms']/a['ms']-1):8.1f} %")
PY
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-200 $O/bench_full.json
timeout 200 python scripts/lbs_bench.py 1 64 512 2>/dev/null | grep LBS | tee $O/lbs_bench.log
echo "total t=$(( $(date +%s) - t0 ))"
