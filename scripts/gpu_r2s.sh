#!/bin/bash
# Round-2 run S: kernel stats of the head alone at B = 1 and B = 8 (where its fixed latency is 21 % / 6 % of a call)
mkdir -p gpurun_out/r2s; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for B in 1 8; do
  rm -rf gpurun_out/r2s/trace$B
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r2s/trace$B" -o t -- python "$R/scripts/head_bench.py" $B) > gpurun_out/r2s/trace$B.log 2>&1
  f=$(find gpurun_out/r2s/trace$B -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2s/head_b${B}_kernel_stats.csv
  echo "== B=$B"; cut -d, -f1-4 gpurun_out/r2s/head_b${B}_kernel_stats.csv | cut -c1-130 | head -24
  grep "B=" gpurun_out/r2s/trace$B.log | tail -1
  rm -rf gpurun_out/r2s/trace$B
done
