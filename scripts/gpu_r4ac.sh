#!/bin/bash
# Round 4, session AC: the committed LBS skin kernel compiled for more waves per SIMD (-fno-slp-vectorize: 204 registers; launch bounds for 3 / 4
# waves per SIMD: 168 / 128 registers with 92 / 244 bytes of scratch) — hand-linked variants of the experiments library under build_ab/
set -u
O=gpurun_out/r4ac; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
echo "committed build:" | tee $O/lbs_occupancy_ab.log
timeout 200 python scripts/lbs_bench.py 64 512 2>/dev/null | tee -a $O/lbs_occupancy_ab.log
cp tokenhmr_amd/lib/libtokenhmr_hip_exp.so /tmp/exp_backup.so
for v in noslp occ3 occ4; do
  cp build_ab/libtokenhmr_hip_exp_lbs_$v.so tokenhmr_amd/lib/libtokenhmr_hip_exp.so
  echo "variant $v:" | tee -a $O/lbs_occupancy_ab.log
  THMR_LIB=exp timeout 200 python scripts/lbs_bench.py 64 512 2>/dev/null | tee -a $O/lbs_occupancy_ab.log
done
cp /tmp/exp_backup.so tokenhmr_amd/lib/libtokenhmr_hip_exp.so
