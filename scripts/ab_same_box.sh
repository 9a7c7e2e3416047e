#!/bin/bash
# Same-box interleaved A/B of two builds of the library: thin wrapper over scripts/ab_same_box.py.
#   scripts/ab_same_box.sh <lib-or-"current"-or-"exp" A> <B> <out.json> [extra args of ab_same_box.py ...]
# e.g. (after `python scripts/build_ab_lib.py 447e554 r4` in the build container):
#   scripts/ab_same_box.sh build_ab/r4/libtokenhmr_hip.so current gpurun_out/r5a/ab_r4_vs_r5.json --reps 5
set -eu
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
A="$1"; B="$2"; OUT="$3"; shift 3
cd "$R"
exec python scripts/ab_same_box.py --a "$A" --b "$B" --out "$OUT" "$@"
