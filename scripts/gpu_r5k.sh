#!/bin/bash
# Round 5, session K: the 8-rank rehearsal on ONE GPU with the round's final build: `bench.py --gpus 8 --single-device
# --backend gloo` — 8 real depth-32 engines as 8 ranks, rank 0 loads, arena broadcast, receivers finalize, forwards in turns, packed
# all-gather, rank 0 recomputes the other ranks' crops — for a global batch of 512 and a ragged one of 509.  One GPU: plumbing, not scaling.
set -u
O=gpurun_out/r5k; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_512.json 2> $O/rank8_512.err; grep -E "Error|error" $O/rank8_512.err | head -5 | cut -c1-300
echo "t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --global-batch 509 --steps 2 --warmup 1 --no-cpu-baseline > $O/rank8_509.json 2> $O/rank8_509.err; grep -E "Error|error" $O/rank8_509.err | head -5 | cut -c1-300
python - <<'PY'
import json
out = {"what": "bench.py --gpus 8 --single-device --backend gloo: 8 real engines as 8 ranks on ONE GPU (plumbing rehearsal, not a scaling measurement)"}
for f, key in (("rank8_512", "global_batch_512"), ("rank8_509", "global_batch_509_ragged")):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5k/{f}.json") if l.startswith("{")][-1])
        out[key] = j
        m = j["multi_gpu"]
        print(key, "value", j["value"], "n_gpus", j["n_gpus"], "gathered_ok", j.get("gathered_records_ok"), "bit_identical", (m.get("cross_rank_check") or {}).get("bit_identical"),
              "crops per rank", [r["crops"] for r in m["per_rank"]], "build", j.get("build"))
    except Exception as e:
        print(key, "parse failed", e)
json.dump(out, open("gpurun_out/r5k/r5_8rank_single_device.json", "w"), indent=1)
PY
echo "total t=$(( $(date +%s) - t0 ))"
