#!/bin/bash
# Round 4, session AD: the driver's launcher form of bench.py (torch.distributed.run, one rank over RCCL) with the split3 mode as the timed
# default: weight-arena broadcast -> finalize -> thmr_set_vit_gemm -> timed steps with the record all-gather
set -u
O=gpurun_out/r4ad; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dist_w1.json 2> $O/bench_dist_w1.err
echo "rc=$?"; tail -3 $O/bench_dist_w1.err | cut -c1-300
python - <<'PY'
import json
for l in open("gpurun_out/r4ad/bench_dist_w1.json"):
    l = l.strip()
    if l.startswith("{"):
        j = json.loads(l)
        print({k: j.get(k) for k in ("value", "vit_gemm", "n_gpus", "ms_per_step")}, "ranks", json.dumps(j.get("ranks"))[:600])
        print("parity", {k: j["parity"][k] for k in ("mismatches", "max_joint_err_m")} if j.get("parity") else None, "cross", j.get("cross_rank_check"))
PY
