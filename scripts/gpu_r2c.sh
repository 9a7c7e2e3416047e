#!/bin/bash
# round 2, run C: persistent decoder kernel (parity + head time A/B vs the round-1 launch chain), attention variants, B=1 latency
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_smpl_bounds.py tests/test_bench_cli.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -30) > gpurun_out/r2c_pytest.log
for v in 3 12; do THMR_ATTN_VARIANT=$v timeout 300 python scripts/attn_bench.py 20 2>&1 | sed "s/^/variant $v: /" >> gpurun_out/r2c_attn.log; done
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2c_bench_fused.json 2> gpurun_out/r2c_bench_fused.err
THMR_LEGACY_HEAD=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2c_bench_legacy.json 2> gpurun_out/r2c_bench_legacy.err
timeout 600 python scripts/graph_latency.py 1 2 4 > gpurun_out/r2c_latency.log 2>&1
THMR_LEGACY_HEAD=1 timeout 600 python scripts/graph_latency.py 1 2 4 > gpurun_out/r2c_latency_legacy.log 2>&1
tail -8 gpurun_out/r2c_pytest.log; cat gpurun_out/r2c_attn.log; tail -3 gpurun_out/r2c_latency.log; tail -3 gpurun_out/r2c_latency_legacy.log
python - <<'PY'
import json
for f in ("fused","legacy"):
    try:
        j=json.loads([l for l in open(f"gpurun_out/r2c_bench_{f}.json") if l.startswith("{")][0])
        print(f, j["value"], j["roofline"]["classes_ms_per_step"], j["parity"])
    except Exception as e:
        print(f, "ERR", e)
PY
