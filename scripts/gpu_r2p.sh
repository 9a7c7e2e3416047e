#!/bin/bash
# Round-2 run P: persistent attention with the odd-slot workgroup starting late — micro sweep + bench at three delays
mkdir -p gpurun_out/r2p
timeout 300 build_ab/attn_timeline > gpurun_out/r2p/attn_timeline.log 2>&1; echo "rc=$?" >> gpurun_out/r2p/attn_timeline.log
grep "persistent\|block order" gpurun_out/r2p/attn_timeline.log
for d in 0 8 12 16; do
  THMR_ATTN_DEPHASE_US=$d timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2p/bench_dephase$d.json 2> gpurun_out/r2p/bench_dephase$d.err
  python - <<EOF
import json
d=json.loads(open("gpurun_out/r2p/bench_dephase$d.json").read().strip().split("\n")[-1])
print("dephase $d us:", d["value"], "crops/s", d["roofline"]["attention"], d["parity"]["mismatches"])
EOF
done
