#!/bin/bash
# Round-2 run R: mid-batch diagnosis — per-class time of the full path at B = 8 / 16 / 32 and the tile sweep at those sizes
mkdir -p gpurun_out/r2r
SKIP_GEMM=1 timeout 300 python scripts/small_batch_prof.py 8 16 32 64 > gpurun_out/r2r/classes.log 2>&1; grep "^B " gpurun_out/r2r/classes.log
timeout 600 python - > gpurun_out/r2r/tile_sweep.log 2>&1 <<'EOF'
import sys, torch
sys.path.insert(0, ".")
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
SHAPES = {"qkv": (3840, 1280, "bias"), "proj": (1280, 1280, "bias_resid"), "fc1": (5120, 1280, "bias_gelu"), "fc2": (1280, 5120, "bias_resid")}
VARS = ["auto", "128x128", "128x160", "64x64", "128x96"]
g = torch.Generator().manual_seed(0)
for B in (8, 16, 32):
    M = 192 * B
    a_all = torch.randn(M, 5120, generator=g).to(dev)
    for nm, (N, K, epi) in SHAPES.items():
        a = a_all[:, :K].contiguous()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if epi == "bias_resid" else None
        t = {v: [] for v in VARS}
        for _ in range(5):
            for v in VARS:
                ops.gemm(a, w, b, r, epi=epi, variant=v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, w, b, r, epi=epi, variant=v)
                e1.record(); torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        med = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
        fl = 2.0 * M * N * K
        print(f"B {B:3d} {nm:5s} floor {fl/157.3e12*1e6:7.1f} us | " + "  ".join(f"{v} {med[v]:7.1f}" for v in VARS) + f" | auto = {fl/med['auto']/1e6:5.1f} TF", flush=True)
EOF
cat gpurun_out/r2r/tile_sweep.log | grep "^B "
