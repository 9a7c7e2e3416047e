"""Mid-batch tile selection check: every big-tile variant of the GEMM on the four ViT shapes at M = 192*B, against what the
auto-selector (launch_gemm's cost model) picks.  python scripts/tile_sweep.py [rounds]  ->  one line per (B, shape)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tokenhmr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = {"qkv": (3840, 1280, "bias"), "proj": (1280, 1280, "bias_resid"), "fc1": (5120, 1280, "bias_gelu"), "fc2": (1280, 5120, "bias_resid")}
VARS = ["auto", "128x128", "128x160", "64x64", "128x96"]
g = torch.Generator().manual_seed(0)
for B in (7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 64):
    M = 192 * B
    a_all = torch.randn(M, 5120, generator=g).to(dev)
    for nm, (N, K, epi) in SHAPES.items():
        a = a_all[:, :K].contiguous()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if epi == "bias_resid" else None
        t = {v: [] for v in VARS}
        for _ in range(rounds):
            for v in VARS:
                ops.gemm(a, w, b, r, epi=epi, variant=v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, w, b, r, epi=epi, variant=v)
                e1.record()
                torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        med = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
        best = min((v for v in VARS if v != "auto"), key=lambda v: med[v])
        tf = 2.0 * M * N * K / (med["auto"] * 1e-6) / 1e12
        print(f"B {B:3d} {nm:5s} auto {med['auto']:8.1f} us ({tf:5.1f} TF)  " + "  ".join(f"{v} {med[v]:8.1f}" for v in VARS[1:]) +
              f"   best {best}  auto/best {med['auto'] / med[best]:.3f}", flush=True)
