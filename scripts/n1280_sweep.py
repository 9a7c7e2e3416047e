"""The two N = 1280 GEMMs (proj K = 1280, fc2 K = 5120) at 7 ... 16 crops: every tile, unsplit and split-K 2 / 4 (the split variants
include the stand-alone reduce + epilogue kernel, ~8-12 us, which the engine fuses into its LayerNorm).
    python scripts/n1280_sweep.py [rounds]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tokenhmr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
TILES = ["64x64", "64x128", "128x64", "128x96", "128x128", "128x160"]
VARS = TILES + [f"{t}/k{k}" for k in (2, 4) for t in ("64x128", "128x64", "128x96", "128x128", "128x160")]
g = torch.Generator().manual_seed(0)
for B in (7, 8, 9, 10, 12, 16):
    M = 192 * B
    for nm, K in (("proj", 1280), ("fc2", 5120)):
        a = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(1280, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(1280, generator=g).to(dev)
        r = torch.randn(M, 1280, generator=g).to(dev)
        t = {v: [] for v in VARS}
        for _ in range(rounds):
            for v in VARS:
                ops.gemm(a, w, b, r, epi="bias_resid", variant=v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, w, b, r, epi="bias_resid", variant=v)
                e1.record()
                torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        med = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
        floor = 2.0 * M * 1280 * K / 157.3e12 * 1e6
        order = sorted(VARS, key=lambda v: med[v])
        print(f"B {B:3d} {nm:5s} floor {floor:6.1f} us | " + "  ".join(f"{v} {med[v]:6.1f}" for v in order[:6]) + " | unsplit: " +
              "  ".join(f"{v} {med[v]:6.1f}" for v in TILES), flush=True)
