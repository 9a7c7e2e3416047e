"""The tiny-M kernel (32x32 tiles, K split over the 8 waves of a workgroup, no LDS staging) against the ring kernel on the ViT's
GEMM shapes at M = 192 * B, B = 1 ... 6.   python scripts/tiny_vs_ring.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
SH = {"qkv": (3840, 1280, "bias_qscale", "ring4"), "proj": (1280, 1280, "bias_resid", "ring4/k4"), "fc1": (5120, 1280, "bias_gelu", "ring4"),
      "fc2": (1280, 5120, "bias_resid", "ring4/k4")}
g = torch.Generator().manual_seed(0)
for B in (1, 2, 3, 4, 6):
    M = 192 * B
    for name, (n, k, epi, ringv) in SH.items():
        a = torch.randn(M, k, generator=g).to(dev)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        r = torch.randn(M, n, generator=g).to(dev) if epi == "bias_resid" else None
        kw = dict(qscale=0.1118, qcols=1280) if epi == "bias_qscale" else {}
        VARS = [ringv, "tiny"]
        outs = {v: ops.gemm(a, w, b, r, epi=epi, variant=v, **kw) for v in VARS}
        err = (outs["tiny"] - outs[ringv]).abs().max().item()
        ts = {v: [] for v in VARS}
        for _ in range(7):
            for v in VARS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, w, b, r, epi=epi, variant=v, **kw)
                e1.record(); torch.cuda.synchronize()
                ts[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        res = {v: round(sorted(t)[len(t) // 2], 1) for v, t in ts.items()}
        fl = 2.0 * M * n * k
        print(f"M {M:4d} {name:5s} floor {fl/157.3e12*1e6:6.1f} us | {ringv:9s} {res[ringv]:7.1f}  tiny {res['tiny']:7.1f} us | max |tiny - ring| {err:.2e}", flush=True)
