"""Interleaved A/B micro-benchmark of the tiled fp32 GEMM variants on the four ViT-H shapes at B=64
(through the C ABI, torch events on the launch stream).  Usage: python scripts/gemm_bench.py [rounds]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops

dev = torch.device("cuda:0")
M = 64 * 192
SHAPES = {"qkv": (M, 3840, 1280, "bias_qscale"), "proj": (M, 1280, 1280, "bias_resid"),
          "fc1": (M, 5120, 1280, "bias_gelu"), "fc2": (M, 1280, 5120, "bias_resid")}
VARIANTS = ["128x128reg", "128x160reg", "128x128", "128x160"]
only = os.environ.get("GEMM_VARIANTS")
if only:
    VARIANTS = only.split(",")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("GEMM_EPI_NONE"):
    SHAPES = {k: (m, n, kk, "none") for k, (m, n, kk, e) in SHAPES.items()}
g = torch.Generator().manual_seed(0)
res = {}
for name, (m, n, k, epi) in SHAPES.items():
    a = torch.randn(m, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    b = torch.randn(n, generator=g).to(dev) if epi != "none" else None
    r = torch.randn(m, n, generator=g).to(dev) if epi == "bias_resid" else None
    kw = dict(qscale=0.1118, qcols=1280) if epi == "bias_qscale" else {}
    times = {v: [] for v in VARIANTS}
    for v in VARIANTS:
        ops.gemm(a, w, b, r, epi=epi, variant=v, **kw)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, w, b, r, epi=epi, variant=v, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * m * n * k
    res[name] = {v: round(fl / (sorted(t)[len(t) // 2] * 1e-3) / 1e12, 1) for v, t in times.items()}
    print(name, res[name], flush=True)
print(json.dumps(res))
