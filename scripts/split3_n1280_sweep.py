"""proj / fc2 shapes (N = 1280) of the split3 mode at B crops: unsplit vs split-K 2, on the tile the rule picks (run with THMR_SPLIT3_TILE=0 / 2
to force 128x256 / 128x128).  us per launch incl. the partial-sum reduce + epilogue kernel for the split variants.
    python scripts/split3_n1280_sweep.py 16 20 24 28 32 40 48"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(it): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / it * 1e3
for B in [int(x) for x in sys.argv[1:]]:
    M = B * 192
    row = {"B": B, "tile": os.environ.get("THMR_SPLIT3_TILE", "rule")}
    for name, K in (("proj", 1280), ("fc2", 5120)):
        a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(1280, K, generator=g) / math.sqrt(K)).to(dev)
        b = torch.randn(1280, generator=g).to(dev); r = torch.randn(M, 1280, generator=g).to(dev)
        sa, sw = ops.split3(a), ops.split3(w)
        for v in ("auto", "auto/k2"):
            row[f"{name} {v}"] = round(timed(lambda: ops.gemm_split3(sa, sw, b, r, epi="bias_resid", variant=v)), 1)
    print(row, flush=True)
