"""Workload for rocprofv3 --pmc passes on the split3 GEMM: the fc1 and fc2 shapes of the ViT at 64 crops, 6 launches each (variant from argv)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
v = sys.argv[1] if len(sys.argv) > 1 else "128x256/w8"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 64 * 192
for N, K, epi in ((5120, 1280, "bias_gelu"), (1280, 5120, "bias_resid")):
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev); r = torch.randn(M, N, generator=g).to(dev)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(6):
        ops.gemm_split3(sa, sw, b, r if epi == "bias_resid" else None, epi=epi, variant=v)
    torch.cuda.synchronize()
