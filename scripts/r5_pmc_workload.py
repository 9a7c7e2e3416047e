"""Workload for the round-5 rocprofv3 --pmc passes (separate passes, --kernel-trace only): the kernels of a 64-crop ViT block in the default
(split3) mode as the engine runs them — the 16x16x32 split3 GEMM on the qkv / proj shapes one workgroup per tile, fc1 (GELU, row-blocked
split3 output) on the mixed grid whose last round runs as half tiles (gemm_split16_tail_kernel), fc2 persistent over the row-blocked operand,
the bf16-pipe attention with split3 output — and the exact-fp32 fc1 GEMM of the opt-out mode.  `n` launches each (default 5).
    python scripts/r5_pmc_workload.py [gemm|attn|all] [n]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 64 * 192
which = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def mk(N, K):
    a = torch.randn(M, K, generator=g); a[:, ::97] *= 40.0
    return a.to(dev), (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev), torch.randn(N, generator=g).to(dev)


if which in ("all", "gemm"):
    a, w, b = mk(3840, 1280)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(n):
        ops.gemm_split3(sa, sw, b, epi="bias_qscale", qscale=80 ** -0.5, qcols=1280, variant="auto")
    a, w, b = mk(1280, 1280)
    r = torch.randn(M, 1280, generator=g).to(dev)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(n):
        ops.gemm_split3(sa, sw, b, r, epi="bias_resid", variant="auto")
    a, w, b = mk(5120, 1280)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(n):
        ops.gemm_split3(sa, sw, b, epi="bias_gelu", variant="tail", out_split=True, out_blocked=True)        # what the rule picks for this shape
    for _ in range(n):
        ops.gemm_split3(sa, sw, b, epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True)  # the plain grid (round 4) beside it
    for _ in range(n):
        ops.gemm(a, w, b, epi="bias_gelu")                     # the exact-fp32 mode's fc1
    a, w, b = mk(1280, 5120)
    sa, sw = ops.split3(a), ops.split3(w)
    sab = ops.split3_block(sa)
    for _ in range(n):
        ops.gemm_split3(sab, sw, b, r, epi="bias_resid", variant="persist", a_blocked_rows=M)
    torch.cuda.synchronize()
if which in ("all", "gemm", "attn"):
    qkv = torch.randn(64, 192, 3840, generator=g)
    qkv[:, :, :1280] *= 80 ** -0.5
    qkv = qkv.to(dev)
    for _ in range(n):
        ops.vit_attention_b16(qkv, out_split=True)
    torch.cuda.synchronize()
