"""Workload for the round-4 rocprofv3 --pmc passes (separate passes, --kernel-trace only): the kernels of a 64-crop ViT block in the split3
mode as the engine runs them now — the 16x16x32 split3 GEMM on the qkv / proj / fc1 (split3 output, row-blocked) shapes one workgroup per
tile and persistent on fc2, LayerNorm and attention with split3 output — the exact-fp32 fc1 GEMM, and ONE round of the per-tile split3 kernel
on 128 and on 256 tiles (half the CUs idle vs none: GRBM_GUI_ACTIVE / duration = the shader clock in both cases).  6 launches each."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 64 * 192


def mk(N, K):
    a = torch.randn(M, K, generator=g); a[:, ::97] *= 40.0
    return a.to(dev), (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev), torch.randn(N, generator=g).to(dev)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "gemm"):
    # as the engine runs them at 64 crops since the 16x16x32 kernel (csrc/gemm_split16.hip): qkv / proj / fc1 one workgroup per tile, fc1 with
    # its split3 output in the row-blocked form, fc2 persistent over the row-blocked operand
    a, w, b = mk(3840, 1280)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(6):
        ops.gemm_split3(sa, sw, b, epi="bias_qscale", qscale=80 ** -0.5, qcols=1280, variant="128x256/w8")
    a, w, b = mk(1280, 1280)
    r = torch.randn(M, 1280, generator=g).to(dev)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(6):
        ops.gemm_split3(sa, sw, b, r, epi="bias_resid", variant="128x256/w8")
    a, w, b = mk(5120, 1280)
    sa, sw = ops.split3(a), ops.split3(w)
    for _ in range(6):
        ops.gemm_split3(sa, sw, b, epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True)
    for _ in range(6):
        ops.gemm(a, w, b, epi="bias_gelu")                     # the exact-fp32 mode's fc1
    a, w, b = mk(1280, 5120)
    sa, sw = ops.split3(a), ops.split3(w)
    sab = ops.split3_block(sa)
    for _ in range(6):
        ops.gemm_split3(sab, sw, b, r, epi="bias_resid", variant="persist", a_blocked_rows=M)
    # the split3 mode's attention (csrc/attention_b16.hip) beside the fp32-MFMA kernel with split3 output it replaced
    qkv = torch.randn(64, 192, 3840, generator=g)
    qkv[:, :, :1280] *= 80 ** -0.5
    qkv = qkv.to(dev)
    for _ in range(6):
        ops.vit_attention_b16(qkv, out_split=True)
    for _ in range(6):
        ops.vit_attention_split3(qkv)
    torch.cuda.synchronize()
if which in ("all", "rows"):
    x = torch.randn(M, 1280, generator=g).to(dev)
    gam, bet = torch.randn(1280, generator=g).to(dev), torch.randn(1280, generator=g).to(dev)
    from tokenhmr_amd import _cabi
    import ctypes as C
    y = torch.empty(M, 160, 3, 8, dtype=torch.int16, device=dev)
    qkv = torch.randn(64, 192, 3840, generator=g).to(dev)
    for _ in range(6):
        ops.layernorm(x, gam, bet, 1e-6)
        ops.vit_attention(qkv)
        ops.vit_attention_split3(qkv)
    torch.cuda.synchronize()
if which in ("all", "half"):
    for tiles_m in (8, 16):
        ah = torch.randn(tiles_m * 128, 1280, generator=g).to(dev)
        wh = (torch.randn(4096, 1280, generator=g) / 36).to(dev)
        sah, swh = ops.split3(ah), ops.split3(wh)
        for _ in range(6):
            ops.gemm_split3(sah, swh, variant="128x256/w8")
    torch.cuda.synchronize()
