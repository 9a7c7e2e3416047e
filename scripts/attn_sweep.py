"""Attention kernel time per launch over the batch size for the forced variant (THMR_ATTN_VARIANT = 1: three 64-query workgroups per
(crop, head); 3: one 192-query workgroup; 5: persistent) — the variants are bit-identical, so the choice per batch size is free.
    THMR_ATTN_VARIANT=1 python scripts/attn_sweep.py 8 10 12 16 21 24 32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
res = {}
for B in [int(x) for x in sys.argv[1:]] or [8, 10, 12, 16, 21, 24, 32, 48, 64]:
    qkv = torch.randn(B, 192, 3840, generator=g).to(dev)
    for _ in range(3):
        ops.vit_attention(qkv)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.vit_attention(qkv)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8 * 1e3)
    res[B] = round(sorted(ts)[len(ts) // 2], 1)
print("variant", os.environ.get("THMR_ATTN_VARIANT", "auto"), res, flush=True)
