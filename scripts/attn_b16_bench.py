"""The attention operator stand-alone: fp32-MFMA kernels (thmr_op_vit_attention / _split3) beside the bf16 x 3-piece kernel
(thmr_op_vit_attention_b16, csrc/attention_b16.hip), us per launch (best of 5 windows of `iters` launches), error of each against fp64.

    python scripts/attn_b16_bench.py [B=64] [iters=50]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(77)
qkv = torch.randn(B, 192, 3840, generator=g)
qkv[:, :, :1280] *= 80 ** -0.5
d = qkv.to(dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / iters * 1e3)
    return round(best, 2)


n = min(B, 4)
t64 = qkv[:n].reshape(n, 192, 3, 16, 80).permute(2, 0, 3, 1, 4).double()
ref64 = ((t64[0] @ t64[1].transpose(-2, -1)).softmax(-1) @ t64[2]).transpose(1, 2).reshape(n, 192, 1280)
res = {"B": B, "flops_per_launch": 4.0 * B * 16 * 192 * 192 * 80}
for name, fn in (("f32_mfma", lambda: ops.vit_attention(d)), ("f32_mfma_split3_out", lambda: ops.vit_attention_split3(d)),
                 ("b16 qt=3", lambda: ops.vit_attention_b16(d, qt=3)), ("b16 qt=3 split3_out", lambda: ops.vit_attention_b16(d, out_split=True, qt=3)),
                 ("b16 qt=1", lambda: ops.vit_attention_b16(d, qt=1)), ("b16 qt=1 split3_out", lambda: ops.vit_attention_b16(d, out_split=True, qt=1))):
    us = timed(fn)
    e = {"us": us, "f32_equiv_tflops": round(res["flops_per_launch"] / us / 1e6, 1)}
    if "split3_out" not in name:
        e["max_err_vs_fp64"] = float((fn()[:n].cpu().double() - ref64).abs().max())
    res[name] = e
print(json.dumps(res))
