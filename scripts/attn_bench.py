"""Micro-benchmark of the ViT attention kernel at B=64 (1024 (crop,head) workgroups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
B = 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, 192, 3840, generator=g).to(dev)
qkv[:, :, :1280] *= 80 ** -0.5
for _ in range(3):
    ops.vit_attention(qkv)
torch.cuda.synchronize()
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.vit_attention(qkv)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4)
t = sorted(ts)[len(ts) // 2]
fl = 4.0 * B * 16 * 192 * 192 * 80
print(f"attention B={B}: {t*1e3:.1f} us/launch  {fl/(t*1e-3)/1e12:.1f} TFLOP/s")
