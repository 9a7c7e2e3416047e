import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops
dev = torch.device("cuda:0")
for B in (11, 16, 24, 32, 48, 64, 96, 128):
    qkv = torch.randn(B, 192, 3840, device=dev)
    for _ in range(3): ops.vit_attention(qkv)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.vit_attention(qkv)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    t = sorted(ts)[3]
    print(f"B {B:4d}  WGs {B*16:5d} ({B*16/256:.2f} per CU)  {t*1e3:7.1f} us   {4.0*B*16*192*192*80/(t*1e-3)/1e12:6.1f} TF", flush=True)
