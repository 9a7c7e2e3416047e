"""Stand-alone timing of the SMPL stage (prep -> blend GEMM -> skin + joints) through the thmr_smpl handle.
    python scripts/lbs_bench.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.smpl import SMPL
from tokenhmr_amd.smpl_assets import make_synthetic_smpl

dev = torch.device("cuda:0")
smpl = make_synthetic_smpl()
for B in [int(x) for x in sys.argv[1:]] or [1, 64, 512]:
    m = SMPL(smpl, max_batch=B, device=dev)
    g = torch.Generator().manual_seed(9)
    R = torch.linalg.qr(torch.randn(B * 24, 3, 3, generator=g))[0]
    R = (R * torch.linalg.det(R).sign()[:, None, None]).reshape(B, 24, 3, 3).to(dev)
    betas = torch.randn(B, 10, generator=g).to(dev)
    for _ in range(30):                      # a 1 ms window right after an idle gap runs at ramping clocks (outliers of 5-8x were seen)
        m(R[:, :1], R[:, 1:], betas, pose2rot=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        m(R[:, :1], R[:, 1:], betas, pose2rot=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = B * (83208 + 904) + 19.79e6
    print(f"LBS B={B}: {ms * 1e3:.1f} us/call  {by / (ms * 1e-3) / 1e9:.0f} GB/s algorithmic ({by / 1e6:.1f} MB)", flush=True)
    m.close()
