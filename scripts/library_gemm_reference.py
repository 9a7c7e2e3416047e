"""Calibration only (nothing here is used by the product): what do the vendor GEMM libraries behind torch.mm reach on THIS part for the
ViT shapes of a 64-crop batch — fp32 (the arithmetic of the exact-fp32 mode) and ONE bf16 product (the split3 mode runs six per element
pair)?  Puts the hand-written kernels' rates (fc1: 146 TFLOP/s exact fp32; 1.39 PFLOP/s of bf16 MFMA work in the split3 mode) beside what
the part sustains for a library kernel of the same shape, and an 8192^3 bf16 GEMM beside the micro-benchmark's sustained rate.

    python scripts/library_gemm_reference.py
"""
import json
import torch

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
g = torch.Generator().manual_seed(0)
res = {}
SHAPES = {"qkv": (12288, 3840, 1280), "proj": (12288, 1280, 1280), "fc1": (12288, 5120, 1280), "fc2": (12288, 1280, 5120), "square 8192": (8192, 8192, 8192)}
for name, (M, N, K) in SHAPES.items():
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        x, y = a.to(dt), w.to(dt).t().contiguous().t()          # y: (N, K) row-major viewed as its transpose's transpose; mm(x, w^T)
        wt = w.to(dt).t()
        for _ in range(5):
            torch.mm(x, wt)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch.mm(x, wt)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        out["f32" if dt == torch.float32 else "bf16"] = {"us": round(best, 1), "tflops": round(2.0 * M * N * K / best / 1e6, 1)}
    res[name] = out
    print(name, json.dumps(out), flush=True)
print(json.dumps(res))
