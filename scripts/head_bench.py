"""Head-only timing: thmr_head_forward (decoder KV GEMM + persistent decoder + mixer stack + logits/softmax + VQ decode +
assemble + LBS) on random ViT features, per batch size; prints the engine's HIP-event classes.  vit_depth = 1 engine (the
head does not depend on the backbone depth), so it starts in seconds.   python scripts/head_bench.py [B ...]
THMR_LEGACY_HEAD=1 selects the round-1 launch chain for A/B."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine

dev = torch.device("cuda:0")
cfg = HMRConfig(vit_depth=1)
Bs = [int(x) for x in sys.argv[1:]] or [1, 8, 64]
eng = Engine(cfg, max_batch=max(Bs), device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
eng.load_smpl(make_synthetic_smpl(cfg, 0))
eng.finalize()
tag = "legacy" if os.environ.get("THMR_LEGACY_HEAD") == "1" else "fused"
for B in Bs:
    ctx = torch.randn(B, 192, 1280, generator=torch.Generator().manual_seed(B)).to(dev)
    for _ in range(3):
        eng.head_forward(ctx)
    eng.status()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.head_forward(ctx)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    eng.prof_enable(True)
    for _ in range(5):
        eng.head_forward(ctx)
    torch.cuda.synchronize()
    eng.prof_enable(False)
    pr = eng.prof_collect()
    if os.environ.get("THMR_DEC_TIMELINE") == "1" and tag == "fused":
        import ctypes as C
        st = (C.c_uint64 * 100)()
        eng.lib.thmr_debug_decoder_timeline(eng.h, st, 100, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        t = [(st[i] - st[0]) * 0.01 for i in range(90)]
        # stamps: [start, after init, after barrier] + per layer 7 x (after step, after barrier) + after final stage
        steps = [round(t[i] - t[i - 1], 2) for i in range(1, 2 + 1 + 6 * 14 + 1)]
        print(f"[timeline B={B}] us between stamps (step, barrier alternating after the first): {steps}", flush=True)
    print(f"[{tag}] B={B}: head_forward {wall:.3f} ms/call | " + " ".join(f"{k}={v['ms'] / 5:.3f}ms" for k, v in pr.items() if v["launches"]), flush=True)
eng.status()
