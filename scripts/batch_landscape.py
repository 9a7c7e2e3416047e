"""Full path at EVERY batch size of a range, one engine, default mode: ms per call, crops/s and the per-class profile of one call — where the
tile / round quantisation of the GEMMs still shows (round 6: which sizes take which decomposition is in csrc/engine.hip, vit_forward).
    python scripts/batch_landscape.py [lo=1] [hi=40] [iters=10]      (THMR_LIB=exp + knobs for an A/B of a rule)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 40
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
cfg = HMRConfig()
eng = Engine(cfg, max_batch=hi, device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
eng.load_smpl(make_synthetic_smpl(cfg, 0))
eng.finalize()
img = torch.randn(hi, 3, 256, 256, generator=torch.Generator().manual_seed(4000)).to(dev)
for B in range(lo, hi + 1):
    x = img[:B].contiguous()
    outs = eng._alloc_outputs(B, taps=False, want_probs=True)
    for _ in range(3):
        eng.forward(x, outputs=outs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            eng.forward(x, outputs=outs)
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / iters)
    eng.prof_enable(True)
    eng.forward(x, outputs=outs)
    torch.cuda.synchronize()
    prof = {k.replace("gemm_", ""): round(v["ms"], 2) for k, v in eng.prof_collect().items() if v["launches"] and k.startswith(("gemm_", "layernorm", "attention"))}
    eng.prof_enable(False)
    print(json.dumps({"B": B, "ms": round(best, 3), "crops_per_s": round(B / best * 1e3, 1), "ms_per_crop": round(best / B, 3), "classes": prof}), flush=True)
eng.status()
