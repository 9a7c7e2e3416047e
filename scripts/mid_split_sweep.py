"""Full-path time per call over mid-size batches for the split factor of proj / fc2 forced by THMR_MID_SPLIT (0 = unsplit, 2, 4;
unset = the engine's rule).  One process per setting (the knob is read at thmr_create).   python scripts/mid_split_sweep.py [B ...]"""
import json
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
cfg = HMRConfig()
Bs = [int(x) for x in sys.argv[1:]] or [7, 8, 10, 11, 12, 14, 16, 20, 23, 24, 32]
eng = Engine(cfg, max_batch=max(Bs), device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
eng.load_smpl(make_synthetic_smpl(cfg, 0))
eng.finalize()
res = {}
for B in Bs:
    img = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(B)).to(dev)
    outs = eng._alloc_outputs(B, taps=False, want_probs=True)
    for _ in range(3):
        eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    res[B] = {"ms": round(ms, 3), "crops_per_s": round(B / ms * 1e3, 1)}
eng.status()
print(json.dumps({"THMR_MID_SPLIT": os.environ.get("THMR_MID_SPLIT", "rule"), "results": res}))
