"""Full path at B crops with the ViT GEMMs in both modes of thmr_set_vit_gemm: "split3" (fp32 operands as three bf16 pieces on the bf16 matrix pipe: the engine default since round 5) and
"f32" (exact-fp32 MFMA, the opt-out).  Per mode: ms per call (best of 3 windows of `iters` calls),
crops/s, the per-class profile of one profiled call, and the difference of the outputs between the modes.

    python scripts/mode_bench.py [B=64] [iters=10]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
cfg = HMRConfig()
eng = Engine(cfg, max_batch=B, device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
eng.load_smpl(make_synthetic_smpl(cfg, 0))
eng.finalize()
g = torch.Generator().manual_seed(4000)
img = torch.randn(B, 3, 256, 256, generator=g).to(dev)
outs = eng._alloc_outputs(B, taps=False, want_probs=True)
keep = {}
for mode in ("f32", "split3", "f32"):
    eng.set_vit_gemm(mode)
    for _ in range(3):
        eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            eng.forward(img, outputs=outs)
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / iters)
    eng.prof_enable(True)
    eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    prof = {k: round(v["ms"], 3) for k, v in eng.prof_collect().items() if v["launches"]}
    eng.prof_enable(False)
    o = eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    snap = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
    line = {"mode": mode, "B": B, "ms_per_call": round(best, 3), "crops_per_s": round(B / best * 1e3, 1), "profile_ms": prof}
    if mode in keep:
        line["bit_identical_to_first_run_of_this_mode"] = all(torch.equal(snap[k], keep[mode][k]) for k in snap)
    keep.setdefault(mode, snap)
    print(json.dumps(line), flush=True)
a, b = keep["f32"], keep["split3"]
diff = {k: float((a[k].float() - b[k].float()).abs().max()) for k in a if a[k].dtype.is_floating_point}
tok = [k for k in a if not a[k].dtype.is_floating_point]
print(json.dumps({"max_abs_diff_split3_vs_f32": diff, "integer_outputs_equal": {k: bool(torch.equal(a[k], b[k])) for k in tok}}))
