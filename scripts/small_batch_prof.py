"""Per-kernel-class time of the full path at small batch sizes (HIP-event profiler of the engine), and the tiled-GEMM
variants on the ViT shapes at M = 192*B.   python scripts/small_batch_prof.py [B ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W, ops
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine

dev = torch.device("cuda:0")
cfg = HMRConfig()
Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
eng = Engine(cfg, max_batch=max(Bs), device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
eng.load_smpl(make_synthetic_smpl(cfg, 0))
eng.finalize()
for B in ([] if os.environ.get("SKIP_PATH") else Bs):
    img = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(B)).to(dev)
    outs = eng._alloc_outputs(B, taps=False, want_probs=True)
    for _ in range(3):
        eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    for _ in range(5):
        eng.forward(img, outputs=outs)
    torch.cuda.synchronize()
    eng.prof_enable(False)
    p = eng.prof_collect()
    print("B", B, {k: round(v["ms"] / 5, 3) for k, v in p.items() if v["launches"]}, "sum", round(sum(v["ms"] for v in p.values()) / 5, 3), flush=True)

VARS = os.environ.get("GEMM_VARIANTS", "auto,64x64,128x128,128x160,ring4,ring8,ring4/k2,ring8/k2,ring4/k4,ring8/k4,ring4/k8,ring8/k8").split(",")
SH = {"qkv": (3840, 1280, "bias_qscale"), "proj": (1280, 1280, "bias_resid"), "fc1": (5120, 1280, "bias_gelu"), "fc2": (1280, 5120, "bias_resid")}
g = torch.Generator().manual_seed(0)
for B in ([] if os.environ.get("SKIP_GEMM") else Bs):
    M = 192 * B
    for name, (n, k, epi) in SH.items():
        a = torch.randn(M, k, generator=g).to(dev)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        r = torch.randn(M, n, generator=g).to(dev) if epi == "bias_resid" else None
        kw = dict(qscale=0.1118, qcols=1280) if epi == "bias_qscale" else {}
        for v in VARS:                                  # warm-up + clock ramp
            for _ in range(3):
                ops.gemm(a, w, b, r, epi=epi, variant=v, **kw)
        ts = {v: [] for v in VARS}
        for _ in range(7):                              # interleaved rounds: DVFS / order effects hit every variant equally
            for v in VARS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, w, b, r, epi=epi, variant=v, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts[v].append(e0.elapsed_time(e1) / 4)
        res = {v: round(sorted(t)[len(t) // 2] * 1e3, 1) for v, t in ts.items()}
        best = min(res, key=res.get)
        print("M", M, name, "us:", res, "best", best, round(2.0 * M * n * k / (res[best] * 1e-6) / 1e12, 1), "TF", flush=True)
