"""Why does the split3 fc1 GEMM take 690-697 us per launch in the pipeline and 644-661 us stand-alone (same kernel, same shapes)?
One difference: the pipeline walks 32 layers = 3.8 GB of split3 weights, so every GEMM streams its weights from HBM, while a stand-alone
loop re-reads one weight set from the memory-side cache.  This probe runs the four ViT GEMMs of a 64-crop batch in layer order (qkv ->
proj -> fc1 with row-blocked split3 output -> persistent fc2 on it) for S = 1, 2, 8 and 32 distinct weight sets and reports the mean
duration per GEMM class (HIP events around every launch, 3 passes of 32 layers after one warm-up pass).

    python scripts/cold_weights_probe.py
"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 64 * 192
x = torch.randn(M, 1280, generator=g).to(dev)
sa = ops.split3(x)
resid = torch.randn(M, 1280, generator=g).to(dev)
SHAPES = {"qkv": (3840, 1280), "proj": (1280, 1280), "fc1": (5120, 1280), "fc2": (1280, 5120)}


def weight_set(seed):
    gg = torch.Generator().manual_seed(100 + seed)
    out = {}
    for k, (N, K) in SHAPES.items():
        out[k] = (ops.split3((torch.randn(N, K, generator=gg) / math.sqrt(K)).to(dev)), torch.randn(N, generator=gg).to(dev))
    return out


res = {}
for S in (1, 2, 8, 32):
    sets = [weight_set(i) for i in range(S)]
    torch.cuda.synchronize()
    ev = {k: [] for k in SHAPES}

    def layer(i, record):
        w = sets[i % S]

        def run(k, fn):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = fn(); e1.record()
                ev[k].append((e0, e1))
                return r
            return fn()
        run("qkv", lambda: ops.gemm_split3(sa, w["qkv"][0], w["qkv"][1], epi="bias_qscale", qscale=80 ** -0.5, qcols=1280, variant="128x256/w8"))
        run("proj", lambda: ops.gemm_split3(sa, w["proj"][0], w["proj"][1], resid, epi="bias_resid", variant="128x256/w8"))
        h = run("fc1", lambda: ops.gemm_split3(sa, w["fc1"][0], w["fc1"][1], epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True))
        run("fc2", lambda: ops.gemm_split3(h, w["fc2"][0], w["fc2"][1], resid, epi="bias_resid", variant="persist", a_blocked_rows=M))

    for i in range(32):
        layer(i, False)
    for _ in range(3):
        for i in range(32):
            layer(i, True)
    torch.cuda.synchronize()
    res[f"{S} weight set(s) = {S * 0.118:.2f} GB"] = {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e3, 1) for k, v in ev.items()}
    print(S, json.dumps(res[f"{S} weight set(s) = {S * 0.118:.2f} GB"]), flush=True)
    del sets
    torch.cuda.empty_cache()
print(json.dumps(res))
