"""Small-batch latency of the full path with and without hipGraph replay (captured through torch.cuda.CUDAGraph around
Engine.forward: thmr_forward never allocates or synchronises, so the whole path is capturable).
    python scripts/graph_latency.py [B ...]"""
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
Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]
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
    ref_v = outs["pred_vertices"].clone()

    def timed(fn, n=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def lat(fn, n=20):                      # sync after every call: what a single request sees
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    eager_tp, eager_lat = timed(lambda: eng.forward(img, outputs=outs)), lat(lambda: eng.forward(img, outputs=outs))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(img, outputs=outs)
    torch.cuda.current_stream().wait_stream(s)
    outs["pred_vertices"].zero_()
    with torch.cuda.graph(g, stream=s):
        eng.forward(img, outputs=outs)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(outs["pred_vertices"], ref_v)
    graph_tp, graph_lat = timed(g.replay), lat(g.replay)
    res[B] = {"eager_ms": round(eager_tp, 3), "eager_latency_ms": round(eager_lat, 3), "graph_ms": round(graph_tp, 3),
              "graph_latency_ms": round(graph_lat, 3), "bit_identical": same}
    print(B, res[B], flush=True)
print(json.dumps(res))
