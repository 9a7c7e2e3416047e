"""Mid-size batches as ONE call vs TWO half-size calls running concurrently on two streams (two engines sharing one weight arena):
the partly filled last rounds of one lane's GEMM grids can fill with the other lane's blocks.   python scripts/two_lane_bench.py [B ...]"""
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
Bs = [int(x) for x in sys.argv[1:]] or [16, 32, 64]
e1 = Engine(cfg, max_batch=max(Bs), device=dev)
e1.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
e1.load_smpl(make_synthetic_smpl(cfg, 0))
e1.finalize()
e2 = Engine(cfg, max_batch=max(Bs) // 2, device=dev, weight_arena=e1.weight_arena)
e2.finalize(assume_all_loaded=True)
e3 = Engine(cfg, max_batch=max(Bs) // 2, device=dev, weight_arena=e1.weight_arena)
e3.finalize(assume_all_loaded=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in Bs:
    img = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(B)).to(dev)
    o1 = e1._alloc_outputs(B)
    oa, ob = e2._alloc_outputs(B // 2), e3._alloc_outputs(B // 2)

    def one():
        e1.forward(img, outputs=o1)

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            e2.forward(img[: B // 2], outputs=oa)
        with torch.cuda.stream(s2):
            e3.forward(img[B // 2:], outputs=ob)
        cur.wait_stream(s1); cur.wait_stream(s2)

    res = {}
    for name, fn in (("one", one), ("two", two), ("one", one), ("two", two)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(B * n / (time.perf_counter() - t0))
    same = torch.equal(o1["pred_vertices"][: B // 2], oa["pred_vertices"]) and torch.equal(o1["pred_vertices"][B // 2:], ob["pred_vertices"])
    print(f"B={B}: one call {max(res['one']):.1f} crops/s | two concurrent half-batches {max(res['two']):.1f} crops/s | bit-identical {same}", flush=True)
e1.status(); e2.status(); e3.status()
