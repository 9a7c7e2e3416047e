"""Timing-only ablations of vit_attention_b16_kernel<3, true> (experiments build, THMR_ATTN_ABL: csrc/attention_b16.hip) — what bounds the
kernel.  Interleaved: every variant is timed in turn, `reps` windows of `iters` launches, median per variant.
  0 = the product kernel                      1 = q loaded for an item's first key block only (no q re-reads; vector work kept)
  2 = no K / V loads after the first block    3 = 1 + 2 (no global loads in steady state)
  4 = q re-split for the first block only (its vector work gone, loads kept)          5 = 1 + 4       7 = all three
    THMR_LIB=exp python scripts/attn_ablation.py [B=64] [iters=40] [reps=7]
"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("THMR_LIB", "exp")
import torch
from tokenhmr_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(77)
qkv = torch.randn(B, 192, 3840, generator=g)
qkv[:, :, :1280] *= 80 ** -0.5
d = qkv.to(dev)
variants = [0, 1, 2, 3, 4, 5, 7, "e1", "e2", "e3", "g256", "g128", "o0", "o1", "o2", "o3"]      # oN: compiled for one workgroup per CU (512 registers), THMR_ATTN_EARLY=N      # eN: THMR_ATTN_EARLY=N (real results: K / V requested N-1 k steps into the S phase)
ts = {v: [] for v in variants}


def run(v):
    os.environ["THMR_ATTN_ABL"] = "0" if isinstance(v, str) else str(v)
    os.environ["THMR_ATTN_EARLY"] = v[1:] if isinstance(v, str) and v[0] in "eo" else "0"
    os.environ["THMR_ATTN_OCC"] = "1" if isinstance(v, str) and v[0] == "o" else "0"
    os.environ["THMR_ATTN_GRID"] = v[1:] if isinstance(v, str) and v[0] == "g" else "0"      # gN: at most N workgroups (256 = one per CU)
    return ops.vit_attention_b16(d, out_split=True, qt=3)


for v in variants:
    for _ in range(5):
        run(v)
torch.cuda.synchronize()
for _ in range(reps):
    for v in variants:
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            run(v)
        t1.record()
        torch.cuda.synchronize()
        ts[v].append(t0.elapsed_time(t1) / iters * 1e3)
ref = run(0)
early_equal = {v: bool(torch.equal(run(v), ref)) for v in variants if isinstance(v, str)}
os.environ["THMR_ATTN_ABL"] = "0"
os.environ["THMR_ATTN_EARLY"] = "0"
os.environ["THMR_ATTN_GRID"] = "0"
os.environ["THMR_ATTN_OCC"] = "0"
res = {"B": B, "iters": iters, "reps": reps,
       "us_per_launch_median": {str(v): round(statistics.median(t), 2) for v, t in ts.items()},
       "us_per_launch_min": {str(v): round(min(t), 2) for v, t in ts.items()},
       "early_variants_bit_identical_to_product": early_equal,
       "what": "0 product; 1 no q re-reads; 2 no K/V loads; 3 = 1+2; 4 no q re-split; 5 = 1+4; 7 = 1+2+4 (timing only, garbage results for v > 0)"}
print(json.dumps(res))
