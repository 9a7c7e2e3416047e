"""fc1 (split3, GELU, row-blocked split3 output) takes 644 us per launch in a loop of its own and 688-697 us in layer order
(scripts/cold_weights_probe.py: cold weights explain 1 % of that).  Which neighbour makes the difference?  Loops of two kernels each,
events around the fc1 launches only.

    python scripts/fc1_neighbour_probe.py
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
sa = ops.split3(torch.randn(M, 1280, generator=g).to(dev))
resid = torch.randn(M, 1280, generator=g).to(dev)


def wset(N, K, seed):
    gg = torch.Generator().manual_seed(seed)
    return ops.split3((torch.randn(N, K, generator=gg) / math.sqrt(K)).to(dev)), torch.randn(N, generator=gg).to(dev)


wq, bq = wset(3840, 1280, 1)
wp, bp = wset(1280, 1280, 2)
w1, b1 = wset(5120, 1280, 3)
w2, b2 = wset(1280, 5120, 4)
big = torch.empty(160 * 1024 * 1024, device=dev)          # 640 MB: more than the memory-side cache
state = {}


def fc1():
    state["h"] = ops.gemm_split3(sa, w1, b1, epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True)


def fc1_f32out():
    ops.gemm_split3(sa, w1, b1, epi="bias_gelu", variant="128x256/w8")


NEIGH = {
    "none (fc1, fc1, ...)": None,
    "fc2 on fc1's output": lambda: ops.gemm_split3(state["h"], w2, b2, resid, epi="bias_resid", variant="persist", a_blocked_rows=M),
    "proj": lambda: ops.gemm_split3(sa, wp, bp, resid, epi="bias_resid", variant="128x256/w8"),
    "qkv": lambda: ops.gemm_split3(sa, wq, bq, epi="bias_qscale", qscale=80 ** -0.5, qcols=1280, variant="128x256/w8"),
    "640 MB fill": lambda: big.fill_(1.0),
    "640 MB read (sum)": lambda: big.sum(),
}
res = {}
for name, nb in NEIGH.items():
    for which, f in (("fc1", fc1), ("fc1 with fp32 output", fc1_f32out)):
        if which != "fc1" and name not in ("none (fc1, fc1, ...)", "fc2 on fc1's output"):
            continue
        fc1()
        for _ in range(10):
            f()
            if nb:
                nb()
        ev = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record()
            ev.append((e0, e1))
            if nb:
                nb()
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        res[f"{which} | neighbour: {name}"] = round(d[len(d) // 2], 1)
        print(f"{which} | neighbour: {name}: {d[len(d) // 2]:.1f} us (min {d[0]:.1f})", flush=True)
print(json.dumps(res))
