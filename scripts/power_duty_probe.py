"""Is the split3 GEMM's rate set by a power / thermal envelope?  The SAME kernel (the fc1 shape of a 64-crop batch, 16x16x32 split3 GEMM with
GELU + split3 row-blocked output, as the engine runs it) launched back to back and with idle gaps of 1x / 3x its own duration between launches
(torch.cuda._sleep: one spinning wave on the stream).  Per setting: the mean duration of the GEMM launches alone (HIP events around each launch,
after a 1.5 s soak at that duty cycle).  A kernel limited by its own schedule takes the same time whatever precedes it; a part that is
holding a power or temperature budget runs it faster after a pause.  The exact-fp32 fc1 GEMM beside it as the control.

    python scripts/power_duty_probe.py
"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, K = 64 * 192, 5120, 1280
a = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
b = torch.randn(N, generator=g).to(dev)
sa, sw = ops.split3(a), ops.split3(w)
kernels = {
    "split3 fc1 (bf16 pipe)": lambda: ops.gemm_split3(sa, sw, b, epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True),
    "exact-fp32 fc1": lambda: ops.gemm(a, w, b, epi="bias_gelu"),
}
# spin cycles per microsecond of torch.cuda._sleep: calibrated once
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
res = {"sleep_cycles_per_us": round(cyc_per_us, 1)}
for name, fn in kernels.items():
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        fn()
    t1.record(); torch.cuda.synchronize()
    base_us = t0.elapsed_time(t1) / 50 * 1e3
    out = {}
    for gap in (0.0, 1.0, 3.0, 0.0):
        sleep_cycles = int(gap * base_us * cyc_per_us)
        # soak at this duty cycle, then measure 60 launches with an event pair each
        t_end = time.perf_counter() + 1.5
        while time.perf_counter() < t_end:
            for _ in range(10):
                fn()
                if sleep_cycles:
                    torch.cuda._sleep(sleep_cycles)
            torch.cuda.synchronize()
        evs = []
        for _ in range(60):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(); fn(); s1.record()
            if sleep_cycles:
                torch.cuda._sleep(sleep_cycles)
            evs.append((s0, s1))
        torch.cuda.synchronize()
        d = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
        key = f"gap {gap:g}x" + (" (again)" if gap == 0.0 and "gap 0x" in out else "")
        out[key] = {"median_us": round(d[len(d) // 2], 1), "min_us": round(d[0], 1), "duty": round(1.0 / (1.0 + gap), 2)}
    res[name] = {"back_to_back_us": round(base_us, 1), **out}
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps(res))
