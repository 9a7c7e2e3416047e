"""What do the shader clock and the board power do while a GEMM kernel runs back to back?  (Is the split3 GEMM's ~1.3 PFLOP/s a power / clock
ceiling of the part, and does a kernel that removes idle time — the persistent one — merely trade it for a lower clock?)

Loops one operator for `--secs` seconds per variant while a thread samples the hwmon files of the GPU (average power, shader clock) — or
`rocm-smi` where they are absent — and prints per variant: launches/s, fp32-equivalent TFLOP/s, mean / max power, mean shader clock.
Also runs the per-tile split3 kernel on 128 and on 256 tiles (half the CUs idle vs none): if one round of 128 tiles takes as long as one
round of 256, the kernel is limited per CU; if it is clearly shorter, by something the CUs share (power, fabric).

    python scripts/power_probe.py [--secs 3]
"""
import argparse
import glob
import json
import math
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from tokenhmr_amd import ops  # noqa: E402


def find_hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        p = glob.glob(d + "/power1_average") + glob.glob(d + "/power1_input")
        f = glob.glob(d + "/freq1_input")
        if p:
            return p[0], (f[0] if f else None)
    return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.pfile, self.ffile = find_hwmon()
        self.stop = False
        self.power, self.clk = [], []

    def run(self):
        while not self.stop:
            try:
                if self.pfile:
                    self.power.append(int(open(self.pfile).read()) * 1e-6)
                    if self.ffile:
                        self.clk.append(int(open(self.ffile).read()) * 1e-6)
                    time.sleep(0.02)
                else:
                    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
                    j = json.loads(out)
                    for card in j.values():
                        for k, v in card.items():
                            if "Power" in k and "W" in k:
                                try:
                                    self.power.append(float(v))
                                except ValueError:
                                    pass
                            if k.startswith("sclk"):
                                try:
                                    self.clk.append(float(str(v).strip("()Mhz ")))
                                except ValueError:
                                    pass
                        break
            except Exception:            # noqa: BLE001 — a probe: keep sampling
                time.sleep(0.1)


def run(name, fn, flop, secs):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    s.stop = True
    s.join(timeout=15)
    row = {"variant": name, "us_per_launch": round(dt / n * 1e6, 1), "f32_equiv_tflops": round(flop * n / dt * 1e-12, 1),
           "power_w_mean": round(sum(s.power) / max(len(s.power), 1), 1), "power_w_max": round(max(s.power, default=0.0), 1),
           "sclk_mhz_mean": round(sum(s.clk) / max(len(s.clk), 1), 1), "samples": len(s.power),
           "source": "hwmon" if s.pfile else "rocm-smi"}
    print(json.dumps(row), flush=True)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", type=float, default=3.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)

    def operands(M, N, K):
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        return a.to(dev), w.to(dev)

    # 1. idle baseline
    s = Sampler(); s.start(); time.sleep(1.0); s.stop = True; s.join(timeout=15)
    print(json.dumps({"variant": "idle", "power_w_mean": round(sum(s.power) / max(len(s.power), 1), 1),
                      "sclk_mhz_mean": round(sum(s.clk) / max(len(s.clk), 1), 1), "source": "hwmon" if s.pfile else "rocm-smi"}), flush=True)
    # 2. the qkv shape of a 64-crop batch: exact-fp32 MFMA kernel, split3 per tile, split3 persistent
    M, N, K = 12288, 3840, 1280
    a, w = operands(M, N, K)
    sa, sw = ops.split3(a), ops.split3(w)
    flop = 2.0 * M * N * K
    run("f32_mfma qkv-shape", lambda: ops.gemm(a, w), flop, args.secs)
    run("split3 per-tile qkv-shape", lambda: ops.gemm_split3(sa, sw, variant="128x256/w8"), flop, args.secs)
    run("split3 persistent qkv-shape", lambda: ops.gemm_split3(sa, sw, variant="persist"), flop, args.secs)
    # 3. one round of the per-tile kernel on half / all of the CUs (K = 1280: 40 K tiles)
    for tiles_m in (8, 16):
        Mh = tiles_m * 128
        ah, wh = operands(Mh, 4096, 1280)
        sah, swh = ops.split3(ah), ops.split3(wh)
        run(f"split3 per-tile, ONE round of {tiles_m * 16} tiles", lambda: ops.gemm_split3(sah, swh, variant="128x256/w8"), 2.0 * Mh * 4096 * 1280, args.secs)
    # 4. more K per tile (fewer epilogues per flop): 256 tiles, K = 5120
    a4, w4 = operands(2048, 4096, 5120)
    s4a, s4w = ops.split3(a4), ops.split3(w4)
    run("split3 per-tile, ONE round of 256 tiles, K = 5120", lambda: ops.gemm_split3(s4a, s4w, variant="128x256/w8"), 2.0 * 2048 * 4096 * 5120, args.secs)
    run("split3 persistent, 256 tiles, K = 5120", lambda: ops.gemm_split3(s4a, s4w, variant="persist"), 2.0 * 2048 * 4096 * 5120, args.secs)


if __name__ == "__main__":
    main()
