"""fp32 GEMM on the bf16 matrix pipe (csrc/gemm_split.hip: three bf16 pieces per operand, six products, fp32 accumulate) beside the
exact-fp32 MFMA kernel, on the four ViT-H GEMM shapes of the hot path (vit.py:82-87,104-126) at B crops:
per shape the time of both kernels with the epilogue the engine uses, the time of converting the A operand, the error of both against
an fp64 product of the same operands (max over the output of |c - c64| / (|a| . |w|)), and fp32-equivalent TFLOP/s (2 M N K / t).

    python scripts/split3_bench.py [--crops 64] [--iters 20]
"""
import argparse
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from tokenhmr_amd import ops  # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-error", action="store_true")
    ap.add_argument("--persist", action="store_true", help="only the default tile and the persistent kernel")
    ap.add_argument("--schedule", action="store_true", help="schedule experiments and timing-only ablations instead of the variant table")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M = args.crops * 192
    shapes = [("qkv", 3840, 1280, "bias_qscale"), ("proj", 1280, 1280, "bias_resid"), ("fc1", 5120, 1280, "bias_gelu"),
              ("fc2", 1280, 5120, "bias_resid")]
    g = torch.Generator(device="cpu").manual_seed(0)
    rows = []
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g)
        a[:, ::97] *= 40.0                                        # a few outlier channels, as the ViT residual stream has
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        da, dw, db, dr = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
        kw = dict(qscale=80 ** -0.5, qcols=1280) if epi == "bias_qscale" else {}
        res = dr if epi == "bias_resid" else None
        sa, sw = ops.split3(da), ops.split3(dw)
        flop = 2.0 * M * N * K
        row = {"gemm": name, "M": M, "N": N, "K": K, "epi": epi}
        t32 = timed(lambda: ops.gemm(da, dw, db, res, epi=epi, **kw), args.iters)
        row["f32_mfma"] = {"us": round(t32 * 1e6, 1), "tflops": round(flop / t32 * 1e-12, 1)}
        row["convert_A_us"] = round(timed(lambda: ops.split3(da), args.iters) * 1e6, 1)
        if args.schedule:      # schedule experiments / timing-only ablations of the default tile, without epilogue
            row = {"gemm": name, "M": M, "N": N, "K": K, "epi": "none"}
            for v in ("128x256/w8", "exp/reads-every-2nd", "abl/no-copies", "abl/no-barrier", "abl/no-reads", "abl/none"):
                t = timed(lambda: ops.gemm_split3(sa, sw, variant=v), args.iters)
                row[v] = {"us": round(t * 1e6, 1), "f32_equiv_tflops": round(flop / t * 1e-12, 1)}
            print(json.dumps(row), flush=True)
            continue
        persist_ok = M % 128 == 0 and (M // 128) * (N // 256) >= 256
        # product kernels (16x16x32 MFMAs) and, for the A/B, the 32x32x16 kernels they replaced (experiments build)
        variants = ("128x256/w8", "persist", "old/128x256/w8", "old/persist") if args.persist else ("128x256/w8", "128x256/w4", "128x128/w4", "256x256/w4", "persist", "old/128x256/w8", "old/persist")
        for v in variants:
            if v.endswith("persist") and not persist_ok:
                continue
            t = timed(lambda: ops.gemm_split3(sa, sw, db, res, epi=epi, variant=v, **kw), args.iters)
            row["split3 " + v] = {"us": round(t * 1e6, 1), "f32_equiv_tflops": round(flop / t * 1e-12, 1), "vs_f32_mfma": round(t32 / t, 3)}
        if name == "fc1":        # what the engine runs: the GELU result written as fc2's split3 operand
            for v in ("128x256/w8", "persist/swap", "old/128x256/w8", "old/persist/lds", "old/persist/swap"):
                if "persist" in v and not persist_ok:
                    continue
                t = timed(lambda: ops.gemm_split3(sa, sw, db, epi=epi, variant=v, out_split=True), args.iters)
                row["split3-out " + v] = {"us": round(t * 1e6, 1), "f32_equiv_tflops": round(flop / t * 1e-12, 1), "vs_f32_mfma": round(t32 / t, 3)}
            for v in ("128x256/w8", "persist/swap"):      # ... in the row-blocked form (round 4: whole-line stores from the swapped-role epilogue)
                if v != "128x256/w8" and not persist_ok:
                    continue
                t = timed(lambda: ops.gemm_split3(sa, sw, db, epi=epi, variant=v, out_split=True, out_blocked=True), args.iters)
                row["split3-out row-blocked " + v] = {"us": round(t * 1e6, 1), "f32_equiv_tflops": round(flop / t * 1e-12, 1), "vs_f32_mfma": round(t32 / t, 3)}
        if name == "fc2":        # ... and read by fc2 as a row-blocked A
            sab = ops.split3_block(sa)
            for v in ("128x256/w8", "persist"):
                if v == "persist" and not persist_ok:
                    continue
                t = timed(lambda: ops.gemm_split3(sab, sw, db, res, epi=epi, variant=v, a_blocked_rows=M), args.iters)
                row["split3 row-blocked A " + v] = {"us": round(t * 1e6, 1), "f32_equiv_tflops": round(flop / t * 1e-12, 1), "vs_f32_mfma": round(t32 / t, 3)}
            del sab
        if not args.no_error:
            c64 = da.double() @ dw.double().t()
            bound = da.double().abs() @ dw.double().abs().t()
            e32 = ((ops.gemm(da, dw).double() - c64).abs() / bound)
            es = ((ops.gemm_split3(sa, sw).double() - c64).abs() / bound)
            row["err_over_abs_dot"] = {"f32_mfma_max": float(e32.max()), "f32_mfma_rms": float(e32.pow(2).mean().sqrt()),
                                       "split3_max": float(es.max()), "split3_rms": float(es.pow(2).mean().sqrt())}
            del c64, bound, e32, es
        rows.append(row)
        print(json.dumps(row), flush=True)
        del da, dw, dr, sa, sw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
