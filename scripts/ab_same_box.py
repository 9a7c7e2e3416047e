"""Same-box, same-process, INTERLEAVED A/B of two builds (or two knob settings) of libtokenhmr_hip on the full path.

Why: boxes of the pool differ by ~5 % and a run-to-run clock spread of 1-2 % sits on top, which is more than most kernel
changes of rounds 4-5 are worth; a number from one box next to a number from another decides nothing.  Here both arms live in ONE process
on ONE GPU — two engines, each on its own shared object — and are timed alternately (A B A B ...), `reps` windows of `iters` back-to-back
calls each, so that clock, temperature and neighbours are the same for both.  Per arm: every window's ms per call, the median, the
per-class HIP-event profile (one profiled call after each window, averaged), and whether the two arms' outputs are bit-identical.

    python scripts/build_ab_lib.py <git-ref> <name>                       # here (no GPU): build_ab/<name>/libtokenhmr_hip.so from that commit
    python scripts/ab_same_box.py --a build_ab/r4/libtokenhmr_hip.so --b current [--batch 64] [--reps 5] [--iters 10]
                                  [--mode split3|f32] [--a-env K=V ...] [--b-env K=V ...] [--out gpurun_out/ab.json]

`current` = the in-tree shipped library, `exp` = the in-tree experiments build (its THMR_* knobs are read when the engine is created, so
--a-env / --b-env select variants of ONE build).  The last line printed is the JSON that is also written to --out.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", required=True)
    ap.add_argument("--b", required=True)
    ap.add_argument("--a-env", nargs="*", default=[])
    ap.add_argument("--b-env", nargs="*", default=[])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mode", choices=["split3", "f32"], default="split3")
    ap.add_argument("--vit-depth", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    import torch
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine

    dev = torch.device("cuda:0")
    cfg = HMRConfig(vit_depth=a.vit_depth)
    sd, tok, smpl = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
    img = torch.randn(a.batch, 3, 256, 256, generator=torch.Generator().manual_seed(4000)).to(dev)

    def make(spec, env):
        which = {"current": None, "exp": True}.get(spec, spec)
        if isinstance(which, str) and not os.path.exists(which):
            sys.exit(f"ab_same_box: {which} does not exist (python scripts/build_ab_lib.py <git-ref> <name>)")
        old = {}
        for kv in env:
            k, v = kv.split("=", 1)
            old[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            e = Engine(cfg, max_batch=a.batch, device=dev, experiments=which)
            e.load_state(sd, tok)
            e.load_smpl(smpl)
            e.finalize()
            e.set_vit_gemm(a.mode)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return e

    arms = {"A": {"spec": a.a, "env": a.a_env}, "B": {"spec": a.b, "env": a.b_env}}
    for name, arm in arms.items():
        arm["eng"] = make(arm["spec"], arm["env"])
        arm["build"] = arm["eng"].lib.thmr_build_info().decode()
        arm["outs"] = arm["eng"]._alloc_outputs(a.batch, taps=False, want_probs=True)
        arm["ms"], arm["prof"] = [], []
        for _ in range(3):
            arm["eng"].forward(img, outputs=arm["outs"])
    torch.cuda.synchronize()
    for rep in range(a.reps):
        for name in (("A", "B") if rep % 2 == 0 else ("B", "A")):          # alternate who goes first, too
            arm = arms[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            arm["eng"].forward(img, outputs=arm["outs"])                    # one untimed call after the switch (weights of the other arm left L2 / MALL)
            e0.record()
            for _ in range(a.iters):
                arm["eng"].forward(img, outputs=arm["outs"])
            e1.record()
            torch.cuda.synchronize()
            arm["ms"].append(e0.elapsed_time(e1) / a.iters)
            arm["eng"].prof_enable(True)
            arm["eng"].forward(img, outputs=arm["outs"])
            torch.cuda.synchronize()
            arm["prof"].append({k: v["ms"] for k, v in arm["eng"].prof_collect().items() if v["launches"]})
            arm["eng"].prof_enable(False)
    snaps = {}
    for name, arm in arms.items():
        o = arm["eng"].forward(img, outputs=arm["outs"])
        torch.cuda.synchronize()
        arm["eng"].status()
        snaps[name] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
    res = {"what": "same-box interleaved A/B (scripts/ab_same_box.py)", "batch": a.batch, "mode": a.mode, "vit_depth": a.vit_depth,
           "reps": a.reps, "iters_per_window": a.iters, "gpu": torch.cuda.get_device_name(0)}
    for name, arm in arms.items():
        med = statistics.median(arm["ms"])
        classes = sorted({k for p in arm["prof"] for k in p})
        res[name] = {"lib": arm["spec"], "env": arm["env"], "build": arm["build"],
                     "ms_per_call_windows": [round(x, 3) for x in arm["ms"]], "ms_per_call_median": round(med, 3),
                     "crops_per_s_median": round(a.batch / med * 1e3, 1),
                     "classes_ms_mean": {k: round(sum(p.get(k, 0.0) for p in arm["prof"]) / len(arm["prof"]), 3) for k in classes}}
    ma, mb = res["A"]["ms_per_call_median"], res["B"]["ms_per_call_median"]
    res["B_over_A_time"] = round(mb / ma, 4)
    res["classes_B_minus_A_ms"] = {k: round(res["B"]["classes_ms_mean"].get(k, 0) - res["A"]["classes_ms_mean"].get(k, 0), 3)
                                   for k in res["A"]["classes_ms_mean"]}
    res["outputs_bit_identical"] = {k: bool(torch.equal(snaps["A"][k], snaps["B"][k])) for k in snaps["A"]}
    res["max_abs_diff"] = {k: float((snaps["A"][k].float() - snaps["B"][k].float()).abs().max()) for k in snaps["A"]}
    line = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line, flush=True)


if __name__ == "__main__":
    main()
