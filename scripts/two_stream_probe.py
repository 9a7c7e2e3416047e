"""Do the ragged last rounds of the ViT GEMMs fill up when a 64-crop batch runs as TWO (or four) independent sub-batches on their own
streams?  qkv / proj leave 6.25 % of the CUs idle in their last round at 64 crops; kernels of another stream's sub-batch could take
those CUs.  Probe: one engine on 64 crops against 2 engines x 32 crops and 4 x 16 on separate streams (each engine its own scratch and
split3 copies; the per-device turnstile serialises their persistent kernels), interleaved windows, whole-path crops/s.

    python scripts/two_stream_probe.py [--reps 5] [--iters 10] [--out gpurun_out/two_stream.json]
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine

    dev = torch.device("cuda:0")
    cfg = HMRConfig()
    sd, tok, smpl = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
    img = torch.randn(64, 3, 256, 256, generator=torch.Generator().manual_seed(4000)).to(dev)

    def make(mb):
        e = Engine(cfg, max_batch=mb, device=dev)
        e.load_state(sd, tok)
        e.load_smpl(smpl)
        e.finalize()
        return e

    arms = {}
    for name, parts in (("1x64", 1), ("2x32", 2), ("4x16", 4)):
        n = 64 // parts
        engs = [make(n) for _ in range(parts)]
        arms[name] = {"engs": engs, "streams": [torch.cuda.Stream(device=dev) for _ in range(parts)], "n": n,
                      "outs": [e._alloc_outputs(n, taps=False, want_probs=True) for e in engs], "ms": []}

    def run(arm, iters):
        for it in range(iters):
            for k, (e, s) in enumerate(zip(arm["engs"], arm["streams"])):
                with torch.cuda.stream(s):
                    e.forward(img[k * arm["n"]:(k + 1) * arm["n"]], outputs=arm["outs"][k])

    for arm in arms.values():
        run(arm, 3)
    torch.cuda.synchronize()
    for rep in range(a.reps):
        for name in (list(arms) if rep % 2 == 0 else list(arms)[::-1]):
            arm = arms[name]
            run(arm, 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s in arm["streams"]:
                s.wait_event(e0)
            run(arm, a.iters)
            for s in arm["streams"]:
                torch.cuda.current_stream().wait_stream(s)
            e1.record()
            torch.cuda.synchronize()
            arm["ms"].append(e0.elapsed_time(e1) / a.iters)
    for arm in arms.values():
        for e in arm["engs"]:
            e.status()
    ref = arms["1x64"]["outs"][0]["pred_vertices"]
    res = {"what": "64 crops as 1 / 2 / 4 concurrent sub-batches on their own streams (scripts/two_stream_probe.py)", "gpu": torch.cuda.get_device_name(0)}
    for name, arm in arms.items():
        med = statistics.median(arm["ms"])
        got = torch.cat([o["pred_vertices"] for o in arm["outs"]])
        res[name] = {"ms_per_64_crops_windows": [round(x, 3) for x in arm["ms"]], "median": round(med, 3), "crops_per_s": round(64e3 / med, 1),
                     "max_abs_diff_vertices_vs_1x64": float((got - ref).abs().max())}
    line = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
