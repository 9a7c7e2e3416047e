"""thmr_forward replayed from a hipGraph against eager launches: the full path at 1 / 2 / 8 / 32 / 64 crops, default mode, release depth.
Round 1 found no difference at one crop (profiles/r1_graph_latency.log); this repeats it for the round-5 kernels and the reference's batch sizes.

    python scripts/graph_probe.py [--out gpurun_out/graph_probe.json]
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
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine

    dev = torch.device("cuda:0")
    cfg = HMRConfig()
    eng = Engine(cfg, max_batch=64, device=dev)
    eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
    eng.load_smpl(make_synthetic_smpl(cfg, 0))
    eng.finalize()
    img = torch.randn(64, 3, 256, 256, generator=torch.Generator().manual_seed(4000)).to(dev)
    res = {"what": "thmr_forward eager vs replayed from a hipGraph (scripts/graph_probe.py)", "mode": eng.vit_gemm(), "rows": []}
    side = torch.cuda.Stream()
    for B in (1, 2, 8, 32, 64):
        iters = 40 if B <= 8 else 12
        buf = img[:B].clone()
        outs = eng._alloc_outputs(B, taps=False, want_probs=True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eng.forward(buf, outputs=outs)
        torch.cuda.synchronize()
        ref = outs["pred_vertices"].clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng.forward(buf, outputs=outs)
        torch.cuda.synchronize()
        t = {"eager": [], "graph": []}
        for rep in range(a.reps):
            for arm in (("eager", "graph") if rep % 2 == 0 else ("graph", "eager")):
                with torch.cuda.stream(side):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    (eng.forward(buf, outputs=outs) if arm == "eager" else graph.replay())
                    e0.record()
                    for _ in range(iters):
                        if arm == "eager":
                            eng.forward(buf, outputs=outs)
                        else:
                            graph.replay()
                    e1.record()
                torch.cuda.synchronize()
                t[arm].append(e0.elapsed_time(e1) / iters)
        eng.status()
        me, mg = statistics.median(t["eager"]), statistics.median(t["graph"])
        res["rows"].append({"batch": B, "eager_ms": round(me, 4), "graph_ms": round(mg, 4), "graph_over_eager": round(mg / me, 4),
                            "eager_crops_per_s": round(B / me * 1e3, 1), "graph_crops_per_s": round(B / mg * 1e3, 1),
                            "bit_identical": bool(torch.equal(outs["pred_vertices"], ref))})
        del graph
    line = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
