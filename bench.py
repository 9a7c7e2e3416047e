#!/usr/bin/env python
"""Benchmark of the TokenHMR inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 64 synthetic 256x256 crops per GPU (BASELINE.json configs[2]: full
TokenHMR = ViT-H + token decoder + VQ lookup/decode + SMPL LBS, batch 64 per GPU, fp32 in / fp32 out like the reference's
inference).  `--workload vit` times configs[1] (ViT-H encoder only).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line with the driver's fields plus `roofline`, `cpu_baseline` and `parity`.

Which arithmetic `value` is measured in (round 5: the ENGINE'S DEFAULT — the same mode load_tokenhmr() / thmr_create deliver to a caller
who passes no mode; bench.py does not touch the switch unless --vit-gemm is given).  The engine has two modes for the four ViT GEMM classes, both fp32 in / fp32 out with
fp32 accumulation: "f32" multiplies on the fp32 MFMA pipe (157 TFLOP/s peak: the path sits at 0.90 of it end to end and cannot move),
"split3" hands every fp32 operand to the bf16 MFMA pipe as three bf16 pieces (8 + 8 + 8 mantissa bits: the pieces sum to the fp32 value)
and keeps six of the nine piece products (what it drops is below 2^-24 of a product).  "split3" is the default (ABI 4); the line carries the exact-fp32 opt-out measured in the
same process as `exact_f32_mode` (`--vit-gemm f32` makes it the timed one), and an untimed `batch_sweep` at the reference's own batch sizes
(1, 8 = demo.py:70, 32 = README.md:316).  `parity` holds both against the reference's own modules and
against their float64 evaluation: on every fixture the split3 result is at least as close to float64 as the reference's fp32 result is.

N > 1: one process per GPU over RCCL.  Either the driver launches this file under `python -m torch.distributed.run
--nproc-per-node N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or — when WORLD_SIZE is
not set — `python bench.py --gpus N` re-executes ITSELF under torch.distributed.run with N ranks on 127.0.0.1
(`--self-launch` forces that path at N = 1 too).  Weak scaling: 64 crops per GPU; rank 0 "reads the checkpoint", ONE RCCL
broadcast replicates the packed weight arena, every step all-gathers the packed per-crop records asynchronously.
`--backend gloo --fake-engine` is a CPU dry run of exactly that orchestration (tests/test_bench_cli.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, dense (the split3 mode's pipe)
PEAK_HBM_GBS = 8000.0
GFLOP_PER_CROP = {"full": 252.10, "vit": 248.01}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="BASELINE configs[3]: shard ONE global batch of this many crops over the ranks (ragged shards allowed, "
                         "e.g. 509 over 8) instead of --batch crops per GPU; the line then says scaling = strong")
    ap.add_argument("--workload", choices=["full", "vit", "pipeline"], default="full",
                    help="full (default, the headline): thmr_forward on resident crops.  vit: the encoder only.  pipeline (one GPU): frames on the "
                         "host -> H2D -> crop kernels -> forward -> evaluator kernels on two streams, nothing else — the `pipeline` block of the "
                         "default line as its own run (e.g. under rocprofv3); its `value` is the pipeline's crops/s, not the headline")
    ap.add_argument("--vit-depth", type=int, default=32)
    ap.add_argument("--vit-gemm", choices=["f32", "split3"], default=None,
                    help="arithmetic of the four ViT GEMM classes in the timed region (both: fp32 in, fp32 accumulate, fp32 out).  Default: "
                         "the engine's own default is left alone = split3 (operands as three bf16 pieces on the bf16 matrix pipe, six "
                         "products; below 3 crops per GPU it runs the exact-fp32 kernels and the line says f32).  f32: thmr_set_vit_gemm(0), "
                         "the fp32 MFMA pipe.  The other mode is measured after the timed region and reported beside it "
                         "(`exact_f32_mode` / `split3_mode`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step packed all-gather at N>1")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (class split, LBS at B=512, parity)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--fake-engine", action="store_true", help="CPU dry run of the N-rank orchestration (with --backend gloo)")
    ap.add_argument("--self-launch", action="store_true", help="re-exec under torch.distributed.run even at --gpus 1")
    ap.add_argument("--single-device", action="store_true",
                    help="TEST ONLY (with --backend gloo): every rank builds its real engine on cuda:0, so the N > 1 orchestration — "
                         "rank 0 loads, broadcast, receivers finalize, sharded forward, gather, cross-rank check — runs with real "
                         "engines on a one-GPU box; `value` is meaningless (the ranks time-slice one GPU)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under torch.distributed.run on 127.0.0.1
    (the container hostname may not resolve) and pass their output through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle
    env.setdefault("OMP_NUM_THREADS", "8")
    argv = [x for x in sys.argv[1:] if x != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


class FakeEngine:
    """CPU stand-in with the engine's surface, for the dry run of the multi-rank orchestration only (never a product path:
    it does no inference).  Outputs are a pure function of each crop, so gathered results can be checked against one rank."""

    def __init__(self, rank):
        import torch
        self.weight_arena = torch.full((4096,), 7 if rank == 0 else 0, dtype=torch.uint8)
        self.loaded = False

    def load_state(self, *a):
        self.loaded = True

    load_smpl = load_state

    def finalize(self, assume_all_loaded=False):
        assert int(self.weight_arena.sum()) == 7 * 4096, "weight arena was not broadcast"

    def _alloc_outputs(self, B, **kw):
        return None

    def forward(self, img, outputs=None):
        import torch
        B = img.shape[0]
        key = img.reshape(B, -1)[:, :8].sum(dim=1)

        def f(n, k):
            return (key[:, None] * (torch.arange(n, dtype=torch.float32)[None] + k)).contiguous()
        return {"pred_vertices": f(6890 * 3, 1).reshape(B, 6890, 3), "pred_keypoints_3d": f(132, 2).reshape(B, 44, 3),
                "pred_keypoints_2d": f(88, 3).reshape(B, 44, 2), "rotmat": f(216, 4).reshape(B, 24, 3, 3), "betas": f(10, 5),
                "pred_cam": f(3, 6), "pred_cam_t": f(3, 7),
                "token_idx": (key[:, None].abs() * 100 + torch.arange(160)[None]).to(torch.int32) % 2048}

    def vit_forward(self, img, out=None):
        return img

    def prof_enable(self, on=True):
        pass

    def prof_collect(self):
        return {}


def usable_cpus():
    """Hardware threads this process may actually use: the affinity mask, capped by the cgroup v2 CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg, sd, tok, smpl, workload, budget_s=150.0):
    """The oracle (CPU restatement pinned bit-exact to the reference's own modules) on this host's cores, the way SURVEY.md
    §8(d) asks: B = 1 (BASELINE.json configs[0]) and B = 8 (best CPU throughput).
    Thread policy = the best of a RECORDED sweep (round 2 asserted 16): torch.set_num_threads(t) for t in {8, 16, 32, 64, 128}
    capped by the usable hardware threads (affinity mask and cgroup quota); per t one warm-up + one timed pass over 8 crops,
    guarded by a one-crop probe (an oversubscribed setting is ~50x slower: it is skipped after the probe, not after 2 minutes);
    then the median of 5 passes at the best t for B = 8 (`value`) and B = 1 (`value_b1`).  The sweep is in the line."""
    import torch
    from oracle import tokenhmr_oracle as O
    ncpu, usable = os.cpu_count(), usable_cpus()
    g = torch.Generator().manual_seed(4001)
    img = torch.randn(8, 3, 256, 256, generator=g)

    def fn(x):
        return O.forward(x, sd, tok, smpl, cfg) if workload == "full" else O.vit_forward(x, sd, cfg)

    def timed(x):
        t0 = time.perf_counter()
        fn(x)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    counts = sorted({min(t, usable) for t in (8, 16, 32, 64, 128)})
    sweep, best_probe = [], None
    with torch.no_grad():
        for t in counts:
            if time.perf_counter() - t_start > budget_s * 0.6:
                sweep.append({"threads": t, "skipped": "time budget"})
                continue
            torch.set_num_threads(t)
            probe = min(timed(img[:1]), timed(img[:1]))                     # one crop, best of two (first call warms the pool)
            if best_probe is not None and probe > 3.0 * best_probe:
                sweep.append({"threads": t, "b1_probe_crops_s": round(1.0 / probe, 3), "skipped": "one-crop probe > 3x slower than the best setting"})
                continue
            best_probe = probe if best_probe is None else min(best_probe, probe)
            timed(img)                                                      # warm-up at B = 8
            dt = timed(img)
            sweep.append({"threads": t, "b1_probe_crops_s": round(1.0 / probe, 3), "b8_crops_s": round(8.0 / dt, 3)})
        ran = [e for e in sweep if "b8_crops_s" in e]
        threads = max(ran, key=lambda e: e["b8_crops_s"])["threads"]
        torch.set_num_threads(threads)
        res = {}
        for B in (8, 1):
            fn(img[:B])
            ts = []
            for _ in range(5):
                ts.append(timed(img[:B]))
                if time.perf_counter() - t_start > budget_s:                # bound the leg on a slow host
                    break
            ts.sort()
            res[B] = (B / ts[len(ts) // 2], len(ts))
    ref_ratio = None
    try:        # how the port's rate relates to the reference's own modules, measured where the reference exists (scripts/cpu_reference_vs_port.py)
        with open(os.path.join(ROOT, "profiles", "r6_cpu_reference_vs_port.json")) as f:
            rj = json.load(f)
        ref_ratio = {"port_over_reference_b8": rj["b8"]["port_over_reference"], "port_over_reference_b1": rj["b1"]["port_over_reference"],
                     "outputs_bit_identical": rj["outputs_bit_identical"], "threads": rj["threads"],
                     "reference_crops_s_b8": rj["b8"]["reference_crops_s"], "port_crops_s_b8": rj["b8"]["port_crops_s"],
                     "where": "build container (8 cores), profiles/r6_cpu_reference_vs_port.json: recorded, not measured on this host"}
    except (OSError, ValueError, KeyError):
        pass
    return {"value": round(res[8][0], 3), "unit": "crops/s", "cores": threads, "kind": "port",
            "value_per_thread": round(res[8][0] / threads, 4),
            "kind_note": ("port = oracle/tokenhmr_oracle.py, the CPU restatement pinned bit-exact to the reference's modules — NOT the reference's own "
                          "modules: the reference is Python and cannot travel to the GPU box in any form.  The same torch CPU operators in the same "
                          "order: timed side by side where the reference exists, the port runs at `reference_ratio` of the reference's own rate"),
            "reference_ratio": ref_ratio,
            "host_cpus": ncpu,
            "usable_cpus": usable, "sweep": sweep,
            "value_b1": round(res[1][0], 3), "value_b8": round(res[8][0], 3),
            "sample": (f"oracle (torch CPU fp32, restatement pinned bit-exact to the reference modules), {workload} path, "
                       f"median of {res[8][1]} passes over 8 crops (value, value_b8) and of {res[1][1]} passes over 1 crop "
                       f"(value_b1 = BASELINE configs[0]) at torch.set_num_threads({threads}) = the best of the recorded sweep "
                       f"over {counts} threads ({usable} usable of {ncpu})")}


def parity_vs_golden(o, B, cfg, workload):
    """One reference-checked step outside the timed region: rank 0's batch IS the input of tests/golden/full_d32_b64.npz
    (64 seeded crops through the reference's own modules, oracle/gen_golden.py), so the bench line states how many of the
    10,240 pose-token indices differ from the reference's and the largest joint error, on the very run it times."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "full_d32_b64.npz")
    if not (workload == "full" and B == 64 and cfg.vit_depth == 32 and os.path.exists(path)):
        return None
    g = np.load(path)
    return _parity_block(o, g, "tests/golden/full_d32_b64.npz (reference modules, B = 64, depth 32)")


def parity_golden_file(cfg, workload):
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "full_d32_b64.npz")
    return np.load(path) if (workload == "full" and cfg.vit_depth == 32 and os.path.exists(path)) else None


class GoldenRows:
    """The first b crops of a 64-crop golden file (crops are independent on this path: no batch statistic, tokenhmr.py:146-188)."""
    PER_CROP = ("token_idx", "top2_gap", "joints", "verts_sample", "joints_f64", "verts_sample_f64")

    def __init__(self, g, b):
        self.g, self.b, self.files = g, b, [f for f in g.files if f not in ("joints_f64", "verts_sample_f64")]

    def __getitem__(self, k):
        return self.g[k][:self.b] if k in self.PER_CROP else self.g[k]


def _parity_block(o, g, against):
    import numpy as np
    idx = o["token_idx"].cpu().numpy()
    ref, gap = g["token_idx"], g["top2_gap"]
    mism = idx != ref
    j, v = o["pred_keypoints_3d"].cpu().numpy(), o["pred_vertices"].cpu().numpy()[:, ::53]
    res = {"tokens": int(ref.size), "mismatches": int(mism.sum()),
           "mismatches_where_gap_gt_1e-3": int((mism & (gap > 1e-3)).sum()),
           "smallest_reference_top2_gap": float(gap.min()),
           "max_joint_err_m": float(np.abs(j - g["joints"]).max()),
           "max_vertex_err_m": float(np.abs(v - g["verts_sample"]).max()),
           "against": against}
    if "joints_f64" in g.files:
        # the same modules of the reference evaluated in float64 (oracle/gen_golden.py): how far THIS result and the reference's own fp32
        # result sit from the value both approximate
        res["vs_reference_float64"] = {"joints_m": float(np.abs(j.astype(np.float64) - g["joints_f64"]).max()),
                                       "vertices_m": float(np.abs(v.astype(np.float64) - g["verts_sample_f64"]).max()),
                                       "reference_fp32_joints_m": float(g["ref32_vs_f64"][0]),
                                       "reference_fp32_vertices_m": float(g["ref32_vs_f64"][1])}
    return res


def parity_set(cfg, dev, mode, first):
    """The other three 64-crop depth-32 fixtures of tests/golden (two more weight / crop seeds of the default-init statistics and the
    "trained-like" state: LayerNorm gains in [0.1, 10], x50 outlier rows in proj / fc2, non-trivial mean parameters), each through its own
    engine in the timed mode, against the reference's own modules: with the timed batch's fixture 4 x 10,240 = 40,960 pose tokens.
    Untimed; tests/test_gpu_model.py::test_b64_tokens_vs_reference_golden asserts the same in both modes."""
    import numpy as np
    import torch
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.engine import Engine
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    out = {"full_d32_b64": {k: first[k] for k in ("tokens", "mismatches", "mismatches_where_gap_gt_1e-3", "max_joint_err_m", "max_vertex_err_m")}}
    for name in ("full_d32_b64_s1", "full_d32_b64_s2", "full_d32_b64_trained"):
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        seed = int(g["meta"][3])
        style = str(g["style"]) if "style" in g.files else "init"
        e = Engine(cfg, max_batch=64, device=dev)
        try:
            e.load_state(W.make_synthetic_state(cfg, seed, style), W.make_synthetic_tokenizer(cfg, seed))
            e.load_smpl(make_synthetic_smpl(cfg, seed))
            e.finalize()
            e.set_vit_gemm(mode)
            img = torch.randn(64, 3, 256, 256, generator=torch.Generator().manual_seed(4000 + seed))
            o = e.forward(img.to(dev), want_probs=False)
            torch.cuda.synchronize()
            e.status()
            blk = _parity_block(o, g, name)
            out[name] = {k: blk[k] for k in ("tokens", "mismatches", "mismatches_where_gap_gt_1e-3", "max_joint_err_m", "max_vertex_err_m")}
            if "vs_reference_float64" in blk:
                out[name]["vs_reference_float64"] = blk["vs_reference_float64"]
            out[name]["weights"] = f"seed {seed}, {style}"
        finally:
            e.close()
            del e
            torch.cuda.empty_cache()
    fx = [v for v in out.values() if isinstance(v, dict) and "tokens" in v]
    out["total"] = {"fixtures": len(fx), "tokens": sum(v["tokens"] for v in fx), "mismatches": sum(v["mismatches"] for v in fx),
                    "mismatches_where_gap_gt_1e-3": sum(v["mismatches_where_gap_gt_1e-3"] for v in fx), "vit_gemm": mode}
    return out


def lbs_at_b512(dev, smpl):
    """north_star asks for the LBS HBM rate; at B = 64 its launches are latency-bound, so it is also timed stand-alone at
    B = 512 (BASELINE configs[3]'s global batch on one device) through the same kernels (thmr_smpl handle)."""
    import torch
    from tokenhmr_amd.smpl import SMPL
    B = 512
    m = SMPL(smpl, max_batch=B, device=dev)
    g = torch.Generator().manual_seed(9)
    R = torch.linalg.qr(torch.randn(B * 24, 3, 3, generator=g))[0]
    R = (R * torch.linalg.det(R).sign()[:, None, None]).reshape(B, 24, 3, 3).to(dev)
    betas = torch.randn(B, 10, generator=g).to(dev)
    for _ in range(20):
        m(R[:, :1], R[:, 1:], betas, pose2rot=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        m(R[:, :1], R[:, 1:], betas, pose2rot=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = B * (83208 + 904) + 19.79e6                        # per crop written + read, constants once
    m.close()
    return {"batch": B, "avg_call_ms": round(ms, 4), "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "bytes": by}


def pipeline_bench(eng, dev, B, steps, warmup, fwd_ms=None):
    """The rows AROUND the path and the pipeline they exist for, measured end to end:

        decoded uint8 1080p frames on the HOST (pinned)  --H2D-->  thmr_cropper_run (blur + warp + normalise, csrc/crop.hip)
            -->  thmr_forward  -->  thmr_eval_pose (pelvis alignment, MPJPE, Procrustes, PVE: csrc/eval.hip) against resident GT

    = what tokenhmr/eval.py:124,144-149 does per batch with a 4-worker cv2 DataLoader, recursive_to, model(batch), evaluator(out, batch).
    Two streams: the copies and crops of batch i + 1 run on a side stream while batch i is in forward / eval on the launch stream
    (double-buffered frames and crops), and nothing synchronises with the host inside the loop (the evaluator accumulates on the device).
    8 frames x B / 8 people per batch; boxes of 250-900 px (the larger ones take the anti-alias blur of vitdet_dataset.py:62-68)."""
    import numpy as np
    import torch
    from tokenhmr_amd.preprocess import Cropper, gen_trans_from_patch_cv, expand_to_aspect_ratio
    from tokenhmr_amd.evaluator import eval_pose_gpu
    F = 8 if B % 8 == 0 else (4 if B % 4 == 0 else 1)
    P, H, W = B // F, 1080, 1920
    rng = np.random.default_rng(11)
    host_frames = [torch.from_numpy(rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)).pin_memory() for _ in range(2)]
    host_gtj = [torch.randn(B, 44, 4).pin_memory() for _ in range(2)]
    host_gtv = [torch.randn(B, 6890, 3).mul_(0.3).pin_memory() for _ in range(2)]
    plans, region_bytes = [], 0
    for f in range(F):
        hgt = rng.uniform(250, 900, P)
        cx, cy = rng.uniform(300, W - 300, P), rng.uniform(300, H - 300, P)
        boxes = np.stack([cx - hgt * 0.2, cy - hgt / 2, cx + hgt * 0.2, cy + hgt / 2], 1).astype(np.float32)
        center, scale = (boxes[:, 2:4] + boxes[:, 0:2]) / 2.0, (boxes[:, 2:4] - boxes[:, 0:2]) / 200.0
        tr, sg = [], []
        for i in range(P):
            bs = expand_to_aspect_ratio(scale[i] * 200, target_aspect_ratio=[192, 256]).max()
            fct = (float(bs) / 256) / 2.0
            sg.append((fct - 1) / 2 if fct > 1.1 else 0.0)
            tr.append(gen_trans_from_patch_cv(center[i][0], center[i][1], bs, bs, 256, 256, 1.0, 0))
            region_bytes += int(min(float(bs), H) * min(float(bs), W)) * 3
        plans.append((np.stack(tr), sg))
    crop_stream = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    cropper = Cropper(dev)
    dfr = [torch.empty(F, H, W, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
    dgj = [torch.empty(B, 44, 4, device=dev) for _ in range(2)]
    dgv = [torch.empty(B, 6890, 3, device=dev) for _ in range(2)]
    img = [torch.empty(B, 3, 256, 256, device=dev) for _ in range(2)]
    outs = eng._alloc_outputs(B, taps=False, want_probs=True)
    kp_list, pelvis = list(range(25, 38)) + [43], 39                 # 3DPW: datasets_eval.yaml:12, experiment/default.yaml:15
    acc = torch.zeros(3, device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    h2d_ev = []

    def stage_in(i, timed=False):
        """side stream: H2D of batch i's frames + GT, then its crops into img[i % 2]"""
        k = i % 2
        with torch.cuda.stream(crop_stream):
            if i >= 2:
                crop_stream.wait_event(consumed[k])                  # batch i - 2 is done with these buffers
            e0 = e1 = None
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(crop_stream)
            dfr[k].copy_(host_frames[k], non_blocking=True)
            dgj[k].copy_(host_gtj[k], non_blocking=True)
            dgv[k].copy_(host_gtv[k], non_blocking=True)
            if timed:
                e1.record(crop_stream)
                h2d_ev.append((e0, e1))
            for f in range(F):
                cropper.warp(dfr[k][f], plans[f][0], plans[f][1], truncate=4.0, out=img[k][f * P:(f + 1) * P])
            ready[k].record(crop_stream)

    def run(n, timed=False):
        stage_in(0, timed)
        for i in range(n):
            k = i % 2
            if i + 1 < n:
                stage_in(i + 1, timed)
            main.wait_event(ready[k])
            o = eng.forward(img[k], outputs=outs)
            mp, re, pve = eval_pose_gpu(o["pred_keypoints_3d"], dgj[k], kp_list, pelvis, 0, o["pred_vertices"], dgv[k])
            acc.add_(torch.stack([mp.sum(), re.sum(), pve.sum()]))
            consumed[k].record(main)

    run(max(2, warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, timed=True)
    torch.cuda.synchronize()
    pipe_ms = (time.perf_counter() - t0) / steps * 1e3
    h2d_ms = sorted(e0.elapsed_time(e1) for e0, e1 in h2d_ev)
    # forward only, same engine, crops resident (the bench's own timed region when it is handed over)
    if fwd_ms is None:
        for _ in range(2):
            eng.forward(img[0], outputs=outs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward(img[0], outputs=outs)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / steps * 1e3

    def alone(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def crops():
        for f in range(F):
            cropper.warp(dfr[0][f], plans[f][0], plans[f][1], truncate=4.0, out=img[0][f * P:(f + 1) * P])
    o = eng.forward(img[0], outputs=outs)
    crop_ms = alone(crops)
    eval_ms = alone(lambda: eval_pose_gpu(o["pred_keypoints_3d"], dgj[0], kp_list, pelvis, 0, o["pred_vertices"], dgv[0]))
    crop_bytes = B * 3 * 256 * 256 * 4 + region_bytes               # crops written + the frame regions under the boxes read once
    eval_bytes = 2 * B * 6890 * 3 * 4 + 2 * B * 44 * 4 * 4
    h2d_bytes = host_frames[0].numel() + host_gtj[0].numel() * 4 + host_gtv[0].numel() * 4
    res = {"what": ("uint8 1080p frames on the host -> H2D + thmr_cropper_run on a side stream -> thmr_forward -> thmr_eval_pose, double-buffered, "
                    "no host synchronisation inside the loop (untimed extra; eval.py:124,144-149)"),
           "frames_per_batch": F, "people_per_frame": P, "steps": steps,
           "crops_per_s": round(B / (pipe_ms * 1e-3), 2), "ms_per_batch": round(pipe_ms, 3),
           "forward_only_ms_per_batch": round(fwd_ms, 3), "vs_forward_only": round(fwd_ms / pipe_ms, 4),
           "h2d": {"ms_per_batch_median": round(h2d_ms[len(h2d_ms) // 2], 3), "bytes": int(h2d_bytes),
                   "GB_per_s": round(h2d_bytes / (h2d_ms[len(h2d_ms) // 2] * 1e-3) / 1e9, 1), "overlapped": "side stream, under the previous batch"},
           "crop_kernels": {"ms_per_batch_alone": round(crop_ms, 3), "bytes": int(crop_bytes), "GB_per_s": round(crop_bytes / (crop_ms * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(crop_bytes / (crop_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launches": F},
           "eval_kernels": {"ms_per_batch_alone": round(eval_ms, 4), "bytes": int(eval_bytes), "GB_per_s": round(eval_bytes / (eval_ms * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(eval_bytes / (eval_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
           "metrics_accumulated_on_device": [round(float(v), 3) for v in (acc / max(1, (max(2, warmup) + steps) * B)).tolist()]}
    try:
        pose = torch.randn(B, 21, 6, device=dev)
        enc_ms = alone(lambda: eng.encode_tokens(pose))
        res["encode_tokens"] = {"ms_per_batch_alone": round(enc_ms, 3), "poses_per_s": round(B / (enc_ms * 1e-3), 1),
                                "what": "thmr_encode_tokens: PoseSPEncoderV1 + argmin-L2 quantiser (vanilla_pose_vqvae.py:334-342), B poses"}
    except Exception as ex:      # encoder half not loaded
        res["encode_tokens"] = {"skipped": f"{type(ex).__name__}: {ex}"[:120]}
    cropper.close()
    return res


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.self_launch):
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ                         # launched by torch.distributed.run (also valid at N = 1)
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch `python bench.py --gpus {a.gpus}` (it starts its own "
                 f"ranks) or `python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus}`")
    cpu_dry = a.fake_engine
    if cpu_dry and a.backend != "gloo":
        sys.exit("--fake-engine is the CPU dry run of the orchestration: use it with --backend gloo")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if a.single_device and a.backend != "gloo":
        sys.exit("--single-device stacks the ranks on one GPU, which RCCL refuses: use it with --backend gloo")
    dev_index = 0 if a.single_device else local_rank
    dev = torch.device("cpu") if cpu_dry else torch.device("cuda", dev_index)
    if not cpu_dry:
        torch.cuda.set_device(dev)

    def sync():
        if not cpu_dry:
            torch.cuda.synchronize()

    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W, dist as D
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl

    cfg = HMRConfig(vit_depth=a.vit_depth)
    # per-rank crops: weak scaling (--batch per GPU, the driver's form) or one global batch dealt in contiguous, balanced shards
    sizes = D.shard_sizes(a.global_batch, world) if a.global_batch else [a.batch] * world
    if min(sizes) < 1:
        sys.exit(f"bench.py: --global-batch {a.global_batch} leaves a rank without crops at {world} ranks")
    B, total_batch = sizes[rank], sum(sizes)
    crop0 = sum(sizes[:rank])                                  # this rank's first crop in the global batch
    build_info = None
    # LOCAL_RANK -> device: every rank of the node must sit on its own GPU (a launcher that exports a wrong LOCAL_RANK would
    # silently stack ranks on one device and the "scaling" would be time-slicing)
    ident = {"rank": rank, "local_rank": local_rank, "device": None if cpu_dry else torch.cuda.current_device(),
             "uuid": None if cpu_dry else str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")) or None,
             "host": socket.gethostname()}
    if not cpu_dry:
        assert torch.cuda.current_device() == dev_index, f"rank {rank}: current device {torch.cuda.current_device()} != LOCAL_RANK {local_rank}"
    idents = [ident]
    if use_dist:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if not cpu_dry and not a.single_device:
            devs = [(i["host"], i["device"]) for i in idents]
            assert len(set(devs)) == world, f"ranks share a GPU: {devs}"
    if cpu_dry:
        eng = FakeEngine(rank)
    else:
        # ALWAYS bring the library up to date first (incremental by content hash; verifies the loaded .so reports the hash of
        # the current sources).  Rank 0 builds, the others wait, so N ranks never race on the object files.
        import __graft_entry__
        if rank == 0:
            __graft_entry__.build()
        if use_dist:
            dist.barrier()
        from tokenhmr_amd import _cabi
        from tokenhmr_amd.engine import Engine
        build_info = _cabi.load().thmr_build_info().decode()
        assert f"src:{__graft_entry__.source_hash()}" in build_info, f"stale libtokenhmr_hip.so: {build_info}"
        eng = Engine(cfg, max_batch=max(sizes), device=dev)
    sd = tok = smpl = None
    read_ckpt = rank == 0                                      # ONLY rank 0 touches the "checkpoint"; the line lists who did
    if rank == 0:
        # rank 0 "reads the checkpoint" (synthetic: no network for real weights) ...
        if not cpu_dry:
            sd, tok = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0)
            tok = dict(tok, **W.make_synthetic_encoder(cfg, 0))       # the tokenizer's encoder half (N4): lets the pipeline extra time thmr_encode_tokens
        smpl = make_synthetic_smpl(cfg, 0)
        eng.load_state(sd, tok)
        eng.load_smpl(smpl)
    bcast_ms = None
    if use_dist:
        # The arena is complete for a broadcast once load_state / load_smpl have returned (thmr_load_weights writes the index
        # tables and flag words; round 2 wrote them in finalize and shipped uninitialised tables to the other ranks): the root
        # may finalize before or after.  Receivers finalize with assume_all_loaded, which REQUIRES the loader's magic word.
        sync()
        dist.barrier()
        t_b = time.perf_counter()
        D.broadcast_weights(eng, src=0)        # ... and ONE RCCL broadcast replicates the packed arena
        sync()
        bcast_ms = (time.perf_counter() - t_b) * 1e3
    eng.finalize(assume_all_loaded=(rank != 0))
    # The timed mode is the ENGINE'S OWN DEFAULT (split3 since round 5 / ABI 4): without --vit-gemm nothing here touches the switch, so
    # `value` is what load_tokenhmr() / thmr_create hand a caller who passes no mode.  Below 3 crops per GPU that default runs the
    # exact-fp32 kernels, and the line says "f32".
    mode_source = "--vit-gemm" if a.vit_gemm is not None else "engine default (no thmr_set_vit_gemm call)"
    if a.vit_gemm is None:
        a.vit_gemm = "split3" if (not cpu_dry and min(sizes) >= 3) else "f32"
    split_mode = a.vit_gemm == "split3"
    if split_mode and (cpu_dry or min(sizes) < 3):
        sys.exit("--vit-gemm split3 needs real engines and at least 3 crops per GPU (below, the mode runs the exact-fp32 kernels)")
    if not cpu_dry:
        if mode_source == "--vit-gemm" and eng.vit_gemm() != a.vit_gemm:
            eng.set_vit_gemm(a.vit_gemm)
        assert min(sizes) < 3 or eng.vit_gemm() == a.vit_gemm, (eng.vit_gemm(), a.vit_gemm)

    def crops_of(r):
        """rank r's shard of the global batch: seeded per rank (rank 0 at 64 crops = tests/golden/full_d32_b64.npz), so any
        rank can regenerate any other rank's crops for the cross-rank check below"""
        gen = torch.Generator().manual_seed(4000 + r)
        shp = (sizes[r], 3, 8, 8) if cpu_dry else (sizes[r], 3, 256, 256)
        return torch.randn(*shp, generator=gen)

    if a.workload == "pipeline":
        if cpu_dry or world != 1:
            sys.exit("--workload pipeline runs on one GPU with a real engine")
        res = pipeline_bench(eng, dev, B, a.steps, a.warmup)
        eng.status()
        print(json.dumps({"metric": "crops_per_sec", "value": res["crops_per_s"], "unit": "crops/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": res["ms_per_batch"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32 (ViT GEMM operands as 3 x bf16 pieces)" if split_mode else "f32", "data": "synthetic",
                          "config": {"workload": "frames -> crops -> TokenHMR full path -> evaluator pipeline (NOT the headline workload)", "batch_per_gpu": B,
                                     "vit_depth": cfg.vit_depth},
                          "pipeline": res, "build": build_info}), flush=True)
        return
    img = crops_of(rank).to(dev)                         # resident in HBM before timing
    outs = eng._alloc_outputs(B, taps=False, want_probs=True)
    feats = None if cpu_dry else torch.empty(B, 192, 1280, device=dev)
    gather = use_dist and not a.no_gather and a.workload == "full"

    pending = []        # the previous step's all-gather, still in flight on the RCCL stream
    last = {}
    # how long the compute stream actually WAITS for the gather it joins (0 when the xGMI traffic hid under the next ViT):
    # HIP events on the launch stream around the wait (RCCL's wait() is a stream dependency, it does not block the host); on the
    # CPU dry run the host clock
    gw = {"ev": [], "host_s": 0.0, "on": False}

    def join():
        while pending:
            h = pending.pop()
            if gw["on"] and not cpu_dry:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                last["records"] = h.wait()
                e1.record()
                gw["ev"].append((e0, e1))
            else:
                t_w = time.perf_counter()
                last["records"] = h.wait()
                if gw["on"]:
                    gw["host_s"] += time.perf_counter() - t_w

    def in_turns(fn):
        """--single-device only: the ranks share ONE GPU, and the engine's persistent kernels (decoder grid barrier, split3 hand-over) need all
        their workgroups resident together — two processes launching them at once can starve each other until the bounded waits time out
        (the in-process turnstile cannot see another process).  So each rank runs its forward alone, in rank order, between barriers.  On
        real multi-GPU runs every rank has its own device and this is a plain call."""
        if not a.single_device:
            return fn()
        res = None
        for r in range(world):
            if r == rank:
                res = fn()
                sync()
            dist.barrier()
        return res

    def step():
        if a.workload == "vit":
            in_turns(lambda: eng.vit_forward(img, out=feats))
            return
        o = in_turns(lambda: eng.forward(img, outputs=outs))
        last["out"] = o
        if gather:
            # packed per-crop records of this step go out over xGMI while the next step's ViT runs; the previous step's
            # gather is joined first, so at most one collective is in flight and every step's records are complete by the
            # time the next-but-one step starts (and all of them before the timed region closes)
            rec = D.pack_records(o)
            join()
            pending.append(D.all_gather_records(rec, total_batch, async_op=True))

    for _ in range(a.warmup):
        step()
    join()
    sync()
    # HIP events around the DOMINANT kernel only (fc1: the GEMM class with the largest share of every step), every 4th of its 32
    # launches per call (one shape): an event pair costs ~2-3 us of stream time — instrumenting the four GEMM classes (round 2)
    # cost 0.75 % of a B = 64 step and 20 % of a B = 1 call.  Its roofline is measured live in the timed region; the other classes
    # come from a separate untimed pass below.
    eng.prof_enable("fc1")
    gw["on"] = True
    ev = [] if cpu_dry else [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for i in range(a.steps):
        step()
        if ev:
            ev[i + 1].record()                      # on the stream the engine launches on (torch's current stream)
    join()                      # the last step's records must have arrived inside the timed region
    sync()
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    gw["on"] = False
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)) if ev else []
    gather_wait_ms = (sum(e0.elapsed_time(e1) for e0, e1 in gw["ev"]) if gw["ev"] else gw["host_s"] * 1e3) / max(1, a.steps)
    eng.prof_enable(False)
    prof = eng.prof_collect()
    if not cpu_dry:
        eng.status()            # no asynchronous device-side error in the timed region (bounded grid barrier of the decoder kernel)
    prof_all, breakdown_steps = {}, 0
    if not a.no_extras and not cpu_dry:
        breakdown_steps = min(2, a.steps)
        eng.prof_enable(True)                       # untimed: every kernel class instrumented, for classes_ms_per_step
        for _ in range(breakdown_steps):
            step()
        join()
        sync()
        eng.prof_enable(False)
        prof_all = eng.prof_collect()
    gathered_ok, cross = None, None
    if gather and "records" in last:
        # every rank must hold all ranks' crops in crop order, and its own rows must equal what it just computed
        rec = last["records"]
        mine = D.pack_records(last["out"])
        gathered_ok = bool(rec.shape[0] == total_batch and torch.equal(rec[crop0:crop0 + B], mine))
        t = torch.tensor([1.0 if gathered_ok else 0.0], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gathered_ok = bool(t.item() == 1.0)
        if rank == 0 and world > 1 and not a.no_extras:
            # ... and the OTHER ranks' rows must be what a correctly loaded model computes: rank 0 (which read the checkpoint)
            # regenerates each rank's seeded crops, runs them through its own engine and compares the gathered rows bit for bit —
            # same kernels, same arithmetic, another GPU.  This is the check that catches a receiver whose broadcast arena was
            # incomplete (the round-2 regression: uninitialised resample tables on ranks != 0), which "my own rows equal what I
            # computed" cannot.
            worst, same = 0.0, True
            for r in range(1, world):
                x_r = crops_of(r).to(dev)
                o_r = eng.forward(x_r) if cpu_dry else eng.forward(x_r, want_probs=False)
                want = D.pack_records(o_r)
                got = rec[sum(sizes[:r]):sum(sizes[:r + 1])]
                same = same and torch.equal(got, want)
                worst = max(worst, float((got[:, :20670 + 132] - want[:, :20670 + 132]).abs().max()))
            sync()
            cross = {"ranks_checked": world - 1, "bit_identical": bool(same), "max_abs_diff_verts_joints_m": worst,
                     "how": "rank 0 recomputed every other rank's seeded shard on its own engine and compared the gathered records"}
    # This rank's shard ALONE — same engine, same resident crops, no collective, no barrier — right after the timed region: the sum over ranks is
    # what N GPUs deliver if the gather and the barriers cost nothing at THIS shard size (`multi_gpu.expected`), so a poor first 8-GPU curve
    # separates "small shards run slower per crop" (see batch_sweep: 4 crops per GPU = 0.63 of the 64-crop rate) from "the gather / broadcast hurts"
    local_rate = None
    if use_dist and not cpu_dry and a.workload == "full":
        def local():
            n_loc = max(3, min(a.steps, 10))
            for _ in range(2):
                eng.forward(img, outputs=outs)
            sync()
            t_l = time.perf_counter()
            for _ in range(n_loc):
                eng.forward(img, outputs=outs)
            sync()
            return B * n_loc / (time.perf_counter() - t_l)
        local_rate = in_turns(local)
    rank_step = None
    if use_dist:
        t = torch.tensor([elapsed], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank view of the timed region, so that a slow first 8-GPU run diagnoses itself: median step time on each rank's
        # launch stream, exposed gather wait, and who read the checkpoint
        mine = {"rank": rank, "step_ms_median": round(step_ms[len(step_ms) // 2], 3) if step_ms else None,
                "gather_wait_ms_per_step": round(gather_wait_ms, 4), "read_checkpoint": bool(read_ckpt), "crops": B,
                "local_crops_per_s": round(local_rate, 2) if local_rate else None,
                "bcast_ms": round(bcast_ms, 1) if bcast_ms is not None else None}
        rank_step = [None] * world
        dist.all_gather_object(rank_step, mine)

    if rank == 0:
        total_crops = total_batch * a.steps
        value = total_crops / elapsed
        # dominant kernel = the GEMM class with the largest share of the timed region
        gemms = {k: v for k, v in prof.items() if k.startswith("gemm_") and v["launches"] > 0}
        dom = max(gemms, key=lambda k: gemms[k]["ms"]) if gemms else None
        roof = None
        if dom:
            d = gemms[dom]
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            # all four ViT GEMM classes: from the untimed fully instrumented pass (only fc1 carries events in the timed region)
            g_all = {k: v for k, v in prof_all.items() if k.startswith("gemm_") and v["launches"] > 0} or gemms
            all_ms = sum(v["ms"] for v in g_all.values())
            all_tf = sum(v["flops"] for v in g_all.values()) / (all_ms * 1e-3) / 1e12
            # split3: the dominant kernel is gemm_split16_kernel on the bf16 matrix pipe, which executes SIX bf16 MFMA flops per
            # fp32-equivalent flop: achieved / peak are bf16 MFMA TFLOP/s there (the fp32-equivalent rate beside them)
            pk, mul = (PEAK_BF16_MFMA_TFLOPS, 6.0) if split_mode else (PEAK_F32_MFMA_TFLOPS, 1.0)
            kern = ("gemm_split16_tail_kernel / gemm_split16_kernel<GELU, split3 output> on v_mfma_f32_16x16x32_bf16" if split_mode else "gemm_f32_kernel")
            roof = {"bound": "mfma", "kernel": f"{kern} ({dom})", "achieved": round(tf * mul, 2),
                    "peak": pk, "unit": "TFLOP/s",
                    "frac": round(tf * mul / pk, 4),
                    "traffic": None, "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches": d["launches"],
                    "flops_per_launch": d["flops"] / d["launches"] * mul,
                    "flops_counted": ("bf16 MFMA flops executed: 6 piece products x 2 M N K (the fp32-equivalent 2 M N K rate is f32_equivalent_tflops)"
                                      if split_mode else "2 M N K"),
                    "all_gemm_achieved": round(all_tf * mul, 2), "all_gemm_frac": round(all_tf * mul / pk, 4),
                    "f32_equivalent_tflops": round(tf, 2),
                    "path_tflops": round(value / world * GFLOP_PER_CROP[a.workload] / 1e3, 2)}
            # HBM traffic of that kernel from PMC counters (separate rocprofv3 --pmc passes on the round's final build, committed under
            # profiles/; FETCH_SIZE doubled per the gfx950 correction) — cannot be collected live inside this process, so it is a
            # RECORDED number and labelled as such
            alg = {"gemm_fc1": (6.0 if split_mode else 4.0) * (12288 * 1280 + 5120 * 1280 + 12288 * 5120)}
            for pmc_file in ("r6_final_pmc.json", "r5_final_pmc.json", "r4_final_pmc.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                        pj = json.load(f)
                except (OSError, ValueError):
                    continue
                # fc1 of a 64-crop batch runs the mixed grid (half-tile tail) since round 5; the round-4 file knows only the plain grid
                # (round 6 added template parameters: the same two kernels under their new names first)
                keys = (["gemm_split16_tail_kernel<2, false, false>", "gemm_split16_kernel<4, 4, 2, false, false, 2, false>",
                         "gemm_split16_tail_kernel<2, false>", "gemm_split16_kernel<4, 2, false, false>"] if split_mode else ["gemm_f32_kernel"]) if dom == "gemm_fc1" else []
                pmc = next((pj[k] for k in keys if k in pj), None)
                if pmc and a.batch == 64 and a.workload in ("full", "vit"):
                    roof["traffic"] = round(pmc["traffic_bytes"])
                    roof["traffic_source"] = (f"profiles/{pmc_file} (rocprofv3 --pmc pass of this kernel at B = 64: FETCH_SIZE*2 + WRITE_SIZE, "
                                              f"bytes/launch; recorded, not live; build {pj.get('_build', 'n/a')})")
                    roof["algorithmic_bytes_per_launch"] = alg[dom]
                    if "mfma_util_profiled" in pmc:
                        roof["pmc_mfma_util"] = pmc["mfma_util_profiled"]
                    break
            if prof_all:
                roof["classes_ms_per_step"] = {k: round(v["ms"] / breakdown_steps, 3) for k, v in prof_all.items() if v["launches"]}
                roof["classes_launches_per_step"] = {k: v["launches"] // breakdown_steps for k, v in prof_all.items() if v["launches"]}
                roof["classes_note"] = (f"per-class split from a separate untimed pass of {breakdown_steps} steps with every launch "
                                        "instrumented; achieved/avg_launch_ms are from the timed region")
                at = prof_all.get("attention")
                if at and at["launches"]:
                    atf = at["flops"] / (at["ms"] * 1e-3) / 1e12
                    # split3 mode (>= 3 crops): csrc/attention_b16.hip — q k^T and p v as six bf16 MFMA products per element pair
                    a_pk, a_mul = (PEAK_BF16_MFMA_TFLOPS, 6.0) if (split_mode and B >= 3) else (PEAK_F32_MFMA_TFLOPS, 1.0)
                    roof["attention"] = {"bound": "mfma", "achieved": round(atf * a_mul, 2), "peak": a_pk, "unit": "TFLOP/s",
                                         "frac": round(atf * a_mul / a_pk, 4), "avg_launch_ms": round(at["ms"] / at["launches"], 4),
                                         "f32_equivalent_tflops": round(atf, 2),
                                         "kernel": ("vit_attention_b16_kernel on v_mfma_f32_16x16x32_bf16 (6 piece products x 4 N N d flops)"
                                                    if a_mul > 1 else "vit_attention_*_kernel on v_mfma_f32_16x16x4_f32")}
                pe = prof_all.get("patch_embed")
                if pe and pe["launches"]:   # north_star asks for the patch-embed HBM rate too (it is MFMA/latency-bound: AI 240 flop/B)
                    gbs = pe["bytes"] / (pe["ms"] * 1e-3) / 1e9
                    roof["patch_embed_hbm"] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_ms": round(pe["ms"] / pe["launches"], 4),
                                               "tflops": round(pe["flops"] / (pe["ms"] * 1e-3) / 1e12, 1)}
                lbs = prof_all.get("lbs")
                if lbs and lbs["launches"]:
                    gbs = lbs["bytes"] / (lbs["ms"] * 1e-3) / 1e9
                    roof["lbs_hbm"] = {"batch": B, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_ms": round(lbs["ms"] / lbs["launches"], 4),
                                       "bound": ("NOT hbm: three dependent launches, latency + LDS-bound (DESIGN.md 3.5); the rate is reported because "
                                                 "north_star asks for it, it is not a fraction of a roof that binds this stage")}
                    if world == 1:
                        roof["lbs_hbm_b512"] = lbs_at_b512(dev, smpl)
        par, facade = None, None
        if not a.no_extras and not cpu_dry and "out" in last:
            par = parity_vs_golden(last["out"], B, cfg, a.workload)
            if world == 1:
                # the drop-in's own rate: `model(batch)` through the facade (tokenhmr_amd/model.py) allocates its output tensors per
                # call (84 MB of cls_logits_softmax at 64 crops) where the timed region above reuses pre-allocated ones
                from tokenhmr_amd.model import TokenHMR
                model = TokenHMR.from_engine(eng)
                for _ in range(3):
                    model({"img": img})
                sync()
                n_f = max(5, min(a.steps, 20))
                t_f = time.perf_counter()
                for _ in range(n_f):
                    model({"img": img})
                sync()
                f_ms = (time.perf_counter() - t_f) / n_f * 1e3
                facade = {"ms_per_call": round(f_ms, 3), "crops_per_s": round(B / (f_ms * 1e-3), 2), "calls": n_f,
                          "vs_engine_forward": round((elapsed / a.steps * 1e3) / f_ms, 4),
                          "vit_gemm": eng.vit_gemm(), "mode_set_by": mode_source,
                          "what": ("TokenHMR facade model({'img': ...}) -> dict, outputs allocated per call (untimed extra); the mode is the one "
                                   "the timed region ran — by default the engine's creation default, i.e. what load_tokenhmr() without a mode "
                                   "argument delivers (tests/test_gpu_checkpoint_files.py asserts that on the reference's file formats)")}
        pipeline = None
        if world == 1 and not a.no_extras and not cpu_dry and a.workload == "full":
            try:
                pipeline = {a.vit_gemm: pipeline_bench(eng, dev, B, max(5, min(a.steps, 20)), 2, elapsed / a.steps * 1e3)}
            except Exception as ex:      # an extra must never cost the headline line
                pipeline = {"error": f"{type(ex).__name__}: {ex}"}
        sweep = None
        if world == 1 and not a.no_extras and not cpu_dry and a.workload == "full" and B >= 32:
            # The reference's OWN batch sizes through the same engine in the timed mode (untimed extra): 1 crop (BASELINE configs[0], a
            # single detection), 8 (demo.py:70 DataLoader batch_size), 32 (README.md:316 eval batch) — and the timed batch again, by the
            # same method, so the ratios are same-process, same-box.  HIP events on the launch stream, pre-allocated outputs.
            try:
                rows = []
                gold = parity_golden_file(cfg, a.workload) if B == 64 else None      # rank 0's batch IS the fixture's input: rows [:b] are its first b crops
                for b in (1, 2, 4, 8, 16, 32, B):
                    ob = eng._alloc_outputs(b, taps=False, want_probs=True)
                    xb = img[:b].contiguous()
                    n_it = 30 if b <= 8 else max(5, min(a.steps, 20))
                    for _ in range(3):
                        eng.forward(xb, outputs=ob)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n_it):
                        eng.forward(xb, outputs=ob)
                    e1.record()
                    sync()
                    ms = e0.elapsed_time(e1) / n_it
                    row = {"batch": b, "ms_per_call": round(ms, 3), "crops_per_s": round(b / (ms * 1e-3), 2), "calls": n_it}
                    if gold is not None:
                        # depth-32 parity of THIS batch size against the reference's own modules (crops are independent: the fixture's
                        # first b rows are the reference's answer for b crops alone) — every regime of the K sums, not only 64 crops
                        blk = _parity_block(ob, GoldenRows(gold, b), f"tests/golden/full_d32_b64.npz rows [:{b}]")
                        row["parity"] = {k: blk[k] for k in ("tokens", "mismatches", "mismatches_where_gap_gt_1e-3", "max_joint_err_m", "max_vertex_err_m")}
                    rows.append(row)
                eng.status()
                ref_row = rows[-1]["crops_per_s"]
                for r in rows:
                    r["vs_timed_batch"] = round(r["crops_per_s"] / ref_row, 4)
                sweep = {"vit_gemm": eng.vit_gemm(), "rows": rows,
                         "timed_batch_row_vs_value": round(ref_row / value, 4),
                         "what": ("untimed extra, run right after the timed region (before the other mode's pass): the reference's own batch sizes "
                                  "(1 = one detection; 8 = demo.py:70; 32 = README.md:316), the sizes between them and the timed batch again by the "
                                  "same method (`timed_batch_row_vs_value` = that row over `value`), same engine and mode, back-to-back calls on "
                                  "resident crops; `parity` per row = token indices / joints / vertices of those b crops against the reference "
                                  "modules' depth-32 golden; one and two crops run the exact-fp32 small-batch kernels in either mode")}
            except Exception as ex:      # an extra must never cost the headline line
                sweep = {"error": f"{type(ex).__name__}: {ex}"}
        other, other_name = None, ("exact_f32_mode" if split_mode else "split3_mode")
        if world == 1 and not a.no_extras and not cpu_dry and B >= 3:
            # SECOND measurement, not `value`: the same K steps with the ViT GEMMs in the engine's OTHER mode (thmr_set_vit_gemm), so one
            # line from one process on one box carries both: the exact-fp32 MFMA path beside the split3 one.  Each with its own parity
            # block against the reference golden, its own class split and its dominant kernel's roofline.
            o_mode = "f32" if split_mode else "split3"
            try:
                eng.set_vit_gemm(o_mode)
                for _ in range(max(2, min(a.warmup, 5))):
                    step()
                sync()
                t_s = time.perf_counter()
                for _ in range(a.steps):
                    step()
                sync()
                s_ms = (time.perf_counter() - t_s) / a.steps * 1e3
                s_par = parity_vs_golden(last["out"], B, cfg, a.workload) if "out" in last else None
                if isinstance(pipeline, dict) and "error" not in pipeline and a.workload == "full":
                    try:
                        pipeline[o_mode] = pipeline_bench(eng, dev, B, max(5, min(a.steps, 20)), 2, s_ms)
                    except Exception as ex:
                        pipeline[o_mode] = {"error": f"{type(ex).__name__}: {ex}"}
                eng.prof_enable(True)
                step()
                sync()
                eng.prof_enable(False)
                s_prof = eng.prof_collect()
                eng.status()
                eng.set_vit_gemm(a.vit_gemm)
                g4 = {k: v for k, v in s_prof.items() if k.startswith("gemm_") and v["launches"]}
                g_ms, g_fl = sum(v["ms"] for v in g4.values()), sum(v["flops"] for v in g4.values())
                fc1 = s_prof.get("gemm_fc1")
                o_pk, o_mul = (PEAK_F32_MFMA_TFLOPS, 1.0) if split_mode else (PEAK_BF16_MFMA_TFLOPS, 6.0)
                fc1_tf = fc1["flops"] / (fc1["ms"] * 1e-3) / 1e12 if fc1 and fc1["launches"] else None
                other = {"value": round(B / (s_ms * 1e-3), 2), "unit": "crops/s", "ms_per_step": round(s_ms, 3), "steps": a.steps,
                         "vs_timed_mode": round((elapsed / a.steps * 1e3) / s_ms, 4),
                         "dtype": ("f32 (fp32 MFMA pipe)" if split_mode else
                                   "f32 operands as 3 x bf16 pieces, 6 bf16 MFMA products per pair, f32 accumulate (ViT GEMMs only)"),
                         "parity": s_par,
                         "classes_ms_per_step": {k: round(v["ms"], 3) for k, v in s_prof.items() if v["launches"]},
                         "roofline": {"bound": "mfma",
                                      "kernel": ("gemm_f32_kernel (gemm_fc1)" if split_mode else
                                                 "gemm_split16_kernel<GELU, split3 output> on v_mfma_f32_16x16x32_bf16 (gemm_fc1)"),
                                      "achieved": round(o_mul * fc1_tf, 1) if fc1_tf else None, "peak": o_pk, "unit": "TFLOP/s",
                                      "frac": round(o_mul * fc1_tf / o_pk, 4) if fc1_tf else None,
                                      "f32_equivalent_tflops": round(fc1_tf, 1) if fc1_tf else None,
                                      "all_gemm_f32_equivalent_tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 1) if g_ms else None},
                         "what": (f"the engine's other ViT GEMM mode (thmr_set_vit_gemm / --vit-gemm {o_mode}), same process, same box, same K steps, "
                                  "measured after the timed region; NOT `value`")}
            except Exception as ex:      # an extra must never cost the headline line
                other = {"error": f"{type(ex).__name__}: {ex}"}
                try:
                    eng.prof_enable(False)
                    eng.set_vit_gemm(a.vit_gemm)
                except Exception:
                    pass
        if par is not None and world == 1 and not a.no_extras and not cpu_dry and a.workload == "full" and B == 64 and cfg.vit_depth == 32:
            try:
                par["set"] = parity_set(cfg, dev, a.vit_gemm, par)
            except Exception as ex:
                par["set"] = {"error": f"{type(ex).__name__}: {ex}"}
        cpu = None
        if world == 1 and not a.no_cpu_baseline and not cpu_dry:
            cpu = cpu_baseline(cfg, sd, tok, smpl, a.workload)
        line = {
            "metric": "crops_per_sec", "value": round(value, 2), "unit": "crops/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if a.global_batch else "weak", "vs_baseline": None,
            "dtype": ("f32 (fp32 in / accumulate / out; the ViT GEMMs multiply each fp32 operand as three bf16 pieces on the bf16 MFMA pipe, six products "
                      "per pair: `vit_gemm` split3; the fp32-MFMA mode is `exact_f32_mode`)") if split_mode else "f32",
            "vit_gemm": a.vit_gemm, "vit_gemm_set_by": mode_source,
            "data": "synthetic",
            "config": {"workload": ("TokenHMR full path (ViT-H/16 + 6-layer token decoder + VQ lookup/decode + SMPL LBS), "
                                    "256x256 crops, random-init weights" if a.workload == "full" else
                                    "ViT-H/16 encoder only, 256x256 crops, random-init weights"),
                       "batch_per_gpu": B if not a.global_batch else sizes, "global_batch": total_batch, "vit_depth": cfg.vit_depth,
                       "parallelism": f"dp{world}", "allgather_outputs": bool(gather), "vit_gemm": a.vit_gemm,
                       "ranks": (dist.get_world_size() if use_dist else 1),
                       "backend": (dist.get_backend() if use_dist else None)},
            "roofline": roof, "cpu_baseline": cpu, "parity": par,
        }
        if facade:
            line["facade"] = facade
        if other:
            line[other_name] = other
        if pipeline:
            line["pipeline"] = pipeline
        if sweep:
            line["batch_sweep"] = sweep
        if step_ms:
            # per-step HIP-event durations on the launch stream (median of >= 20 timed iterations);
            # `value` / `ms_per_step` stay the barrier-bracketed wall-clock numbers the driver cross-checks
            med = step_ms[len(step_ms) // 2]
            line["step_ms"] = {"median": round(med, 3), "min": round(step_ms[0], 3), "max": round(step_ms[-1], 3), "n": len(step_ms)}
            line["value_at_median_step"] = round(total_batch / (med * 1e-3), 2)
        if gathered_ok is not None:
            line["gathered_records_ok"] = gathered_ok
        if use_dist:
            meds = [r["step_ms_median"] for r in rank_step if r["step_ms_median"] is not None]
            line["multi_gpu"] = {
                "bcast_ms": max(r["bcast_ms"] for r in rank_step) if bcast_ms is not None else None,
                "bcast_bytes": int(eng.weight_arena.numel()),
                "gather_ms_exposed": max(r["gather_wait_ms_per_step"] for r in rank_step),
                "gather_bytes_per_step": int(total_batch * D.RECORD_WORDS * 4) if gather else 0,
                "rank_step_ms": {"min": min(meds), "max": max(meds)} if meds else None,
                "per_rank": rank_step,
                "checkpoint_readers": [r["rank"] for r in rank_step if r["read_checkpoint"]],
                "devices": [{k: i[k] for k in ("rank", "local_rank", "device", "uuid")} for i in idents],
                "cross_rank_check": cross,
                "expected": (round(sum(r["local_crops_per_s"] for r in rank_step), 2) if all(r.get("local_crops_per_s") for r in rank_step) else None),
                "efficiency_vs_expected": (round(value / sum(r["local_crops_per_s"] for r in rank_step), 4)
                                           if all(r.get("local_crops_per_s") for r in rank_step) and not a.single_device else None),
                "expected_note": ("expected = sum over ranks of that rank's own shard timed ALONE on its GPU right after the timed region (per_rank[*]."
                                  "local_crops_per_s: no gather, no barrier); efficiency_vs_expected = value / expected = what the collectives and the "
                                  "barriers cost at this shard size; how the shard size itself costs is batch_sweep of the 1-GPU line"
                                  + ("; --single-device: the ranks time-slice ONE GPU in turns, value is serialised and the ratio is not formed" if a.single_device else "")),
                "note": ("bcast_ms: one broadcast of the packed weight arena (max over ranks, host clock, synchronised); "
                         "gather_ms_exposed: time the launch stream waited per step for the previous step's packed all-gather "
                         "(HIP events around the join; 0 = hidden under the next ViT); rank_step_ms: per-rank median step on the "
                         "launch stream")}
        if build_info:
            line["build"] = build_info
        if cpu_dry:
            line["dry_run"] = "fake engine on CPU tensors: orchestration only, `value` is meaningless"
        if a.single_device:
            line["dry_run"] = ("--single-device: real engines, all ranks time-slicing cuda:0 over gloo, forwards SERIALISED rank by rank between barriers inside "
                               "the timed region: orchestration + cross-rank check only, `value` is not concurrent throughput")
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
            line["gpu_over_cpu_b1"] = round(value / cpu["value_b1"], 1)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
