"""BUILD CONTAINER ONLY (needs /root/reference): the reference's OWN modules (ViT, TransformerDecoder, FCBlock, MixerLayer, PoseSPDecoderV1,
QuantizeEMAReset — imported in place by oracle/ref_import.py, wired by oracle/gen_golden.py) timed beside the oracle port
(oracle/tokenhmr_oracle.py) on the same host cores, same inputs, same thread count.

Why: bench.py's `cpu_baseline` is `kind: "port"` — the reference is Python, and a Python reference cannot travel to the
GPU box in any form (source or bytecode), so the box's CPU leg can only time the port.  What CAN be measured is how far the port's rate is
from the reference's own on one machine: this script, on the build container's 8 cores -> profiles/r6_cpu_reference_vs_port.json.  Both are
the same torch CPU operators in the same order (the oracle is pinned bit-exact to these modules, tests/test_oracle_golden.py), so the ratio
is expected at 1.0 — and then the GPU box's port number stands for the reference's on that box.

    python scripts/cpu_reference_vs_port.py [threads=8] [out.json]
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import gen_golden as G, ref_import, tokenhmr_oracle as O  # noqa: E402
from tokenhmr_amd.config import RELEASE as cfg  # noqa: E402
from tokenhmr_amd import weights as W  # noqa: E402
from tokenhmr_amd.smpl_assets import make_synthetic_smpl  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
out = sys.argv[2] if len(sys.argv) > 2 else None
if not ref_import.available():
    sys.exit("needs /root/reference (build container)")
torch.set_num_threads(threads)
sd, tok, smpl = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
img = G.make_inputs(8, 0)
ns, vit, head, dec, quant = G.build_reference(cfg, sd, tok)


@torch.no_grad()
def reference(x):
    """tokenhmr.py:146-188 on the reference's modules (the wiring of oracle/gen_golden.reference_forward, modules built once)."""
    B = x.shape[0]
    feats = vit(x)
    c = feats.flatten(2).permute(0, 2, 1)
    token_out = head.transformer(torch.zeros(B, 1, 1), context=c).squeeze(1)
    dp = head.decpose
    cf = dp.mixer_trans(token_out).reshape(B, cfg.token_num, -1)
    for layer in dp.mixer_head:
        cf = layer(cf)
    probs = dp.class_pred_layer(dp.mixer_norm_layer(cf)).softmax(-1)
    bpose = dec(quant.dequantize_logits(probs).permute(0, 2, 1))["pred_pose_body_6d"].reshape(B, -1)
    pose6d = torch.cat([head.decpose_grot(token_out), bpose, head.decpose_hands(token_out)], -1) + head.init_body_pose
    betas = head.decshape(token_out) + head.init_betas
    cam = head.deccam(token_out) + head.init_cam
    R = ns.geometry.rot6d_to_rotmat(pose6d).view(B, 24, 3, 3)
    focal = cfg.focal_length * torch.ones(B, 2)
    cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * focal[:, 0] / (cfg.img_size * cam[:, 0] + 1e-9)], dim=-1)
    verts, joints = O.smpl_forward(R[:, [0]], R[:, 1:], betas, smpl)          # smplx is absent: the restated LBS on both sides (0.006 % of the flops)
    return verts, ns.geometry.perspective_projection(joints, translation=cam_t, focal_length=focal / cfg.img_size)


@torch.no_grad()
def port(x):
    o = O.forward(x, sd, tok, smpl, cfg)
    return o["pred_vertices"], o["pred_keypoints_2d"]


def timed(fn, x, n):
    fn(x)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn(x)
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


res = {"host_cpus": os.cpu_count(), "threads": threads, "torch": torch.__version__,
       "what": ("the reference's own modules (imported in place from /root/reference) vs the oracle port, full path, fp32, same crops, same "
                "torch.set_num_threads; median of the timed passes, interleaved")}
v_ref, _ = reference(img[:2])
v_port, _ = port(img[:2])
res["outputs_bit_identical"] = bool(torch.equal(v_ref, v_port))
for B, n in ((8, 3), (1, 5)):
    x = img[:B]
    a, ta = timed(reference, x, n)
    b, tb = timed(port, x, n)
    a2, ta2 = timed(reference, x, n)
    med_ref = statistics.median(ta + ta2)
    res[f"b{B}"] = {"reference_crops_s": round(B / med_ref, 3), "port_crops_s": round(B / b, 3), "port_over_reference": round(med_ref / b, 4),
                    "reference_s_per_pass": [round(t, 3) for t in ta + ta2], "port_s_per_pass": [round(t, 3) for t in tb]}
    res[f"b{B}"]["reference_crops_s_per_thread"] = round(B / med_ref / threads, 4)
print(json.dumps(res))
if out:
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
