"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz from the REAL reference modules.

Run in the build container (needs /root/reference):   python oracle/gen_golden.py
For each case it
  1. builds seeded synthetic weights with tokenhmr_amd.weights (reference key names),
  2. instantiates the reference's own nn.Modules (imported in place by oracle/ref_import.py:
     ViT, TransformerDecoder, FCBlock, MixerLayer, PoseSPDecoderV1, QuantizeEMAReset,
     rot6d_to_rotmat, perspective_projection) and load_state_dict(strict=True)s the weights
     — which also proves the weight-name contract against the real modules,
  3. wires them exactly as token_head.py:86-128 / token_classifier.py:89-108 /
     tokenhmr.py:146-188 do (those three files cannot be imported: package-relative
     imports + yacs/pl), runs seeded crops, and
  4. freezes small slices of every stage boundary as float32 .npz.
The SMPL stage uses the oracle's restated LBS (smplx is absent; parity unpinned there).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tokenhmr_amd.config import HMRConfig  # noqa: E402
from tokenhmr_amd import weights as W  # noqa: E402
from tokenhmr_amd.smpl_assets import make_synthetic_smpl  # noqa: E402
from oracle import ref_import, tokenhmr_oracle as O  # noqa: E402

SAMPLE_TOKENS = [0, 5, 77, 100, 191]
VERT_STRIDE = 13


def make_inputs(B, seed=0):
    g = torch.Generator(device="cpu").manual_seed(4000 + seed)
    return torch.randn(B, 3, 256, 256, generator=g, dtype=torch.float32)


class RefHead(torch.nn.Module):
    """Attribute names mirror SMPLTokenDecoderHead / TokenClassfier so that the reference
    checkpoint keys load with strict=True."""

    def __init__(self, ns, cfg):
        super().__init__()
        nn = torch.nn
        self.transformer = ns.pose_transformer.TransformerDecoder(
            num_tokens=1, token_dim=1, dim=cfg.dec_dim, depth=cfg.dec_depth, heads=cfg.dec_heads,
            mlp_dim=cfg.dec_mlp, dim_head=cfg.dec_head_dim, dropout=0.0, emb_dropout=0.0,
            norm="layer", context_dim=cfg.dim)
        self.decpose_grot = nn.Linear(cfg.dec_dim, 6)
        self.decshape = nn.Linear(cfg.dec_dim, 10)
        self.deccam = nn.Linear(cfg.dec_dim, 3)
        self.decpose_hands = nn.Linear(cfg.dec_dim, 12)
        dp = nn.Module()
        dp.mixer_trans = ns.modules.FCBlock(cfg.dec_dim, cfg.token_num * cfg.mix_hidden)
        dp.mixer_head = nn.ModuleList([
            ns.modules.MixerLayer(cfg.mix_hidden, cfg.mix_hidden_inter, cfg.token_num, cfg.mix_token_inter, 0.0)
            for _ in range(cfg.mix_blocks)])
        dp.mixer_norm_layer = ns.modules.FCBlock(cfg.mix_hidden, cfg.mix_hidden)
        dp.class_pred_layer = nn.Linear(cfg.mix_hidden, cfg.token_classes)
        self.decpose = dp
        self.register_buffer("init_body_pose", torch.zeros(1, 144))
        self.register_buffer("init_betas", torch.zeros(1, 10))
        self.register_buffer("init_cam", torch.zeros(1, 3))


def build_reference(cfg, sd, tok):
    ns = ref_import.load()
    vit = ns.vit.ViT(img_size=(256, 192), patch_size=16, embed_dim=cfg.dim, depth=cfg.vit_depth,
                     num_heads=cfg.heads, ratio=1, use_checkpoint=False, mlp_ratio=4,
                     qkv_bias=True, drop_path_rate=0.55)
    vit.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    torch.nn.Module.train(vit, False)   # ViT.train() returns None (vit.py:345-348)
    head = RefHead(ns, cfg)
    head.load_state_dict({k[len("smpl_head."):]: v for k, v in sd.items() if k.startswith("smpl_head.")}, strict=True)
    head.eval()
    dec = ns.vqvae.PoseSPDecoderV1(rot_type="rot6d", output_dim=6, output_emb_width=cfg.code_dim, down_t=1,
                                   width=cfg.vq_width, depth=2, token_size_div=4, num_tokens=float(cfg.token_num),
                                   dilation_growth_rate=cfg.vq_dilation, num_joints=cfg.vq_joints,
                                   mesh_inference=False)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in tok.items() if k.startswith("decoder.")}, strict=True)
    dec.eval()
    quant = ns.quantize_cnn.QuantizeEMAReset(cfg.token_classes, cfg.code_dim)
    quant.load_state_dict({"codebook": tok["quantizer.codebook"]}, strict=True)
    quant.eval()
    return ns, vit, head, dec, quant


@torch.no_grad()
def reference_forward(img, cfg, sd, tok, smpl, dtype=torch.float32):
    """dtype = float64: the SAME reference modules evaluated in double precision (`.double()`): the value an fp32 implementation —
    the reference's own included — approximates.  Used to put a fixture's fp32 rounding noise next to this build's deviation."""
    ns, vit, head, dec, quant = build_reference(cfg, sd, tok)
    if dtype != torch.float32:
        for mod in (vit, head, dec, quant):
            mod.to(dtype)
        img = img.to(dtype)
        smpl = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in smpl.items()}
    B = img.shape[0]
    feats = vit(img)                                                        # tokenhmr.py:151
    x = feats.flatten(2).permute(0, 2, 1)                                   # token_head.py:69 (einops rearrange)
    token = torch.zeros(B, 1, 1, dtype=dtype)                               # token_head.py:91
    token_out = head.transformer(token, context=x).squeeze(1)               # :95-96
    grot = head.decpose_grot(token_out)                                     # :99
    dp = head.decpose                                                       # token_classifier.py:89-108
    cf = dp.mixer_trans(token_out).reshape(B, cfg.token_num, -1)
    for layer in dp.mixer_head:
        cf = layer(cf)
    cf = dp.mixer_norm_layer(cf)
    logits = dp.class_pred_layer(cf)
    probs = logits.softmax(-1)
    feat = quant.dequantize_logits(probs)                                   # vanilla_pose_vqvae.py:294-297
    bpose = dec(feat.permute(0, 2, 1))["pred_pose_body_6d"].reshape(B, -1)
    hands = head.decpose_hands(token_out)                                   # token_head.py:101
    pose6d = torch.cat([grot, bpose, hands], -1) + head.init_body_pose      # :103
    betas = head.decshape(token_out) + head.init_betas
    cam = head.deccam(token_out) + head.init_cam
    R = ns.geometry.rot6d_to_rotmat(pose6d).view(B, 24, 3, 3)               # :123
    focal = cfg.focal_length * torch.ones(B, 2, dtype=dtype)                # tokenhmr.py:165-169
    cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * focal[:, 0] / (cfg.img_size * cam[:, 0] + 1e-9)], dim=-1)
    verts, joints = O.smpl_forward(R[:, [0]], R[:, 1:], betas, smpl)        # restated smplx (unpinned)
    kp2d = ns.geometry.perspective_projection(joints, translation=cam_t, focal_length=focal / cfg.img_size)
    # argmin-L2 quantiser (S1 / K19) on the reference's own module, fed with the soft features
    q_in = feat.reshape(-1, cfg.code_dim)[: 64]
    q_idx = quant.quantize(q_in)
    return dict(vit_features=x, token_out=token_out, cls_logits=logits, probs=probs, pose6d=pose6d,
                betas=betas, cam=cam, rotmat=R, cam_t=cam_t, verts=verts, joints=joints, kp2d=kp2d,
                vq_in=q_in, vq_idx=q_idx)


def freeze(ref, cfg):
    logits = ref["cls_logits"]
    top2 = logits.topk(2, dim=-1).values
    out = {
        "vit_features_sample": ref["vit_features"][:, SAMPLE_TOKENS, :],
        "token_out": ref["token_out"],
        "logits_sample": logits[:, ::16, :][:, :, ::8],
        "token_idx": O.token_indices(logits),
        "top2_gap": top2[..., 0] - top2[..., 1],
        "probs_max": ref["probs"].max(-1).values,
        "pose6d": ref["pose6d"], "betas": ref["betas"], "cam": ref["cam"], "rotmat": ref["rotmat"],
        "cam_t": ref["cam_t"], "verts_sample": ref["verts"][:, ::VERT_STRIDE], "joints": ref["joints"],
        "kp2d": ref["kp2d"], "vq_in": ref["vq_in"], "vq_idx": ref["vq_idx"].to(torch.int32),
    }
    return {k: v.numpy() for k, v in out.items()}


def freeze_compact(ref, cfg):
    """B = 64 at full depth: only what the token-index / joints claim needs (~0.3 MB): every token index with its top-2
    logit gap (64 x 160 = 10,240 tokens), the top-1 probability, joints, camera, rotations, betas, projected keypoints and
    every 53rd vertex."""
    logits = ref["cls_logits"]
    top2 = logits.topk(2, dim=-1).values
    out = {
        "token_idx": O.token_indices(logits).to(torch.int32),
        "top2_gap": top2[..., 0] - top2[..., 1],
        "probs_max": ref["probs"].max(-1).values,
        "token_out_sample": ref["token_out"][:, ::16],
        "pose6d": ref["pose6d"], "betas": ref["betas"], "cam": ref["cam"], "rotmat": ref["rotmat"],
        "cam_t": ref["cam_t"], "verts_sample": ref["verts"][:, ::VERT_STRIDE_B64], "joints": ref["joints"],
        "kp2d": ref["kp2d"],
    }
    return {k: v.numpy() for k, v in out.items()}


VERT_STRIDE_B64 = 53

CASES = {
    # name: (vit_depth, dec_depth, batch, seed[, weight style])
    "small_d2": (2, 2, 2, 0),
    "full_d32": (32, 6, 2, 0),
    # BASELINE.json configs[2] at its own size: 64 distinct crops = 10,240 pose tokens through the reference's modules.
    # The crops are make_inputs(64, 0) == bench.py's rank-0 batch, so the bench line can report parity on its own input.
    "full_d32_b64": (32, 6, 64, 0),
    # the same size on two more weight / crop seeds, and on a "trained-like" state (weights.make_synthetic_state(style="trained"):
    # LayerNorm gains in [0.1, 10], x50 outlier channels in proj / fc2, non-trivial mean parameters): 4 x 10,240 = 40,960 tokens
    "full_d32_b64_s1": (32, 6, 64, 1),
    "full_d32_b64_s2": (32, 6, 64, 2),
    "full_d32_b64_trained": (32, 6, 64, 3, "trained"),
    "small_d2_trained": (2, 2, 2, 3, "trained"),
}


def main():
    torch.set_num_threads(os.cpu_count())
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    for name in names:
        vd, dd, B, seed = CASES[name][:4]
        style = CASES[name][4] if len(CASES[name]) > 4 else "init"
        cfg = HMRConfig(vit_depth=vd, dec_depth=dd)
        sd = W.make_synthetic_state(cfg, seed, style)
        tok = W.make_synthetic_tokenizer(cfg, seed)
        smpl = make_synthetic_smpl(cfg, seed)
        img = make_inputs(B, seed)
        ref = reference_forward(img, cfg, sd, tok, smpl)
        g = freeze_compact(ref, cfg) if "_b64" in name else freeze(ref, cfg)
        if "_b64" in name:
            # the same modules in float64: how far the reference's OWN fp32 result is from the value it approximates, per fixture
            r64 = reference_forward(img, cfg, sd, tok, smpl, dtype=torch.float64)
            g["joints_f64"] = r64["joints"].numpy()
            g["verts_sample_f64"] = r64["verts"][:, ::VERT_STRIDE_B64].numpy()
            g["token_idx_f64"] = O.token_indices(r64["cls_logits"]).to(torch.int32).numpy()
            g["ref32_vs_f64"] = np.array([float((ref["joints"].double() - r64["joints"]).abs().max()),
                                          float((ref["verts"].double() - r64["verts"]).abs().max()),
                                          float((O.token_indices(ref["cls_logits"]) != O.token_indices(r64["cls_logits"])).sum())])
            print(f"[{name}] reference fp32 vs the same modules in fp64: joints {g['ref32_vs_f64'][0]:.3e} m, vertices {g['ref32_vs_f64'][1]:.3e} m, "
                  f"token indices that differ {int(g['ref32_vs_f64'][2])} of {B * cfg.token_num}")
            del r64
        g["meta"] = np.array([vd, dd, B, seed], dtype=np.int64)
        g["style"] = np.array(style)
        gap = torch.from_numpy(g["top2_gap"])
        print(f"[{name}] logits |max| {ref['cls_logits'].abs().max():.3f}  top-2 gaps: min {gap.min():.3e}, "
              f"< 1e-4: {int((gap < 1e-4).sum())}, < 1e-3: {int((gap < 1e-3).sum())}, < 1e-2: {int((gap < 1e-2).sum())} of {gap.numel()};  "
              f"distinct token ids {len(np.unique(g['token_idx']))};  vit feature |max| {ref['vit_features'].abs().max():.2f}")
        g["weights_checksum"] = np.array([W.checksum(sd), W.checksum(tok)], dtype=np.float64)
        g["img_checksum"] = np.array([float(img.double().sum()), float(img[:, :, ::7, ::5].double().abs().sum())])
        # oracle vs live reference, reported at generation time
        with torch.no_grad():
            orc = O.forward(img, sd, tok, smpl, cfg)
        g["oracle_vs_reference_maxdiff"] = np.array([(ref[a] - orc[b]).abs().max().item() for a, b in
                                                     [("vit_features", "vit_features"), ("cls_logits", "cls_logits"),
                                                      ("verts", "pred_vertices"), ("joints", "pred_keypoints_3d")]])
        g["oracle_idx_mismatches"] = np.array([int((O.token_indices(ref["cls_logits"]) != orc["token_idx"]).sum())])
        for k_ref, k_or in [("vit_features", "vit_features"), ("token_out", "token_out"), ("cls_logits", "cls_logits"),
                            ("pose6d", "pose6d"), ("verts", "pred_vertices"), ("joints", "pred_keypoints_3d"),
                            ("kp2d", "pred_keypoints_2d")]:
            d = (ref[k_ref] - orc[k_or]).abs().max().item()
            print(f"[{name}] oracle vs reference  {k_ref:14s} max|diff| = {d:.3e}")
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **g)
        print(f"[{name}] wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


if __name__ == "__main__":
    main()
