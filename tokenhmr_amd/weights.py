"""Weight-name contract of the TokenHMR inference path + seeded synthetic weights.

The names and shapes are the reference's checkpoint contract:
  * TokenHMR Lightning ckpt  ['state_dict'] -> 'backbone.*', 'smpl_head.*'
    (tokenhmr/lib/utils/misc.py:242-256, :215-240)
  * tokenizer.pth            ['net'] -> 'decoder.decoder.<i>...', 'quantizer.codebook'
    (tokenization/models/vanilla_pose_vqvae.py:24-40, :299-301)
`spec()` is the single source of truth for both the synthetic generator used by
tests/bench (no checkpoints exist offline) and the engine's weight packer.

Synthetic init (there is no network for checkpoints): PyTorch-default-like
U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear/Conv, LayerNorm gamma 1+0.1*randn and
beta 0.05*randn (so gamma/beta handling is actually exercised), randn codebook
(the reference's zero-initialised codebook, quantize_cnn.py:18, would collapse every
pose), 0.02*randn ViT pos_embed (vit.py:254-255), randn decoder pos_embedding
(pose_transformer.py:329).  Deterministic for a given (seed, torch version).
"""
from collections import OrderedDict
import math

import torch

from .config import HMRConfig, RELEASE


def spec(cfg: HMRConfig = RELEASE):
    """Ordered list of (name, shape, kind, fan_in) for every tensor the hot path reads."""
    S = []
    D, H = cfg.dim, cfg.mlp_dim

    def lin(name, out_f, in_f, bias=True):
        S.append((name + ".weight", (out_f, in_f), "w", in_f))
        if bias:
            S.append((name + ".bias", (out_f,), "b", in_f))

    def ln(name, d):
        S.append((name + ".weight", (d,), "ln_w", 0))
        S.append((name + ".bias", (d,), "ln_b", 0))

    # ---- backbone (vit.py) ----
    S.append(("backbone.pos_embed", (1, cfg.tokens + 1, D), "pos", 0))
    S.append(("backbone.patch_embed.proj.weight", (D, 3, cfg.patch, cfg.patch), "w", 3 * cfg.patch * cfg.patch))
    S.append(("backbone.patch_embed.proj.bias", (D,), "b", 3 * cfg.patch * cfg.patch))
    for i in range(cfg.vit_depth):
        p = f"backbone.blocks.{i}."
        ln(p + "norm1", D)
        lin(p + "attn.qkv", 3 * D, D)
        lin(p + "attn.proj", D, D)
        ln(p + "norm2", D)
        lin(p + "mlp.fc1", H, D)
        lin(p + "mlp.fc2", D, H)
    ln("backbone.last_norm", D)

    # ---- smpl_head.transformer (pose_transformer.py) ----
    T = "smpl_head.transformer."
    E, I = cfg.dec_dim, cfg.inner
    S.append((T + "pos_embedding", (1, 1, E), "randn", 0))
    lin(T + "to_token_embedding", E, 1)
    for l in range(cfg.dec_depth):
        p = T + f"transformer.layers.{l}."
        ln(p + "0.norm", E)
        lin(p + "0.fn.to_qkv", 3 * I, E, bias=False)
        lin(p + "0.fn.to_out.0", E, I)
        ln(p + "1.norm", E)
        lin(p + "1.fn.to_kv", 2 * I, cfg.dim, bias=False)
        lin(p + "1.fn.to_q", I, E, bias=False)
        lin(p + "1.fn.to_out.0", E, I)
        ln(p + "2.norm", E)
        lin(p + "2.fn.net.0", cfg.dec_mlp, E)
        lin(p + "2.fn.net.3", E, cfg.dec_mlp)
    # ---- read-outs (token_head.py:40-43) ----
    lin("smpl_head.decpose_grot", 6, E)
    lin("smpl_head.decshape", 10, E)
    lin("smpl_head.deccam", 3, E)
    lin("smpl_head.decpose_hands", 12, E)
    # ---- token classifier (token_classifier.py:71-82, modules.py) ----
    C = "smpl_head.decpose."
    Tn, Hd = cfg.token_num, cfg.mix_hidden
    lin(C + "mixer_trans.ff.0", Tn * Hd, E)
    ln(C + "mixer_trans.ff.1", Tn * Hd)
    for m in range(cfg.mix_blocks):
        p = C + f"mixer_head.{m}."
        ln(p + "layernorm1", Hd)
        lin(p + "MLP_token.ff.0", cfg.mix_token_inter, Tn)
        lin(p + "MLP_token.ff.3", Tn, cfg.mix_token_inter)
        ln(p + "layernorm2", Hd)
        lin(p + "MLP_channel.ff.0", cfg.mix_hidden_inter, Hd)
        lin(p + "MLP_channel.ff.3", Hd, cfg.mix_hidden_inter)
    lin(C + "mixer_norm_layer.ff.0", Hd, Hd)
    ln(C + "mixer_norm_layer.ff.1", Hd)
    lin(C + "class_pred_layer", cfg.token_classes, Hd)
    # ---- mean params (token_head.py:55-60) ----
    S.append(("smpl_head.init_body_pose", (1, 144), "mean_pose", 0))
    S.append(("smpl_head.init_betas", (1, 10), "zeros", 0))
    S.append(("smpl_head.init_cam", (1, 3), "mean_cam", 0))
    return S


def tokenizer_spec(cfg: HMRConfig = RELEASE):
    """tokenizer.pth ['net'] entries the decode path reads (vanilla_pose_vqvae.py:135-154)."""
    W, C = cfg.vq_width, cfg.code_dim
    S = []

    def conv(name, co, ci, k):
        S.append((name + ".weight", (co, ci, k), "w", ci * k))
        S.append((name + ".bias", (co,), "b", ci * k))

    conv("decoder.decoder.0", W, C, 3)
    for i in (3, 6, 9, 12):
        conv(f"decoder.decoder.{i}", W, W, 3)
    for blk in (0, 1):
        conv(f"decoder.decoder.14.0.model.{blk}.conv1", W, W, 3)
        conv(f"decoder.decoder.14.0.model.{blk}.conv2", W, W, 1)
    conv("decoder.decoder.14.1", W, W, 3)
    conv("decoder.decoder.15", 6, W, 3)
    S.append(("quantizer.codebook", (cfg.token_classes, C), "randn", 0))
    return S


def tokenizer_encoder_spec(cfg: HMRConfig = RELEASE):
    """OPTIONAL tokenizer.pth ['net'] 'encoder.encoder.*' entries (PoseSPEncoderV1, vanilla_pose_vqvae.py:66-88 with the
    release ARCH: input_dim 6, token_size_mul 4, down_t 1) — only needed for the encode path (EncodeTokens)."""
    W, C = cfg.vq_width, cfg.code_dim
    S = []

    def conv(name, co, ci, k):
        S.append((name + ".weight", (co, ci, k), "w", ci * k))
        S.append((name + ".bias", (co,), "b", ci * k))

    conv("encoder.encoder.0", W, 6, 3)
    for i in (3, 6, 9, 12):
        conv(f"encoder.encoder.{i}", W, W, 3)
    conv("encoder.encoder.14.0", W, W, 4)
    for blk in (0, 1):
        conv(f"encoder.encoder.14.1.model.{blk}.conv1", W, W, 3)
        conv(f"encoder.encoder.14.1.model.{blk}.conv2", W, W, 1)
    conv("encoder.encoder.15", C, W, 3)
    return S


def make_synthetic_encoder(cfg: HMRConfig = RELEASE, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(2500 + seed)
    sd = OrderedDict()
    for name, shape, kind, fan_in in tokenizer_encoder_spec(cfg):
        sd[name] = _fill(shape, kind, fan_in, g)
    return sd


def _fill(shape, kind, fan_in, g):
    if kind in ("w", "b"):
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
    if kind == "ln_w":
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "ln_b":
        return 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "pos":
        return 0.02 * torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "randn":
        return torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "zeros":
        return torch.zeros(shape, dtype=torch.float32)
    if kind == "mean_pose":   # identity rotation in 6D for 24 joints
        return torch.tensor([1.0, 0, 0, 0, 1, 0], dtype=torch.float32).repeat(24).reshape(shape)
    if kind == "mean_cam":
        return torch.tensor([0.9, 0.0, 0.0], dtype=torch.float32).reshape(shape)
    raise ValueError(kind)


def make_synthetic_state(cfg: HMRConfig = RELEASE, seed: int = 0, style: str = "init"):
    """Seeded TokenHMR state_dict (reference key names).

    style "init": the default-init statistics described in the module docstring.
    style "trained": the same draw, then reshaped towards what a TRAINED ViT-H looks like (the released weights are trained,
    README.md:92-97, and cannot be fetched here): per-channel LayerNorm gains log-uniform in [0.1, 10] with biases of a few
    tenths, ~1 % outlier output channels scaled x50 in every `attn.proj` / `mlp.fc2` weight (the "massive activation" channels
    of the residual stream), and non-trivial mean parameters (a random pose of ~0.3 rad per joint, betas, a shifted camera).
    The "init" tensors of a (cfg, seed) are bit-for-bit the same whatever styles exist (own generator for the reshaping)."""
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    sd = OrderedDict()
    for name, shape, kind, fan_in in spec(cfg):
        sd[name] = _fill(shape, kind, fan_in, g)
    if style == "init":
        return sd
    if style != "trained":
        raise ValueError(f"unknown weight style '{style}'")
    g2 = torch.Generator(device="cpu").manual_seed(7000 + seed)
    for name, shape, kind, fan_in in spec(cfg):
        if name.endswith(("mixer_trans.ff.1.weight", "mixer_trans.ff.1.bias")):
            continue                                  # LayerNorm(10240) in front of the classifier: left at its init statistics
        if kind == "ln_w":
            lo, hi = math.log(0.1), math.log(10.0)
            sd[name] = torch.exp(lo + (hi - lo) * torch.rand(shape, generator=g2, dtype=torch.float32))
        elif kind == "ln_b":
            sd[name] = 0.3 * torch.randn(shape, generator=g2, dtype=torch.float32)
        elif kind == "w" and name.startswith("backbone.blocks.") and name.endswith(("attn.proj.weight", "mlp.fc2.weight")):
            n_out = shape[0]
            pick = torch.randperm(n_out, generator=g2)[: max(1, n_out // 100)]
            w = sd[name].clone()
            w[pick] *= 50.0
            sd[name] = w
    # mean parameters: 24 random rotations of ~0.3 rad in the reference's 6D form (the first two ROWS of R: geometry.py:73-84
    # reads elements 0-2 as a1, 3-5 as a2 and stacks b1, b2, b3 as rows), betas, camera
    aa = 0.3 * torch.randn(24, 3, generator=g2, dtype=torch.float64)
    th = aa.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = aa / th
    K = torch.zeros(24, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    R = torch.eye(3, dtype=torch.float64) + torch.sin(th)[..., None] * K + (1 - torch.cos(th))[..., None] * (K @ K)
    sd["smpl_head.init_body_pose"] = R[:, :2, :].reshape(1, 144).float()
    sd["smpl_head.init_betas"] = 0.5 * torch.randn(1, 10, generator=g2, dtype=torch.float32)
    sd["smpl_head.init_cam"] = torch.tensor([[0.85, 0.06, -0.04]], dtype=torch.float32)
    return sd


def make_synthetic_tokenizer(cfg: HMRConfig = RELEASE, seed: int = 0):
    """Seeded tokenizer 'net' dict (decoder + codebook)."""
    g = torch.Generator(device="cpu").manual_seed(2000 + seed)
    sd = OrderedDict()
    for name, shape, kind, fan_in in tokenizer_spec(cfg):
        sd[name] = _fill(shape, kind, fan_in, g)
    return sd


def checksum(sd) -> float:
    """Cheap fingerprint so golden fixtures can detect a weight-generator drift."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        flat = v.reshape(-1).double()
        n = min(flat.numel(), 4096)
        acc += float(flat[:n].sum()) * (1.0 + (i % 7))
    return acc


def validate_state(sd, cfg: HMRConfig = RELEASE, tokenizer=None):
    """Raise KeyError/ValueError if a required tensor is missing or mis-shaped
    (same failure class as load_state_dict(strict=True), misc.py:229)."""
    for name, shape, _, _ in spec(cfg):
        if name not in sd:
            raise KeyError(f"missing tensor '{name}' in TokenHMR state_dict")
        if tuple(sd[name].shape) != tuple(shape):
            raise ValueError(f"'{name}': expected {tuple(shape)}, got {tuple(sd[name].shape)}")
    if tokenizer is not None:
        for name, shape, _, _ in tokenizer_spec(cfg):
            if name not in tokenizer:
                raise KeyError(f"missing tensor '{name}' in tokenizer net")
            if tuple(tokenizer[name].shape) != tuple(shape):
                raise ValueError(f"'{name}': expected {tuple(shape)}, got {tuple(tokenizer[name].shape)}")
