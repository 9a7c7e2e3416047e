"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the TokenHMR inference hot path.

This file is the *oracle* the HIP path is checked against.  It is imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product path
(tokenhmr_amd/) never imports it and has no CPU fallback.

Every function restates one reference function with plain torch CPU ops and cites
the reference file:line it follows (paths relative to /root/reference).  The oracle
is pinned two ways (tests/test_oracle_golden.py):
  1. against golden tensors produced by the reference's *own* nn.Modules run in the
     build container (oracle/gen_golden.py -> tests/golden/*.npz), and
  2. live against those modules whenever /root/reference is present.
SMPL LBS is the exception: its arithmetic lives in the un-vendored third-party
package smplx==0.1.28 (tokenhmr/requirements.txt:3) which is absent offline, and
the reference repo holds no golden vectors for it -> the LBS stage is a restatement
of smplx's published algorithm (lbs.py::lbs, body_models.py::SMPLLayer.forward,
vertex_joint_selector.py) and its parity is **unpinned** at that boundary.
"""
import math

import torch
import torch.nn.functional as F

from tokenhmr_amd.config import HMRConfig, RELEASE


# --------------------------------------------------------------------------- ViT-H
def patch_embed(img, sd, cfg: HMRConfig = RELEASE):
    """vit.py:341-343 (column crop 32:-32), :170-176 PatchEmbed (Conv2d k16 s16 p2),
    :327 (+pos_embed[:,1:] + pos_embed[:,:1])."""
    x = img[:, :, :, cfg.crop_x0:cfg.img_size - cfg.crop_x0]
    x = F.conv2d(x, sd["backbone.patch_embed.proj.weight"], sd["backbone.patch_embed.proj.bias"],
                 stride=cfg.patch, padding=cfg.patch_pad)
    x = x.flatten(2).transpose(1, 2)
    pos = sd["backbone.pos_embed"]
    return x + pos[:, 1:] + pos[:, :1]


def vit_attention(x, sd, i, cfg: HMRConfig = RELEASE):
    """vit.py:110-126 Attention.forward (q scaled *before* q@k^T, fp32 softmax)."""
    B, N, C = x.shape
    p = f"backbone.blocks.{i}.attn."
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, cfg.heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (cfg.head_dim ** -0.5)
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def vit_block(x, sd, i, cfg: HMRConfig = RELEASE):
    """vit.py:148-151 Block.forward; Mlp :82-87 (exact-erf GELU); LN eps 1e-6 (:222)."""
    p = f"backbone.blocks.{i}."
    D = cfg.dim
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.vit_ln_eps)
    x = x + vit_attention(h, sd, i, cfg)
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.vit_ln_eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def vit_forward(img, sd, cfg: HMRConfig = RELEASE, taps=None):
    """vit.py:320-343.  Returns the final features token-major (B,192,1280); the
    reference's (B,1280,16,12) permute (:337) is undone by token_head.py:69."""
    x = patch_embed(img, sd, cfg)
    if taps is not None:
        taps["patch"] = x.clone()
    for i in range(cfg.vit_depth):
        x = vit_block(x, sd, i, cfg)
        if taps is not None and i in (0, cfg.vit_depth - 1):
            taps[f"block{i}"] = x.clone()
    return F.layer_norm(x, (cfg.dim,), sd["backbone.last_norm.weight"], sd["backbone.last_norm.bias"], cfg.vit_ln_eps)


# --------------------------------------------------------------------------- decoder
def decoder_forward(ctx, sd, cfg: HMRConfig = RELEASE):
    """pose_transformer.py:349-357 TransformerDecoder.forward with the zero input token of
    token_head.py:91, :191-201 TransformerCrossAttn, :75-86 Attention, :111-124
    CrossAttention (context NOT normalised, scale applied after the dot), :40-52 FF."""
    B = ctx.shape[0]
    T = "smpl_head.transformer."
    E, Hh, dh = cfg.dec_dim, cfg.dec_heads, cfg.dec_head_dim
    token = torch.zeros(B, 1, 1, dtype=ctx.dtype)
    x = F.linear(token, sd[T + "to_token_embedding.weight"], sd[T + "to_token_embedding.bias"])
    x = x + sd[T + "pos_embedding"][:, :1]
    scale = dh ** -0.5
    for l in range(cfg.dec_depth):
        p = T + f"transformer.layers.{l}."
        # self-attention over ONE token
        h = F.layer_norm(x, (E,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"], cfg.ln_eps)
        q, k, v = F.linear(h, sd[p + "0.fn.to_qkv.weight"]).chunk(3, dim=-1)
        q, k, v = [t.reshape(B, 1, Hh, dh).transpose(1, 2) for t in (q, k, v)]
        a = (torch.matmul(q, k.transpose(-1, -2)) * scale).softmax(dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, 1, Hh * dh)
        x = F.linear(o, sd[p + "0.fn.to_out.0.weight"], sd[p + "0.fn.to_out.0.bias"]) + x
        # cross-attention
        h = F.layer_norm(x, (E,), sd[p + "1.norm.weight"], sd[p + "1.norm.bias"], cfg.ln_eps)
        k, v = F.linear(ctx, sd[p + "1.fn.to_kv.weight"]).chunk(2, dim=-1)
        q = F.linear(h, sd[p + "1.fn.to_q.weight"])
        q = q.reshape(B, 1, Hh, dh).transpose(1, 2)
        k = k.reshape(B, -1, Hh, dh).transpose(1, 2)
        v = v.reshape(B, -1, Hh, dh).transpose(1, 2)
        a = (torch.matmul(q, k.transpose(-1, -2)) * scale).softmax(dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, 1, Hh * dh)
        x = F.linear(o, sd[p + "1.fn.to_out.0.weight"], sd[p + "1.fn.to_out.0.bias"]) + x
        # feed-forward
        h = F.layer_norm(x, (E,), sd[p + "2.norm.weight"], sd[p + "2.norm.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(h, sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"]))
        x = F.linear(h, sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"]) + x
    return x.squeeze(1)


# --------------------------------------------------------------------------- classifier
def _fcblock(x, sd, p, eps):
    """heads/modules.py:11-22 FCBlock = Linear -> LayerNorm -> ReLU."""
    y = F.linear(x, sd[p + "ff.0.weight"], sd[p + "ff.0.bias"])
    y = F.layer_norm(y, (y.shape[-1],), sd[p + "ff.1.weight"], sd[p + "ff.1.bias"], eps)
    return F.relu(y)


def _mlpblock(x, sd, p):
    """heads/modules.py:25-38 MLPBlock = Linear -> GELU -> Linear."""
    return F.linear(F.gelu(F.linear(x, sd[p + "ff.0.weight"], sd[p + "ff.0.bias"])),
                    sd[p + "ff.3.weight"], sd[p + "ff.3.bias"])


def classifier_logits(token_out, sd, cfg: HMRConfig = RELEASE):
    """token_classifier.py:89-101: mixer_trans -> 4x MixerLayer (modules.py:55-63)
    -> mixer_norm_layer -> class_pred_layer.  Returns raw logits (B,160,2048)."""
    C = "smpl_head.decpose."
    B = token_out.shape[0]
    x = _fcblock(token_out, sd, C + "mixer_trans.", cfg.ln_eps).reshape(B, cfg.token_num, -1)
    Hd = cfg.mix_hidden
    for m in range(cfg.mix_blocks):
        p = C + f"mixer_head.{m}."
        y = F.layer_norm(x, (Hd,), sd[p + "layernorm1.weight"], sd[p + "layernorm1.bias"], cfg.ln_eps)
        y = _mlpblock(y.transpose(2, 1), sd, p + "MLP_token.").transpose(2, 1)
        z = F.layer_norm(x + y, (Hd,), sd[p + "layernorm2.weight"], sd[p + "layernorm2.bias"], cfg.ln_eps)
        z = _mlpblock(z, sd, p + "MLP_channel.")
        x = x + y + z
    x = _fcblock(x, sd, C + "mixer_norm_layer.", cfg.ln_eps)
    return F.linear(x, sd[C + "class_pred_layer.weight"], sd[C + "class_pred_layer.bias"])


def token_indices(logits):
    """Build-defined 'pose-token indices' (SURVEY.md S1): argmax_k logits[b,t,k], lowest index
    on ties.  The reference never materialises indices at inference."""
    m = logits.max(dim=-1, keepdim=True).values
    K = logits.shape[-1]
    ar = torch.arange(K).expand_as(logits)
    return torch.where(logits == m, ar, torch.full_like(ar, K)).min(dim=-1).values.to(torch.int32)


# --------------------------------------------------------------------------- VQ-VAE decode
def nearest_index(t_in: int, t_out: int):
    """torch nn.Upsample(size) 'nearest' source index: min(floor(dst * (float)in/out), in-1)
    evaluated in fp32 (ATen upsample.h nearest_neighbor_compute_source_index)."""
    scale = torch.tensor(t_in, dtype=torch.float32) / torch.tensor(t_out, dtype=torch.float32)
    dst = torch.arange(t_out, dtype=torch.float32)
    return torch.clamp(torch.floor(dst * scale).to(torch.int64), max=t_in - 1)


def vq_decode(softmax_probs, tok, cfg: HMRConfig = RELEASE):
    """vanilla_pose_vqvae.py:294-297 DecodeTokens.forward: quantize_cnn.py:92-93
    dequantize_logits (probs @ codebook) -> permute -> :135-154 decoder stack
    (resnet.py:49-69 ResConv1DBlock, relu, norm=None) -> :156-159 (B,21,6)."""
    feat = torch.matmul(softmax_probs, tok["quantizer.codebook"])     # (B,160,256)
    x = feat.permute(0, 2, 1)                                          # (B,256,160)
    d = "decoder.decoder."
    x = F.relu(F.conv1d(x, tok[d + "0.weight"], tok[d + "0.bias"], padding=1))
    lens = cfg.vq_lengths
    for li, t_out in zip((3, 6, 9, 12), lens[1:]):
        idx = nearest_index(x.shape[-1], t_out)
        x = x[:, :, idx]
        x = F.relu(F.conv1d(x, tok[d + f"{li}.weight"], tok[d + f"{li}.bias"], padding=1))
    # Resnet1D(depth 2, dilation_growth 3, reverse_dilation) -> dilations 3 then 1 (resnet.py:75-77)
    for blk, dil in ((0, cfg.vq_dilation), (1, 1)):
        p = d + f"14.0.model.{blk}."
        h = F.relu(x)
        h = F.conv1d(h, tok[p + "conv1.weight"], tok[p + "conv1.bias"], padding=dil, dilation=dil)
        h = F.relu(h)
        h = F.conv1d(h, tok[p + "conv2.weight"], tok[p + "conv2.bias"])
        x = h + x
    x = F.conv1d(x, tok[d + "14.1.weight"], tok[d + "14.1.bias"], padding=1)
    x = F.conv1d(x, tok[d + "15.weight"], tok[d + "15.bias"], padding=1)
    return x.permute(0, 2, 1)                                           # (B,21,6)


def vq_encode(pose6d, enc, codebook):
    """EncodeTokens.forward (vanilla_pose_vqvae.py:334-342): PoseSPEncoderV1.forward (:90-111, layers :66-88 with the
    release ARCH) -> QuantizeEMAReset.preprocess (quantize_cnn.py:74-78) -> quantize (:80-86).
    pose6d (B,21,6) -> (idx (B*160,), latent (B*160,256), dist)."""
    e = "encoder.encoder."
    x = pose6d.reshape(pose6d.shape[0], pose6d.shape[1], -1).permute(0, 2, 1)          # preprocess, :90-94
    x = F.relu(F.conv1d(x, enc[e + "0.weight"], enc[e + "0.bias"], padding=1))
    x = x[:, :, nearest_index(21, 40)]                                                  # nn.Upsample(size=40)
    x = F.relu(F.conv1d(x, enc[e + "3.weight"], enc[e + "3.bias"], padding=1))
    for li in (6, 9, 12):                                                               # nn.Upsample(scale_factor=2)
        t = x.shape[-1]
        x = x[:, :, torch.arange(2 * t) // 2]
        x = F.relu(F.conv1d(x, enc[e + f"{li}.weight"], enc[e + f"{li}.bias"], padding=1))
    x = F.conv1d(x, enc[e + "14.0.weight"], enc[e + "14.0.bias"], stride=2, padding=1)
    for blk, dil in ((0, 3), (1, 1)):                                                   # Resnet1D reverse_dilation
        p = e + f"14.1.model.{blk}."
        h = F.conv1d(F.relu(x), enc[p + "conv1.weight"], enc[p + "conv1.bias"], padding=dil, dilation=dil)
        x = F.conv1d(F.relu(h), enc[p + "conv2.weight"], enc[p + "conv2.bias"]) + x
    x = F.conv1d(x, enc[e + "15.weight"], enc[e + "15.bias"], padding=1)                # (B,256,160)
    lat = x.permute(0, 2, 1).contiguous().view(-1, x.shape[1])                          # quantize_cnn.py:74-78
    idx, dist = vq_quantize(lat, codebook)
    return idx, lat, dist


def vq_quantize(x, codebook):
    """quantize_cnn.py:80-86 QuantizeEMAReset.quantize: argmin_k of the *expanded* distance
    sum(x^2) - 2 x.C^T + sum(C^2), evaluated in that order in fp32."""
    k_w = codebook.t()
    dist = torch.sum(x ** 2, dim=-1, keepdim=True) - 2 * torch.matmul(x, k_w) + torch.sum(k_w ** 2, dim=0, keepdim=True)
    _, idx = torch.min(dist, dim=-1)
    return idx, dist


# --------------------------------------------------------------------------- geometry
def rot6d_to_rotmat(x):
    """tokenhmr/lib/utils/geometry.py:64-84: a1=x[0:3], a2=x[3:6]; rows b1,b2,b1xb2;
    F.normalize (L2, eps 1e-12)."""
    x = x.reshape(-1, 2, 3).permute(0, 2, 1).contiguous()
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum("bi,bi->b", b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def aa_to_rotmat(theta):
    """tokenhmr/lib/utils/geometry.py:5-44: axis-angle -> quaternion (angle = ||theta + 1e-8||, axis = theta / angle) ->
    quat_to_rotmat (re-normalised quaternion, the 9 quadratic forms)."""
    norm = torch.norm(theta + 1e-8, p=2, dim=1)
    angle = norm.unsqueeze(-1)
    normalized = theta / angle
    angle = angle * 0.5
    quat = torch.cat([torch.cos(angle), torch.sin(angle) * normalized], dim=1)
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def perspective_projection(points, translation, focal):
    """geometry.py:86-124 with rotation=I, camera_center=0: ((p+t)/z) * f."""
    p = points + translation.unsqueeze(1)
    p = p / p[:, :, -1].unsqueeze(-1)
    return p[:, :, :2] * focal.unsqueeze(1)


# --------------------------------------------------------------------------- SMPL (smplx==0.1.28)
def smpl_forward(global_orient, body_pose, betas, smpl):
    """smplx.SMPLLayer.forward(pose2rot=False) -> smplx.lbs.lbs, then the reference wrapper
    tokenhmr/lib/models/smpl_wrapper.py:27-41 (joint_map remap + J19 regressor).
    Restated from smplx's published algorithm; parity UNPINNED."""
    B = betas.shape[0]
    R = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, -1, 3, 3)], dim=1)   # (B,24,3,3)
    v_shaped = smpl["v_template"][None] + torch.einsum("bl,mkl->bmk", betas, smpl["shapedirs"])
    J = torch.einsum("bik,ji->bjk", v_shaped, smpl["J_regressor"])
    ident = torch.eye(3, dtype=R.dtype)
    pose_feature = (R[:, 1:] - ident).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, smpl["posedirs"]).reshape(B, -1, 3)
    parents = smpl["parents"].long()
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, R.shape[1], 4, 4, dtype=R.dtype)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1.0
    chain = [T[:, 0]]
    for i in range(1, R.shape[1]):
        chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
    G = torch.stack(chain, dim=1)
    J_transformed = G[:, :, :3, 3]
    Jh = F.pad(J, [0, 1]).unsqueeze(-1)                       # (B,24,4,1), w=0
    A = G - F.pad(torch.matmul(G, Jh), [3, 0])                # remove rest-pose joint
    W = smpl["lbs_weights"]
    Tv = torch.matmul(W[None].expand(B, -1, -1), A.reshape(B, R.shape[1], 16)).reshape(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=R.dtype)], dim=2)
    verts = torch.matmul(Tv, vh.unsqueeze(-1))[:, :, :3, 0]
    joints = torch.cat([J_transformed, verts[:, smpl["extra_verts"].long()]], dim=1)       # 45
    joints = joints[:, smpl["joint_map"].long()]                                            # 25
    if smpl.get("update_hips", False):                                                      # smpl_wrapper.py:33-36
        joints[:, [9, 12]] = joints[:, [9, 12]] + \
            0.25 * (joints[:, [9, 12]] - joints[:, [12, 9]]) + \
            0.5 * (joints[:, [8]] - 0.5 * (joints[:, [9, 12]] + joints[:, [12, 9]]))
    extra = torch.einsum("bik,ji->bjk", verts, smpl["J19_regressor"])                       # 19
    return verts, torch.cat([joints, extra], dim=1)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues (smplx==0.1.28; un-vendored -> restated, UNPINNED): (N,3) axis-angle -> (N,3,3)."""
    N = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle).unsqueeze(1), torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros(N, 1, dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(N, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def smpl_forward_axis_angle(global_orient, body_pose, betas, smpl):
    """smplx.SMPL.forward(pose2rot=True) as used for GT meshes (tokenhmr/lib/datasets/image_dataset.py:254-270)."""
    B = betas.shape[0]
    full = torch.cat([global_orient.reshape(B, 3), body_pose.reshape(B, 69)], dim=1)
    R = batch_rodrigues(full.reshape(-1, 3)).view(B, 24, 3, 3)
    return smpl_forward(R[:, :1], R[:, 1:], betas, smpl)


# --------------------------------------------------------------------------- full path
def head_forward(ctx, sd, tok, cfg: HMRConfig = RELEASE):
    """token_head.py:65-128 SMPLTokenDecoderHead.forward (IEF_ITERS=1, zero token)."""
    B = ctx.shape[0]
    token_out = decoder_forward(ctx, sd, cfg)
    H = "smpl_head."
    grot = F.linear(token_out, sd[H + "decpose_grot.weight"], sd[H + "decpose_grot.bias"])
    logits = classifier_logits(token_out, sd, cfg)
    probs = logits.softmax(-1)                                          # token_classifier.py:104
    bpose = vq_decode(probs, tok, cfg).reshape(B, -1)                   # :105-107
    hands = F.linear(token_out, sd[H + "decpose_hands.weight"], sd[H + "decpose_hands.bias"])
    pose6d = torch.cat([grot, bpose, hands], -1) + sd[H + "init_body_pose"]
    betas = F.linear(token_out, sd[H + "decshape.weight"], sd[H + "decshape.bias"]) + sd[H + "init_betas"]
    cam = F.linear(token_out, sd[H + "deccam.weight"], sd[H + "deccam.bias"]) + sd[H + "init_cam"]
    rotmat = rot6d_to_rotmat(pose6d).view(B, cfg.n_joints, 3, 3)
    return dict(token_out=token_out, logits=logits, probs=probs, token_idx=token_indices(logits),
                pose6d=pose6d, betas=betas, cam=cam, rotmat=rotmat)


def forward(img, sd, tok, smpl, cfg: HMRConfig = RELEASE, taps=None):
    """tokenhmr/lib/models/tokenhmr.py:135-188 forward_step (eval)."""
    B = img.shape[0]
    ctx = vit_forward(img, sd, cfg, taps)
    h = head_forward(ctx, sd, tok, cfg)
    cam = h["cam"]
    focal = cfg.focal_length * torch.ones(B, 2, dtype=img.dtype)
    cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * focal[:, 0] / (cfg.img_size * cam[:, 0] + 1e-9)], dim=-1)
    R = h["rotmat"]
    verts, joints = smpl_forward(R[:, [0]], R[:, 1:], h["betas"], smpl)
    kp2d = perspective_projection(joints, cam_t, focal / cfg.img_size)
    out = {
        "cls_logits_softmax": h["probs"],
        "pred_cam": cam,
        "pred_smpl_params": {"global_orient": R[:, [0]], "body_pose": R[:, 1:], "betas": h["betas"]},
        "pred_cam_t": cam_t,
        "focal_length": focal,
        "pred_keypoints_3d": joints,
        "pred_vertices": verts,
        "pred_keypoints_2d": kp2d,
        # extras (not in the reference dict) used by parity tests
        "vit_features": ctx, "token_out": h["token_out"], "cls_logits": h["logits"],
        "token_idx": h["token_idx"], "pose6d": h["pose6d"],
    }
    return out
