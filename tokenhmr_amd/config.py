"""Hyper-parameters of the TokenHMR inference hot path (release config).

These are *data* taken from the reference, not code:
  ViT-H/16        tokenhmr/lib/models/backbones/vit.py:12-24
  decoder         tokenhmr/lib/configs_hydra/experiment/tokenhmr_release.yaml:73-81,
                  tokenhmr/lib/models/heads/token_head.py:30-38
  classifier      tokenhmr/lib/models/heads/token_classifier.py:56-63
  tokenizer ARCH  tokenization/configs/tokenizer_amass_moyo.yaml:41-54
  SMPL / camera   tokenhmr_release.yaml:31-37,57 ; experiment/default.yaml:11-15

Only `vit_depth` and `dec_depth` may be reduced (for fast parity tests); every
other dimension is baked into the HIP kernels (192 tokens, 16x80 heads, ...).
"""
from dataclasses import dataclass, asdict
import numpy as np


@dataclass(frozen=True)
class HMRConfig:
    # ViT-H/16 on a 256x192 window of the 256x256 crop
    img_size: int = 256
    crop_w: int = 192            # vit.py:342  x[:, :, :, 32:-32]
    crop_x0: int = 32
    patch: int = 16
    patch_pad: int = 2           # vit.py:168  4 + 2*(ratio//2-1), ratio=1
    grid_h: int = 16
    grid_w: int = 12
    tokens: int = 192
    dim: int = 1280
    heads: int = 16
    head_dim: int = 80
    mlp_dim: int = 5120
    vit_depth: int = 32
    vit_ln_eps: float = 1e-6     # vit.py:222
    # cross-attention decoder
    dec_dim: int = 1024
    dec_depth: int = 6
    dec_heads: int = 8
    dec_head_dim: int = 64
    dec_mlp: int = 1024
    ln_eps: float = 1e-5         # torch.nn.LayerNorm default
    # token classifier (MLP-Mixer)
    token_num: int = 160
    token_classes: int = 2048
    mix_hidden: int = 64
    mix_hidden_inter: int = 256
    mix_token_inter: int = 64
    mix_blocks: int = 4
    # VQ-VAE decoder
    code_dim: int = 256
    vq_width: int = 512
    vq_joints: int = 21
    vq_dilation: int = 3
    # SMPL
    n_verts: int = 6890
    n_joints: int = 24
    n_betas: int = 10
    n_posedirs: int = 207
    n_extra: int = 21            # smplx vertex_ids['smplh'] picks
    n_j19: int = 19
    n_out_joints: int = 44
    focal_length: float = 5000.0

    @property
    def inner(self) -> int:      # decoder attention inner dim
        return self.dec_heads * self.dec_head_dim

    @property
    def vq_lengths(self):
        """Temporal sizes of the VQ decoder: 160 -> 125 -> 90 -> 55 -> 21
        (vanilla_pose_vqvae.py:139, np.linspace(21,160,4,endpoint=False)[::-1])."""
        ups = list(np.linspace(self.vq_joints, self.token_num, 4, endpoint=False, dtype=int)[::-1])
        return [self.token_num] + [int(u) for u in ups]

    def to_dict(self):
        return asdict(self)


RELEASE = HMRConfig()

# smplx.SMPL kinematic tree (smplx==0.1.28, body_models.py / SMPL pkl 'kintree_table')
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# smplx vertex_ids['smplh'] in VertexJointSelector order (nose, reye, leye, rear, lear,
# LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel, l/r thumb..pinky tips)
SMPL_EXTRA_VERTS = [332, 6260, 2800, 4071, 583,
                    3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673,
                    6191, 5782, 5905, 6016, 6133]

# tokenhmr/lib/models/smpl_wrapper.py:19-20
SMPL_TO_OPENPOSE = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4,
                    7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34]
