"""Host-side mirror of the reference's model API for the inference hot path.

    model, cfg = load_tokenhmr(checkpoint_path, model_cfg)        tokenhmr/lib/models/__init__.py:3-26
    out = model(batch)                                            tokenhmr/lib/models/tokenhmr.py:330-338
so that tokenhmr/eval.py:147, demo.py:78 and track.py:39 can call it unchanged: same ctor entry
point, same `forward(batch) -> dict` keys / shapes / dtypes (tokenhmr.py:156-188), `.to()`,
`.eval()`, `.cfg`, `.smpl.faces`.  All arithmetic happens in libtokenhmr_hip.so.
"""
import os
import types

import torch

from .config import HMRConfig, RELEASE
from .engine import Engine
from . import weights as W
from .smpl_assets import load_smpl_pkl


class _SmplHandle:
    """What callers read from `model.smpl` (demo.py:52 uses .faces)."""

    def __init__(self, faces):
        self.faces = faces.cpu().numpy() if torch.is_tensor(faces) else faces


class TokenHMR:
    def __init__(self, cfg: HMRConfig = RELEASE, max_batch: int = 64, device="cuda:0", model_cfg=None, vit_gemm=None, persistent=True):
        self.hmr_cfg = cfg
        self.cfg = model_cfg if model_cfg is not None else types.SimpleNamespace(**cfg.to_dict())
        self.max_batch = max_batch
        self.device = torch.device(device)
        self.engine = Engine(cfg, max_batch=max_batch, device=device, vit_gemm=vit_gemm, persistent=persistent)
        self.smpl = _SmplHandle(torch.zeros(13776, 3, dtype=torch.int64))
        self.training = False
        self.return_taps = False

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_state(cls, cfg, state, tokenizer, smpl, max_batch=64, device="cuda:0", model_cfg=None, vit_gemm=None, persistent=True):
        """vit_gemm / persistent: creation-time choices of the engine (Engine.__init__): "f32" creates it in the opt-out mode, so that
        finalize builds no split3 copies at all; persistent=False for a GPU shared with another process."""
        W.validate_state(state, cfg, tokenizer)
        m = cls(cfg, max_batch=max_batch, device=device, model_cfg=model_cfg, vit_gemm=vit_gemm, persistent=persistent)
        m.engine.load_state(state, tokenizer)
        m.engine.load_smpl(smpl)
        m.engine.finalize()
        m.smpl = _SmplHandle(smpl["faces"])
        return m

    @classmethod
    def from_engine(cls, engine, faces=None, model_cfg=None):
        """The facade around an engine that is already loaded and finalized (bench.py times it beside Engine.forward)."""
        m = cls.__new__(cls)
        m.hmr_cfg = engine.cfg
        m.cfg = model_cfg if model_cfg is not None else types.SimpleNamespace(**engine.cfg.to_dict())
        m.max_batch, m.device, m.engine = engine.max_batch, engine.device, engine
        m.smpl = _SmplHandle(faces if faces is not None else torch.zeros(13776, 3, dtype=torch.int64))
        m.training, m.return_taps = False, False
        return m

    # ---- nn.Module-like surface used by eval.py:52-54 / demo.py:35-37 ------------------------
    def to(self, device):
        if torch.device(device).type == "cuda" and torch.device(device) != self.engine.device and torch.device(device).index is not None:
            raise RuntimeError("engine is pinned to %s; build it on the target device" % self.engine.device)
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("tokenhmr_amd implements the inference path only")
        return self

    def __call__(self, batch):
        return self.forward(batch)

    # ---- the hot path -------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, batch):
        """tokenhmr.py:135-188: batch['img'] (B,3,256,256) fp32 -> output dict.  Batches larger
        than max_batch are processed in max_batch chunks."""
        img = batch["img"]
        if img.device != self.engine.device:      # type AND index: a cuda:1 batch must not reach kernels launched on cuda:0
            raise RuntimeError(f"batch['img'] is on {img.device}, the engine on {self.engine.device}: move the batch first "
                               "(recursive_to(batch, device), eval.py:145) — there is no CPU path and no peer-access path")
        if img.dtype != torch.float32:
            img = img.float()
        B = img.shape[0]
        if B <= self.max_batch:
            return self._pack(self.engine.forward(img, taps=self.return_taps))
        outs = [self._pack(self.engine.forward(img[i:i + self.max_batch], taps=self.return_taps))
                for i in range(0, B, self.max_batch)]
        return _cat_outputs(outs)

    def forward_step(self, batch, train=False):
        if train:
            raise NotImplementedError("inference only")
        return self.forward(batch)

    def _pack(self, o):
        R = o["rotmat"]
        out = {
            "cls_logits_softmax": o["cls_logits_softmax"],
            "pred_cam": o["pred_cam"],
            "pred_smpl_params": {"global_orient": R[:, :1], "body_pose": R[:, 1:], "betas": o["betas"]},
            "pred_cam_t": o["pred_cam_t"],
            "focal_length": o["focal_length"],
            "pred_keypoints_3d": o["pred_keypoints_3d"],
            "pred_vertices": o["pred_vertices"],
            "pred_keypoints_2d": o["pred_keypoints_2d"],
            "token_idx": o["token_idx"],          # build-defined extra
        }
        for k in ("vit_features", "token_out", "cls_logits", "pose6d"):
            if k in o:
                out[k] = o[k]
        return out


def _cat_outputs(outs):
    res = {}
    for k in outs[0]:
        if isinstance(outs[0][k], dict):
            res[k] = {kk: torch.cat([o[k][kk] for o in outs], 0) for kk in outs[0][k]}
        else:
            res[k] = torch.cat([o[k] for o in outs], 0)
    return res


# ------------------------------------------------------------------------------------------------
class ConfigNode(dict):
    """What the callers do with the yacs CfgNode that `get_config` returns (tokenhmr/lib/configs/__init__.py:87-103), without yacs:
    attribute AND mapping access (`cfg.MODEL.IMAGE_SIZE`, `'BBOX_SHAPE' in cfg.MODEL`, `cfg.get('ckpt_path')`, `dict(cfg.SMPL)` at
    eval.py:119), assignment (`cfg.ckpt_path = ...`, lib/models/__init__.py:12), and `defrost()` / `freeze()` / `clone()` as no-op-ish
    calls (`:9,23`)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigNode(v) if isinstance(v, dict) and not isinstance(v, ConfigNode) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        return self

    def freeze(self):
        return self

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), ConfigNode):
                self[k].merge(v)
            else:
                self[k] = v
        return self


# the defaults `get_config` merges the file into (tokenhmr/lib/configs/__init__.py:15-62): data, not code
_REFERENCE_DEFAULTS = {
    "GENERAL": {"RESUME": True, "TIME_TO_RUN": 3300, "VAL_STEPS": 100, "LOG_STEPS": 100, "CHECKPOINT_STEPS": 20000,
                "CHECKPOINT_DIR": "checkpoints", "SUMMARY_DIR": "tensorboard", "NUM_GPUS": 1, "NUM_WORKERS": 4, "MIXED_PRECISION": True,
                "ALLOW_CUDA": True, "PIN_MEMORY": False, "DISTRIBUTED": False, "LOCAL_RANK": 0, "USE_SYNCBN": False, "WORLD_SIZE": 1},
    "TRAIN": {"NUM_EPOCHS": 100, "BATCH_SIZE": 32, "SHUFFLE": True, "WARMUP": False, "NORMALIZE_PER_IMAGE": False, "CLIP_GRAD": False,
              "CLIP_GRAD_VALUE": 1.0},
    "LOSS_WEIGHTS": {},
    "DATASETS": {"CONFIG": {"SCALE_FACTOR": 0.3, "ROT_FACTOR": 30, "TRANS_FACTOR": 0.02, "COLOR_SCALE": 0.2, "ROT_AUG_RATE": 0.6,
                            "TRANS_AUG_RATE": 0.5, "DO_FLIP": True, "FLIP_AUG_RATE": 0.5, "EXTREME_CROP_AUG_RATE": 0.10}},
    "MODEL": {"IMAGE_SIZE": 224},
    "EXTRA": {"FOCAL_LENGTH": 5000},
}


def _read_yaml_cfg(path, merge=True):
    """model_config.yaml reader without yacs (absent offline): the reference's defaults with the file merged in, as `get_config(path)`
    does, returned as a ConfigNode."""
    import yaml
    cfg = ConfigNode(_REFERENCE_DEFAULTS if merge else {})
    with open(path) as f:
        cfg.merge(yaml.safe_load(f) or {})
    return cfg, yaml


def load_tokenhmr(checkpoint_path="", model_cfg="", dataset_dir="", is_train_state=False, is_demo=False,
                  max_batch=64, device="cuda:0", strict=True, vit_gemm=None, shared_gpu=None):
    """Drop-in for tokenhmr/lib/models/__init__.py:3-26: (model, cfg) from the reference's files.  See read_reference_files
    for what is read and how; `device` is where the engine is built (the reference builds on the CPU and the caller moves the
    module, eval.py:52-54 — an engine cannot be moved, so pass the target here; `.to()` of the same device is a no-op).
    `vit_gemm`: "split3" (the engine's default since round 5 — what bench.py's headline measures: the ViT GEMMs and attention of calls of 3
    crops and more on the bf16 matrix pipe with fp32 operands as three bf16 pieces, fp32 accumulation; fp32-grade) or "f32" (the opt-out:
    exact-fp32 MFMA everywhere, ~0.65x the rate); None reads $THMR_VIT_GEMM, so that an unmodified eval.py can be switched from the shell,
    and leaves the engine's default alone when that is unset.  The mode is chosen BEFORE the engine is created (round 6): the opt-out never
    builds or holds the split3 weight copies (3.8 GB at release depth).
    `shared_gpu=True` (or $THMR_SHARED_GPU=1): another PROCESS uses this GPU too — the engine then runs none of the kernels that need all
    their workgroups resident at once (persistent decoder, persistent split3 GEMM), which two processes could starve each other on until
    their bounded waits (~0.5 s) expire; same results, a few per cent slower (header: THMR_CFG_NO_PERSISTENT)."""
    hcfg, state, tok, smpl, cfg = read_reference_files(checkpoint_path, model_cfg, dataset_dir, is_train_state, strict)
    mode = vit_gemm if vit_gemm is not None else os.environ.get("THMR_VIT_GEMM")
    if mode not in (None, "f32", "split3"):
        raise ValueError(f"vit_gemm / $THMR_VIT_GEMM must be 'f32' or 'split3', got {mode!r}")
    if shared_gpu is None:
        shared_gpu = os.environ.get("THMR_SHARED_GPU", "") == "1"
    model = TokenHMR.from_state(hcfg, state, tok, smpl, max_batch=max_batch, device=device, model_cfg=cfg, vit_gemm=mode,
                                persistent=not shared_gpu)
    if mode is not None and mode != model.engine.vit_gemm():      # (only a build without creation flags gets here)
        model.engine.set_vit_gemm(mode)
    return model, cfg


def read_reference_files(checkpoint_path="", model_cfg="", dataset_dir="", is_train_state=False, strict=True):
    """The file-reading half of load_tokenhmr (no GPU needed): -> (HMRConfig, state, tokenizer, smpl, cfg).
    Mirrors tokenhmr/lib/models/__init__.py:3-26 (eval branch of TokenHMR.__init__, tokenhmr.py:49-53,84-85).

    Reads the Lightning checkpoint ['state_dict'] (misc.py:242-256), the tokenizer checkpoint named
    by MODEL.TOKENIZER_CHECKPOINT_PATH ['net'] + ['hparams'].ARCH (vanilla_pose_vqvae.py:265-278,299-301) and the SMPL
    pickles named by SMPL.MODEL_PATH / SMPL.JOINT_REGRESSOR_EXTRA, and returns (model, cfg).

    Both checkpoints are un-pickled by `ckpt_io.load_checkpoint` (restricted find_class): the yacs CfgNode in tokenizer.pth
    and the config nodes under a Lightning checkpoint's 'hyper_parameters' need neither yacs, omegaconf nor
    pytorch_lightning to be importable.  `strict=False` mirrors what the reference effectively does with unexpected
    'backbone.*' / 'smpl_head.*' keys (prepare_statedict, misc.py:228-238: log a warning and carry on); missing or
    mis-shaped tensors are always an error here (the reference would run on a random initialisation).
    """
    from . import ckpt_io
    if is_train_state:
        raise NotImplementedError("tokenhmr_amd implements the inference path only")
    cfg, _ = _read_yaml_cfg(model_cfg)
    cfg.ckpt_path = checkpoint_path
    if getattr(cfg.MODEL.BACKBONE, "TYPE", "vit") != "vit":
        raise NotImplementedError("Backbone type is not implemented")
    if getattr(cfg.MODEL.SMPL_HEAD, "TYPE", "token") != "token":
        raise ValueError("Unknown SMPL head type for this engine: only 'token' (tokenhmr_release.yaml:65)")
    assert cfg.MODEL.IMAGE_SIZE == 256, f"MODEL.IMAGE_SIZE ({cfg.MODEL.IMAGE_SIZE}) should be 256 for ViT backbone"
    if "BBOX_SHAPE" not in cfg.MODEL:
        cfg.MODEL.BBOX_SHAPE = [192, 256]
    if dataset_dir != "":
        cfg.DATASETS.DATASET_DIR = dataset_dir
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Missing full pretrained model from {checkpoint_path}")   # reference: exit(1), misc.py:252-254
    td = dict(cfg.MODEL.SMPL_HEAD.TRANSFORMER_DECODER)
    ckpt = ckpt_io.load_checkpoint(checkpoint_path)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise KeyError(f"{checkpoint_path}: no 'state_dict' entry (misc.py:249 reads torch.load(...)['state_dict'])")
    full = ckpt["state_dict"]
    # the backbone depth is a property of the checkpoint (32 for the released ViT-H, vit.py:17), not of model_config.yaml
    blocks = {int(k.split(".")[2]) for k in full if k.startswith("backbone.blocks.")}
    if not blocks or blocks != set(range(len(blocks))):
        raise KeyError(f"checkpoint has no contiguous 'backbone.blocks.N.*' tensors (found indices {sorted(blocks)[:8]})")
    hcfg = HMRConfig(vit_depth=len(blocks), dec_depth=int(td.get("depth", 6)))
    known = {n for n, *_ in W.spec(hcfg)}
    state = ckpt_io.select_state(full, ("backbone.", "smpl_head."), known, strict=strict, what=os.path.basename(checkpoint_path))
    tck = ckpt_io.load_checkpoint(cfg.MODEL.TOKENIZER_CHECKPOINT_PATH)
    if not isinstance(tck, dict) or "net" not in tck:
        raise KeyError(f"{cfg.MODEL.TOKENIZER_CHECKPOINT_PATH}: no 'net' entry (vanilla_pose_vqvae.py:299-301)")
    # DecodeTokens builds its decoder from ckpt['hparams'].ARCH (vanilla_pose_vqvae.py:266-278); the engine's VQ kernels are built
    # for ONE architecture, so the file's must be that one
    ckpt_io.check_tokenizer_arch(ckpt_io.tokenizer_arch(tck), hcfg)
    net = tck["net"]
    tok = {k: v for k, v in net.items() if k.startswith("decoder.decoder.") or k == "quantizer.codebook"}
    enc_names = [n for n, *_ in W.tokenizer_encoder_spec(hcfg)]
    if all(n in net for n in enc_names):        # the tokenizer's encoder half (EncodeTokens, :304-346) enables engine.encode_tokens()
        tok.update({n: net[n] for n in enc_names})
    mean = __import__("numpy").load(cfg.SMPL.MEAN_PARAMS)
    state.setdefault("smpl_head.init_body_pose", torch.from_numpy(mean["pose"].astype("float32")).unsqueeze(0))
    state.setdefault("smpl_head.init_betas", torch.from_numpy(mean["shape"].astype("float32")).unsqueeze(0))
    state.setdefault("smpl_head.init_cam", torch.from_numpy(mean["cam"].astype("float32")).unsqueeze(0))
    gender = str(getattr(cfg.SMPL, "GENDER", "neutral")).upper()
    smpl = load_smpl_pkl(os.path.join(cfg.SMPL.MODEL_PATH, f"SMPL_{gender}.pkl"), cfg.SMPL.JOINT_REGRESSOR_EXTRA, hcfg)
    # tokenhmr.py:84-85 passes every cfg.SMPL key (lower-cased) to SMPL(...): update_hips is one of its keyword arguments
    smpl["update_hips"] = bool(getattr(cfg.SMPL, "UPDATE_HIPS", getattr(cfg.SMPL, "update_hips", False)))
    W.validate_state(state, hcfg, tok)
    return hcfg, state, tok, smpl, cfg
