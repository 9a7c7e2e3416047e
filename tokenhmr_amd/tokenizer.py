"""Drop-ins for the tokenizer's two stand-alone entry points:

    from tokenhmr_amd.tokenizer import DecodeTokens, EncodeTokens       # was: from tokenization.models.vanilla_pose_vqvae import ...
    pose6d = DecodeTokens(ckpt_path)(logits)        # (B,160,2048) token probabilities -> (B,21,6)   vanilla_pose_vqvae.py:258-301
    idx    = EncodeTokens(ckpt_path)(pose6d)        # (B,21,6) -> (B*160,) int64 code indices         vanilla_pose_vqvae.py:304-346

Same constructor arguments, same call, same result layout as the reference classes; the arithmetic runs in the HIP engine
(`thmr_vq_decode`: soft codebook lookup + PoseSPDecoderV1; `thmr_encode_tokens`: PoseSPEncoderV1 + argmin-L2 quantiser).  The file is
the reference's `tokenizer.pth` ({'net', 'hparams': yacs CfgNode}), read without yacs (`ckpt_io.load_checkpoint`), its `hparams.ARCH` checked
against the architecture the kernels are built for.  An engine serves the whole TokenHMR path, so a tokenizer-only one is an engine of
ViT / decoder depth 1 whose other weights are zeros (~80 MB); pass `engine=` to share the engine of a loaded model instead
(`load_tokenhmr(...)[0].engine`: its tokenizer is the one the model was trained with).
"""
import torch

from . import ckpt_io
from . import weights as W
from .config import HMRConfig


def _tokenizer_engine(ckpt_path, device, max_batch, need_encoder):
    from .engine import Engine
    from .smpl_assets import make_synthetic_smpl
    ckpt = ckpt_io.load_checkpoint(ckpt_path)
    if not isinstance(ckpt, dict) or "net" not in ckpt:
        raise KeyError(f"{ckpt_path}: not a tokenizer checkpoint (no 'net' entry, vanilla_pose_vqvae.py:299-301)")
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    ckpt_io.check_tokenizer_arch(ckpt_io.tokenizer_arch(ckpt), cfg)
    net = {k: v for k, v in ckpt["net"].items() if torch.is_tensor(v)}
    names = [n for n, *_ in W.tokenizer_spec(cfg)]
    enc_names = [n for n, *_ in W.tokenizer_encoder_spec(cfg)]
    missing = [n for n in names + (enc_names if need_encoder else []) if n not in net]
    if missing:
        raise KeyError(f"{ckpt_path}: tokenizer tensors missing from ckpt['net']: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
    tok = {n: net[n].float() for n in names}
    tok.update({n: net[n].float() for n in enc_names if all(m in net for m in enc_names)})
    # the rest of the contract (backbone / head of depth 1): zeros — never read by the two tokenizer entry points
    rest = {n: torch.zeros(shape) for n, shape, *_ in W.spec(cfg)}
    eng = Engine(cfg, max_batch=max_batch, device=device)
    eng.load_state(rest, tok)
    eng.load_smpl(make_synthetic_smpl(cfg, 0))
    eng.finalize()
    return eng


class _TokenizerModule:
    def __init__(self, ckpt_path, device, max_batch, engine, need_encoder):
        self.device = torch.device(device) if engine is None else engine.device
        self.max_batch = max_batch if engine is None else engine.max_batch
        self.engine = engine if engine is not None else _tokenizer_engine(ckpt_path, self.device, max_batch, need_encoder)

    # nn.Module surface the callers touch
    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("tokenhmr_amd implements the inference path only")
        return self

    def to(self, device):
        d = torch.device(device)
        if d.type != self.device.type or (d.index is not None and d.index != self.device.index):
            raise ValueError(f"this module's engine lives on {self.device}; build it with device={device!r} instead of moving it")
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def __call__(self, x):
        with torch.no_grad():
            return self.forward(x)

    def _chunks(self, x):
        for i in range(0, x.shape[0], self.max_batch):
            yield x[i:i + self.max_batch]


class DecodeTokens(_TokenizerModule):
    """vanilla_pose_vqvae.py:258-301.  `mesh_inference` only adds an SMPL mesh to the reference decoder's side outputs, which
    DecodeTokens.forward does not return; accepted and ignored."""

    def __init__(self, ckpt_path="", mesh_inference=False, device="cuda:0", max_batch=64, engine=None):
        super().__init__(ckpt_path, device, max_batch, engine, need_encoder=False)

    def forward(self, logits):
        """logits: (B,160,2048) token PROBABILITIES (the reference names them logits; token_classifier.py:104-106 passes the softmax) ->
        pred_pose_body_6d (B,21,6)."""
        if logits.dim() != 3 or tuple(logits.shape[1:]) != (160, 2048):
            raise ValueError(f"DecodeTokens expects (B,160,2048), got {tuple(logits.shape)}")
        if logits.shape[0] == 0:
            return torch.empty(0, 21, 6, device=self.device)
        return torch.cat([self.engine.vq_decode(c) for c in self._chunks(logits)], 0)


class EncodeTokens(_TokenizerModule):
    """vanilla_pose_vqvae.py:304-346."""

    def __init__(self, ckpt_path="", device="cuda:0", max_batch=64, engine=None):
        super().__init__(ckpt_path, device, max_batch, engine, need_encoder=True)

    def forward(self, x):
        """x: (B,21,6) rot6d body pose -> code indices, int64, flattened to (B*160,) as QuantizeEMAReset.quantize returns them
        (quantize_cnn.py:80-86 on the (B*160,256) rows of `preprocess`, :74-78); `.view(B, -1)` gives VanillaTokenizer.encode's layout."""
        if x.dim() != 3 or tuple(x.shape[1:]) != (21, 6):
            raise ValueError(f"EncodeTokens expects (B,21,6), got {tuple(x.shape)}")
        if x.shape[0] == 0:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        return torch.cat([self.engine.encode_tokens(c) for c in self._chunks(x)], 0).reshape(-1).to(torch.int64)
