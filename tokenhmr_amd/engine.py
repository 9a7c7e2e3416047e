"""Python handle on the C-ABI engine (include/tokenhmr_hip.h).  torch is used only for device
memory (arenas, I/O tensors), the current stream and torch.distributed — never for compute."""
import ctypes as C

import torch

from . import _cabi
from .config import HMRConfig, RELEASE
from . import weights as W


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One engine per GPU.  The weight arena is a torch uint8 tensor so that
    torch.distributed.broadcast (RCCL) can replicate it across ranks."""

    def __init__(self, cfg: HMRConfig = RELEASE, max_batch: int = 64, device="cuda:0", weight_arena=None, experiments=None,
                 vit_gemm=None, persistent=True):
        """weight_arena: share another engine's (already loaded) packed weights — a second engine on the same GPU then only
        adds its own scratch arena and split3 activation operands (finalize it with assume_all_loaded=True); the split3 weight copies are
        shared too (one per weight arena and process).
        vit_gemm: None / "split3" = the library's creation default; "f32" = CREATE the engine in the exact-fp32 mode (thmr_config.flags
        THMR_CFG_VIT_GEMM_F32), so that finalize never builds the 3.8 GB of split3 weight copies an opt-out caller does not want.
        persistent=False (THMR_CFG_NO_PERSISTENT, or $THMR_SHARED_GPU=1): none of the kernels that need all their workgroups resident at
        once — for a GPU shared with another PROCESS (header); same results, a few per cent slower."""
        # experiments: None = the shipped library (or THMR_LIB=exp for a whole process); True = the -DTHMR_EXPERIMENTS build, which reads the
        # THMR_* A/B knobs and carries the debug hooks — tests and scripts only; a path = another build of the library (scripts/ab_same_box.py).
        # Each shared object has its OWN per-device turnstile for the persistent kernels (decoder grid barrier, split3 hand-over), so engines
        # of two different libraries must not run CONCURRENTLY (two streams / threads) on one device; one after the other is fine.
        self.lib = _cabi.load(exp=experiments)
        self.cfg = cfg
        self._abi = self.lib.thmr_abi_version()          # (3 only for a previous round's build loaded by path: scripts/ab_same_box.py)
        self.max_batch = int(max_batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _cabi.EngineError("tokenhmr_amd runs on a HIP device only (no CPU fallback)")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        import os
        if vit_gemm not in (None, "f32", "split3"):
            raise ValueError(f"vit_gemm must be None, 'f32' or 'split3', got {vit_gemm!r}")
        if os.environ.get("THMR_SHARED_GPU", "") == "1":
            persistent = False
        flags = (_cabi.CFG_VIT_GEMM_F32 if vit_gemm == "f32" else 0) | (0 if persistent else _cabi.CFG_NO_PERSISTENT)
        if flags and self._abi < 5:
            raise _cabi.EngineError("this build of the library (ABI < 5) has no creation flags")
        self.persistent = bool(persistent)
        self._ccfg = _cabi.Config(abi_version=self._abi, vit_depth=cfg.vit_depth, dec_depth=cfg.dec_depth,
                                  max_batch=self.max_batch, device=idx, flags=flags)
        wb, sb = C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.thmr_arena_bytes(C.byref(self._ccfg), C.byref(wb), C.byref(sb)))
        self.weight_bytes, self.scratch_bytes = wb.value, sb.value
        with torch.cuda.device(self.device):
            if weight_arena is not None:
                if weight_arena.numel() != self.weight_bytes or weight_arena.dtype != torch.uint8 or weight_arena.device != self.device:
                    raise ValueError("weight_arena must be the uint8 arena of an engine with the same config on the same device")
                self.weight_arena = weight_arena
            else:
                self.weight_arena = torch.empty(self.weight_bytes, dtype=torch.uint8, device=self.device)
            self.scratch_arena = torch.empty(self.scratch_bytes, dtype=torch.uint8, device=self.device)
            h = C.c_void_p(0)
            self._check(self.lib.thmr_create(C.byref(self._ccfg), _ptr(self.weight_arena), _ptr(self.scratch_arena),
                                             C.byref(h)))
        self.h = h
        self._keep = []

    def _check(self, rc, handle=None):
        _cabi.check(rc, handle, self.lib)

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            try:
                ctx = torch.cuda.device(self.device)
                ctx.__enter__()
            except Exception:          # interpreter shutdown: torch may already be gone
                ctx = None
            self.lib.thmr_destroy(h)
            if ctx is not None:
                try:
                    ctx.__exit__(None, None, None)
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state(self, state, tokenizer=None):
        """state: TokenHMR state_dict ('backbone.*','smpl_head.*'); tokenizer: tokenizer 'net' dict.
        Host or device fp32 tensors in the reference layouts (strict: missing/unknown keys raise)."""
        items = list(state.items()) + (list(tokenizer.items()) if tokenizer else [])
        wanted = {n for n, *_ in W.spec(self.cfg)} | {n for n, *_ in W.tokenizer_spec(self.cfg)}
        wanted |= {n for n, *_ in W.tokenizer_encoder_spec(self.cfg)}      # optional: enables encode_tokens()
        descs, keep = [], []
        for name, t in items:
            if name not in wanted:
                if name.startswith(("encoder.", "body_model", "decoder.body_model", "quantizer.")):
                    continue   # filtered by the reference too (vanilla_pose_vqvae.py:24-40)
                raise KeyError(f"unexpected tensor '{name}' (strict load)")
            t = t.detach()
            if t.dtype != torch.float32:
                t = t.float()
            t = t.contiguous()
            keep.append(t)
            descs.append(_cabi.TensorDesc(name=name.encode(), data=t.data_ptr(), numel=t.numel(),
                                          on_device=1 if t.is_cuda else 0))
        arr = (_cabi.TensorDesc * len(descs))(*descs)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_load_weights(self.h, arr, len(descs), _stream_ptr(self.device)), self.h)
            torch.cuda.current_stream(self.device).synchronize()   # host staging tensors may now be freed

    def load_smpl(self, smpl):
        keys = ["v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J19_regressor"]
        ikeys = ["parents", "extra_verts", "joint_map"]
        ts = {k: smpl[k].detach().float().contiguous().cpu() for k in keys}
        ts.update({k: smpl[k].detach().to(torch.int32).contiguous().cpu() for k in ikeys})
        shapes = {"v_template": (6890, 3), "shapedirs": (6890, 3, 10), "posedirs": (207, 20670),
                  "J_regressor": (24, 6890), "lbs_weights": (6890, 24), "J19_regressor": (19, 6890),
                  "parents": (24,), "extra_verts": (21,), "joint_map": (25,)}
        for k, s in shapes.items():
            if tuple(ts[k].shape) != s:
                raise ValueError(f"SMPL '{k}': expected {s}, got {tuple(ts[k].shape)}")
        # SMPL(update_hips=...) of the reference wrapper (smpl_wrapper.py:11,33-36); absent = False like its default
        d = _cabi.SmplDesc(**{k: ts[k].data_ptr() for k in keys + ikeys}, on_device=0,
                           update_hips=1 if smpl.get("update_hips", False) else 0)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_load_smpl(self.h, C.byref(d), _stream_ptr(self.device)), self.h)
            torch.cuda.current_stream(self.device).synchronize()

    def finalize(self, assume_all_loaded=False):
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_finalize_weights(self.h, 1 if assume_all_loaded else 0, _stream_ptr(self.device)), self.h)

    # ------------------------------------------------------------------ forward
    def _alloc_outputs(self, B, taps=False, want_probs=True):
        """Fresh output tensors for one call (never reused across calls: a caller may keep the previous call's dict), carved out
        of ONE allocation for the eight small tensors (ten separate torch.empty calls cost ~0.1 ms per call, 3 % of a one-crop forward)
        plus one each for the vertices and the MB-per-crop tensors.  Every tensor starts on a 256-byte boundary; token_idx is an int32 view."""
        f32 = torch.float32
        spec = [("pred_cam", (B, 3)), ("rotmat", (B, 24, 3, 3)), ("betas", (B, 10)), ("pred_cam_t", (B, 3)), ("focal_length", (B, 2)),
                ("pred_keypoints_3d", (B, 44, 3)), ("pred_vertices", (B, 6890, 3)), ("pred_keypoints_2d", (B, 44, 2)),
                ("token_idx", (B, 160))]
        if want_probs:
            spec.append(("cls_logits_softmax", (B, 160, 2048)))
        if taps:
            spec += [("vit_features", (B, 192, 1280)), ("token_out", (B, 1024)), ("cls_logits", (B, 160, 2048)), ("pose6d", (B, 144))]
        # 80 KB ... MB per crop: own allocations, so a kept (or torch.save-d) small tensor neither pins nor serialises them.  The small
        # tensors share ONE storage: they are views — `.clone()` what is to outlive the dict cheaply or to be pickled on its own
        big = {"cls_logits_softmax", "vit_features", "cls_logits", "pred_vertices"}
        offs, total = {}, 0
        for name, shape in spec:
            if name in big:
                continue
            n = 1
            for d in shape:
                n *= d
            offs[name] = (total, n)
            total += (n + 63) & ~63
        buf = torch.empty(total, device=self.device, dtype=f32)
        o = {}
        for name, shape in spec:
            if name in big:
                o[name] = torch.empty(shape, device=self.device, dtype=f32)
                continue
            off, n = offs[name]
            t = buf.narrow(0, off, n)
            o[name] = (t.view(torch.int32) if name == "token_idx" else t).view(shape)
        return o

    def _outputs_struct(self, o):
        return _cabi.Outputs(**{k: (o[k].data_ptr() if k in o else None) for k in _cabi.OUTPUT_FIELDS})

    def _check_img(self, img):
        if not (img.is_cuda and img.device == self.device):
            raise ValueError(f"img must live on {self.device}")
        if img.dtype != torch.float32 or img.dim() != 4 or tuple(img.shape[1:]) != (3, 256, 256):
            raise ValueError(f"img must be float32 (B,3,256,256), got {img.dtype} {tuple(img.shape)}")
        if img.shape[0] > self.max_batch:
            raise ValueError(f"batch {img.shape[0]} > max_batch {self.max_batch}")
        return img.contiguous()

    def forward(self, img, taps=False, want_probs=True, outputs=None):
        img = self._check_img(img)
        B = img.shape[0]
        o = outputs if outputs is not None else self._alloc_outputs(B, taps, want_probs)
        st = self._outputs_struct(o)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_forward(self.h, _ptr(img), B, C.byref(st), _stream_ptr(self.device)), self.h)
        return o

    def status(self):
        """Synchronises the current stream and raises if a kernel of this engine reported an asynchronous error."""
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_engine_status(self.h, _stream_ptr(self.device)), self.h)

    def vit_forward(self, img, out=None):
        img = self._check_img(img)
        B = img.shape[0]
        if out is None:
            out = torch.empty(B, 192, 1280, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_vit_forward(self.h, _ptr(img), B, _ptr(out), _stream_ptr(self.device)), self.h)
        return out

    def head_forward(self, ctx, taps=False, want_probs=True):
        ctx = ctx.contiguous()
        B = ctx.shape[0]
        o = self._alloc_outputs(B, taps, want_probs)
        o.pop("vit_features", None)
        st = self._outputs_struct(o)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_head_forward(self.h, _ptr(ctx), B, C.byref(st), _stream_ptr(self.device)), self.h)
        return o

    def lbs_forward(self, rotmat, betas, cam=None):
        B = betas.shape[0]
        dev, f32 = self.device, torch.float32
        rotmat, betas = rotmat.contiguous(), betas.contiguous()
        verts = torch.empty(B, 6890, 3, device=dev, dtype=f32)
        joints = torch.empty(B, 44, 3, device=dev, dtype=f32)
        cam_t = torch.empty(B, 3, device=dev, dtype=f32) if cam is not None else None
        kp2d = torch.empty(B, 44, 2, device=dev, dtype=f32) if cam is not None else None
        cam = cam.contiguous() if cam is not None else None
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_lbs_forward(self.h, _ptr(rotmat), _ptr(betas), _ptr(cam), B, _ptr(verts), _ptr(joints),
                                                  _ptr(cam_t), _ptr(kp2d), _stream_ptr(self.device)), self.h)
        return verts, joints, cam_t, kp2d

    def encode_tokens(self, pose6d, want_latent=False):
        """EncodeTokens.forward (vanilla_pose_vqvae.py:334-342): (B,21,6) rot6d body pose -> (B,160) int32 code indices."""
        pose6d = pose6d.to(self.device, torch.float32).contiguous()
        B = pose6d.shape[0]
        if tuple(pose6d.shape[1:]) != (21, 6):
            raise ValueError(f"pose must be (B,21,6), got {tuple(pose6d.shape)}")
        idx = torch.empty(B, 160, device=self.device, dtype=torch.int32)
        lat = torch.empty(B, 160, 256, device=self.device, dtype=torch.float32) if want_latent else None
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_encode_tokens(self.h, _ptr(pose6d), B, _ptr(idx), _ptr(lat), _stream_ptr(self.device)), self.h)
        return (idx, lat) if want_latent else idx

    def vq_decode(self, probs):
        """DecodeTokens.forward (vanilla_pose_vqvae.py:294-297): (B,160,2048) token probabilities -> (B,21,6) rot6d pose."""
        probs = probs.to(self.device, torch.float32).contiguous()
        B = probs.shape[0]
        if tuple(probs.shape[1:]) != (160, 2048):
            raise ValueError(f"probs must be (B,160,2048), got {tuple(probs.shape)}")
        pose = torch.empty(B, 21, 6, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_vq_decode(self.h, _ptr(probs), B, _ptr(pose), _stream_ptr(self.device)), self.h)
        return pose

    def vq_argmin(self, x, want_dist=False):
        x = x.contiguous()
        rows = x.shape[0]
        idx = torch.empty(rows, device=self.device, dtype=torch.int32)
        dist = torch.empty(rows, 2048, device=self.device, dtype=torch.float32) if want_dist else None
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_vq_argmin(self.h, _ptr(x), rows, _ptr(idx), _ptr(dist), _stream_ptr(self.device)), self.h)
        return (idx, dist) if want_dist else idx

    # ------------------------------------------------------------------ how the ViT GEMMs are multiplied
    VIT_GEMM = {"f32": 0, "split3": 1}

    def set_vit_gemm(self, mode="split3"):
        """"split3" (what an engine is created in): fp32 operands as three bf16 pieces on the bf16 matrix pipe (six products, fp32
        accumulate; fp32-grade, not bitwise fp32) for calls of at least 3 crops.  "f32": the opt-out, exact-fp32 MFMA everywhere — see
        thmr_set_vit_gemm in the header."""
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_set_vit_gemm(self.h, self.VIT_GEMM[mode], _stream_ptr(self.device)), self.h)

    def vit_gemm(self):
        return {v: k for k, v in self.VIT_GEMM.items()}[self.lib.thmr_get_vit_gemm(self.h)]

    def mode_bytes(self, mode=None):
        """Device memory the engine allocates itself, outside its two arenas, in `mode` (default: the current one): dict of the split3
        weight copies (shared among the engines of one weight arena), the split3 activation operands, the hand-over workspace."""
        m = self.VIT_GEMM[mode] if mode is not None else self.lib.thmr_get_vit_gemm(self.h)
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.thmr_mode_bytes(C.byref(self._ccfg), m, C.byref(a), C.byref(b), C.byref(c)))
        return {"split_weights": a.value, "split_activations": b.value, "workspace": c.value}

    # ------------------------------------------------------------------ profiler
    def prof_enable(self, on=True):
        """on: False / True (every kernel class) / "gemm" (the four ViT GEMM classes) / "fc1" (only the dominant kernel: cheapest,
        see the header)."""
        mode = {"gemm": 2, "fc1": 3}.get(on, 1 if on else 0)
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_prof_enable(self.h, mode), self.h)

    def prof_collect(self, reset=True):
        arr = (_cabi.ProfEntry * len(_cabi.PROF_NAMES))()
        with torch.cuda.device(self.device):
            self._check(self.lib.thmr_prof_collect(self.h, arr, 1 if reset else 0), self.h)
        return {n: dict(ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes, launches=arr[i].launches)
                for i, n in enumerate(_cabi.PROF_NAMES)}
