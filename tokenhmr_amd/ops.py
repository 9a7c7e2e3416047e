"""Stateless operator wrappers over the C ABI (thmr_op_*), used for per-kernel parity tests and
micro-benchmarks.  Inputs/outputs are contiguous fp32 CUDA tensors; no CPU path exists."""
import ctypes as C

import torch

from . import _cabi

# Variant ids of thmr_op_gemm beyond the ones the engine uses (include/tokenhmr_hip.h lists those): A/B material kept for the per-kernel tests
# and scripts/ — 0 / 1 = register-staged 128x128 / 128x160 tiles (lost to the LDS-DMA tiles 7 / 8 in round 1), 12 / 13 = 64x128 / 128x64 tiles
# (faster stand-alone, slower in the pipeline: round 3), 110 + j = the ring kernel 8-deep (ring depth A/B), 31-53 = timing-only ablations
# of builds with -DTHMR_GEMM_ABLATION.  thmr_op_gemm_split3: 1 = 128x256 on 4 waves, 4 = 256x256, 100-102 = ring kernel on split3 operands,
# 20 / 22 / 310-312 = the round-4 first versions on v_mfma_f32_32x32x16_bf16, 3 / 31-37 = schedule experiments; thmr_op_gemm_split3_out_split3:
# 1, 4, 100, 301 (persistent kernel, LDS epilogue), 311 / 312 — all of these exist in the experiments build only (SPLIT3_EXP_ONLY below).
EPI = {"none": 0, "bias": 1, "bias_gelu": 2, "bias_relu": 3, "bias_resid": 4, "bias_qscale": 5, "bias_pos": 6}
VARIANT = {"auto": -1, "128x128reg": 0, "128x160reg": 1, "skinny": 2, "128x128": 7, "128x160": 8, "64x64": 9, "128x96": 10, "tiny": 11, "64x128": 12, "128x64": 13, "ring16": 120}
# small-M ring kernel: "ring4" / "ring8" = LDS ring depth, optional "/k<S>" = split-K factor (1, 2, 4, 8, 16)
VARIANT.update({f"ring{r}" + (f"/k{1 << j}" if j else ""): 100 + (10 if r == 8 else 0) + j for r in (4, 8) for j in range(5)})
# split-K on the big LDS-DMA tiles (mid-size batches): "<tile>/k2", "<tile>/k4", "auto/k2", "auto/k4"
VARIANT.update({f"{t}/k{k}": 100 * k + c for t, c in (("auto", 0), ("128x128", 7), ("128x160", 8), ("128x96", 10), ("64x128", 12), ("128x64", 13)) for k in (2, 4)})
# (timing-only ablation kernels 31-53 exist only in builds with -DTHMR_GEMM_ABLATION, see gemm_f32.hip)


_EXP = None      # None: the shipped library (or THMR_LIB=exp); True: the experiments build (variants that lost their A/B, knobs)


class experiments_build:
    """`with ops.experiments_build(): ...` — the wrappers below call the -DTHMR_EXPERIMENTS library inside the block (tests / scripts)."""

    def __enter__(self):
        global _EXP
        self._old, _EXP = _EXP, True
        return self

    def __exit__(self, *a):
        global _EXP
        _EXP = self._old
        return False


def _L():
    return _cabi.load(exp=_EXP)


def _check(rc):
    _cabi.check(rc, None, _L())


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _req(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("ops need contiguous float32 CUDA tensors")


def gemm(a, w, bias=None, resid=None, epi="none", qscale=1.0, qcols=0, variant="auto"):
    """C = epilogue(a @ w.T);  a (M,K), w (N,K) — torch.nn.Linear layout."""
    _req(a, w, bias, resid)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        _check(_L().thmr_op_gemm(_p(a), K, _p(w), _p(bias), _p(resid), _p(out), N, M, N, K, EPI[epi],
                                             float(qscale), int(qcols), VARIANT[variant], _s(a)))
    return out


SPLIT3_VARIANT = {"auto": -1, "128x256/w8": 0, "tail": 5,        # tail: the 128x256 grid with its ragged last round as 128x128 half tiles (qualifying shapes only)
                   "128x128/w8": 6, "tail/w8": 7, "128x128/w4/s3": 8, "128x128/w8/s3": 9, "128x256/w8/front": 10, "128x128/w4/s3/front": 11,   # front: every copy of a K tile issued right behind the barrier
                      # s3: the four-wave tile with a three-stage K ring
                                   # round 6: the 128x128 tile / the tail's half tiles on eight waves of 64x32 (measured slower / equal: not the rule's choice)
                   "128x256/w4": 1, "128x128/w4": 2, "256x256/w4": 4, "ring": 100, "ring/k2": 101, "ring/k4": 102, "auto/k2": 202, "auto/k4": 204,
                  # 256 persistent workgroups over a tile stream (csrc/gemm_split_persist.hip; M % 128 == 0, N % 256 == 0, >= 256 tiles):
                  # fp32 output / split3 output through the LDS transposition / split3 output through swapped operand roles
                  "persist": 300, "persist/swap": 302, "persist/128x128": 320, "persist/128x128/k2": 322, "persist/128x128/k4": 324,      # 320: the stream over 128x128 tiles, three-stage ring (round 6)
                  # round-4 first versions on v_mfma_f32_32x32x16_bf16 (the product kernels moved to 16x16x32, csrc/gemm_split16.hip): experiments build
                  "old/128x256/w8": 20, "old/128x128/w4": 22, "old/persist": 310, "old/persist/lds": 311, "old/persist/swap": 312,
                  # schedule experiments (epilogue "none" only; the abl/* ones are timing-only, their results are garbage)
                  "exp/reads-every-2nd": 3, "abl/no-copies": 31, "abl/no-barrier": 32, "abl/no-reads": 34, "abl/none": 37}


# variants that exist only in the experiments build (measured slower or timing-only; csrc/gemm_split.hip, gemm_split_persist.hip): asking
# for one of them routes THAT call to libtokenhmr_hip_exp.so
SPLIT3_EXP_ONLY = {"128x128/w4/s3/front", "128x128/w8", "128x128/w8/s3", "tail/w8", "128x256/w4", "256x256/w4", "ring", "ring/k2", "ring/k4", "exp/reads-every-2nd", "abl/no-copies",
                   "abl/no-barrier", "abl/no-reads", "abl/none", "old/128x256/w8", "old/128x128/w4", "old/persist", "old/persist/lds", "old/persist/swap"}


def _L_for(exp_only):
    return _cabi.load(exp=True) if exp_only else _L()


def split3(x):
    """fp32 (R,K) -> the "split3" operand of gemm_split3: every element as three bf16 pieces h + m + l, laid out [R][K/8][3][8]
    (returned as an int16 tensor of shape (R, K/8, 3, 8); see csrc/gemm_split.hip).  K % 8 == 0."""
    _req(x)
    R, K = x.shape
    out = torch.empty(R, K // 8, 3, 8, device=x.device, dtype=torch.int16)
    with torch.cuda.device(x.device):
        _check(_L().thmr_op_split3(_p(x), K, _p(out), K, R, K, _s(x)))
    return out


def split3_block(x_s):
    """row-major split3 operand (R, K/8, 3, 8) -> the ROW-BLOCKED form (ceil(R/32), K/8, 3, 32, 8) the engine uses between fc1 and fc2
    (csrc/common.h GemmArgs::a_blk): 32-row panels of 512 contiguous bytes per 16-byte chunk; rows past R are zero."""
    R, G = x_s.shape[0], x_s.shape[1]
    Rp = (R + 31) // 32 * 32
    pad = torch.zeros(Rp, G, 3, 8, device=x_s.device, dtype=x_s.dtype)
    pad[:R] = x_s
    return pad.view(Rp // 32, 32, G, 3, 8).permute(0, 2, 3, 1, 4).contiguous()


def split3_unblock(x_b, R):
    """inverse of split3_block: (ceil(R/32), K/8, 3, 32, 8) -> (R, K/8, 3, 8)"""
    nb, G = x_b.shape[0], x_b.shape[1]
    return x_b.permute(0, 3, 1, 2, 4).reshape(nb * 32, G, 3, 8)[:R].contiguous()


def gemm_split3(a_s, w_s, bias=None, resid=None, epi="none", qscale=1.0, qcols=0, variant="128x256/w8", out_split=False,
                a_blocked_rows=None, out_blocked=False):
    """C = epilogue(a @ w.T) for split3 operands (`split3(a)`, `split3(w)`) on the bf16 matrix pipe with fp32-grade results: six
    bf16 products per element pair, fp32 accumulation.  `out_split`: the result as a split3 operand (what the engine's fc1 hands fc2);
    `out_blocked`: that operand in the row-blocked form (`split3_block`).  `a_blocked_rows=M`: `a_s` IS in the row-blocked form and has M rows.
    What the engine runs for its ViT GEMMs in the default mode (`Engine.set_vit_gemm("split3")`, the creation default since ABI 4)."""
    _req(bias, resid)
    if a_blocked_rows is not None:
        if not (a_s.is_cuda and a_s.dtype == torch.int16 and a_s.is_contiguous() and a_s.dim() == 5 and a_s.shape[2:] == (3, 32, 8)):
            raise ValueError("a row-blocked split3 operand is int16 (ceil(R/32), K/8, 3, 32, 8)")
        M, K = int(a_blocked_rows), a_s.shape[1] * 8
    else:
        M, K = a_s.shape[0], a_s.shape[1] * 8
    for t in ((w_s,) if a_blocked_rows is not None else (a_s, w_s)):
        if not (t.is_cuda and t.dtype == torch.int16 and t.is_contiguous() and t.dim() == 4 and t.shape[2:] == (3, 8)):
            raise ValueError("gemm_split3 needs split3 operands (int16, (R, K/8, 3, 8))")
    N = w_s.shape[0]
    if w_s.shape[1] * 8 != K:
        raise ValueError("K mismatch")
    lib = _L_for(variant in SPLIT3_EXP_ONLY)
    code = SPLIT3_VARIANT[variant]
    if (out_blocked or a_blocked_rows is not None) and code < 0:
        raise ValueError("name the kernel (not 'auto') with a row-blocked operand")
    if out_split:
        if out_blocked:
            alloc = torch.empty if M % 32 == 0 else torch.zeros            # the kernels never write the pad rows of the last block
            out = alloc((M + 31) // 32, N // 8, 3, 32, 8, device=a_s.device, dtype=torch.int16)
        else:
            out = torch.empty(M, N // 8, 3, 8, device=a_s.device, dtype=torch.int16)
        with torch.cuda.device(a_s.device):
            _cabi.check(lib.thmr_op_gemm_split3_out_split3(_p(a_s), K, _p(w_s), K, _p(bias), _p(out), N, M, N, K, EPI[epi],
                                                           float(qscale), int(qcols), code + (1000 if out_blocked else 0), _s(a_s)), None, lib)
        return out
    out = torch.empty(M, N, device=a_s.device, dtype=torch.float32)
    with torch.cuda.device(a_s.device):
        _cabi.check(lib.thmr_op_gemm_split3(_p(a_s), K, _p(w_s), K, _p(bias), _p(resid), _p(out), N, M, N, K, EPI[epi],
                                            float(qscale), int(qcols), code + (1000 if a_blocked_rows is not None else 0), _s(a_s)), None, lib)
    return out


def layernorm(x, gamma, beta, eps, relu=False):
    _req(x, gamma, beta)
    rows, D = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(_L().thmr_op_layernorm(_p(x), _p(gamma), _p(beta), _p(y), rows, D, float(eps), int(relu), _s(x)))
    return y


ATTN_VARIANT = {"auto": 0, "q64": 1, "q192": 3, "persistent": 5, "q16x12": 12, "keysplit": 6, "keysplit/q16": 61, "keysplit/q32": 62, "keysplit/q48": 63}


def vit_attention(qkv, variant="auto"):
    """qkv (B,192,3840) with q pre-scaled -> (B,192,1280).  variant: "auto" (batch-size rule) or a forced kernel, see ATTN_VARIANT."""
    _req(qkv)
    B = qkv.shape[0]
    out = torch.empty(B, 192, 1280, device=qkv.device, dtype=torch.float32)
    with torch.cuda.device(qkv.device):
        if variant == "auto":
            _check(_L().thmr_op_vit_attention(_p(qkv), _p(out), B, _s(qkv)))
        else:
            lib = _L_for(variant == "q16x12")           # 12 waves of 16 queries: measured no faster, experiments build only
            _cabi.check(lib.thmr_op_vit_attention_variant(_p(qkv), _p(out), B, ATTN_VARIANT[variant], _s(qkv)), None, lib)
    return out


def vit_attention_split3(qkv):
    """vit_attention with the output as a split3 operand (int16 (B*192, 160, 3, 8)): what the engine's split3 mode hands the proj GEMM."""
    _req(qkv)
    B = qkv.shape[0]
    out = torch.empty(B * 192, 160, 3, 8, device=qkv.device, dtype=torch.int16)
    with torch.cuda.device(qkv.device):
        _check(_L().thmr_op_vit_attention_split3(_p(qkv), _p(out), B, _s(qkv)))
    return out


def vit_attention_b16(qkv, out_split=False, qt=0):
    """The attention on the bf16 matrix pipe (three bf16 pieces per operand, six products, fp32 accumulate: csrc/attention_b16.hip).
    out_split: the result as a split3 operand (int16 (B*192, 160, 3, 8)) instead of fp32 (B,192,1280); qt: 0 = batch-size rule, 1 / 3 = forced."""
    _req(qkv)
    B = qkv.shape[0]
    out = (torch.empty(B * 192, 160, 3, 8, device=qkv.device, dtype=torch.int16) if out_split
           else torch.empty(B, 192, 1280, device=qkv.device, dtype=torch.float32))
    with torch.cuda.device(qkv.device):
        _check(_L().thmr_op_vit_attention_b16(_p(qkv), _p(out), B, int(out_split), int(qt), _s(qkv)))
    return out


def rot6d_to_rotmat(x):
    _req(x)
    x2 = x.reshape(-1, 6).contiguous()
    n = x2.shape[0]
    R = torch.empty(n, 3, 3, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _check(_L().thmr_op_rot6d(_p(x2), _p(R), n, _s(x)))
    return R


def aa_to_rotmat(theta):
    """geometry.py:5-44 aa_to_rotmat: (n,3) axis-angle -> (n,3,3)."""
    _req(theta)
    x = theta.reshape(-1, 3).contiguous()
    n = x.shape[0]
    R = torch.empty(n, 3, 3, device=theta.device, dtype=torch.float32)
    with torch.cuda.device(theta.device):
        _check(_L().thmr_op_aa_to_rotmat(_p(x), _p(R), n, _s(theta)))
    return R
