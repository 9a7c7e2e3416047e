"""Per-kernel parity (-m gpu): each HIP kernel, called through the C ABI, against the plain fp32
torch CPU statement of the same reference op.  Tolerances are written next to each check: the HIP
kernels are exact-fp32 (fmaf-chain MFMA), so differences vs. MKL/oneDNN are summation-order only."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _gemm_ref(a, w, bias, resid, epi, qscale, qcols):
    c = a.double() @ w.double().t()
    if epi != "none":
        c = c + bias.double()
    if epi == "bias_gelu":
        c = F.gelu(c)
    elif epi == "bias_relu":
        c = F.relu(c)
    elif epi == "bias_resid":
        c = resid.double() + c
    elif epi == "bias_qscale":
        c[:, :qcols] = c[:, :qcols] * qscale
    elif epi == "bias_pos":
        M = a.shape[0]
        t = torch.arange(M) % 192
        c = (c + resid.double()[1 + t]) + resid.double()[0]
    return c.float()


# (M, N, K): exact tiles, ragged M/N edges, tiny N, K = one tile, the real ViT shapes at B=2
SHAPES = [(128, 128, 32), (256, 320, 64), (384, 1280, 1280), (200, 72, 96), (130, 6, 1536), (37, 31, 64),
          (384, 3840, 1280), (384, 1280, 5120), (1, 160, 64), (1344, 512, 512)]


@pytest.mark.parametrize("variant", ["128x128", "128x160", "128x96", "64x64", "64x128", "128x64", "128x128reg", "128x160reg", "auto"])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_shapes(built_lib, cuda_dev, shape, variant):
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3)
    out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), epi="bias", variant=variant).cpu()
    ref = _gemm_ref(a, w, b, None, "bias", 1.0, 0)
    # fp32 fmaf chain vs fp64: |err| <~ 1e-6 * sum|a*w| ~ 1e-6*sqrt(K); allow 2e-5 abs at |C| ~ 1
    assert torch.allclose(out, ref, atol=3e-5, rtol=1e-5), (out - ref).abs().max()


@pytest.mark.parametrize("epi", ["none", "bias", "bias_gelu", "bias_relu", "bias_resid", "bias_qscale", "bias_pos"])
def test_gemm_epilogues(built_lib, cuda_dev, epi):
    from tokenhmr_amd import ops
    M, N, K = 384, 256, 128          # M = 2*192 so that the pos-embed row index wraps
    a, w, b = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=0.1), _rand(N, seed=6)
    resid = _rand(193, N, seed=7) if epi == "bias_pos" else _rand(M, N, seed=7)
    kw = dict(qscale=80 ** -0.5, qcols=100)
    out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), None if epi == "none" else b.to(cuda_dev),
                   resid.to(cuda_dev) if epi in ("bias_resid", "bias_pos") else None, epi=epi, **kw).cpu()
    ref = _gemm_ref(a, w, b, resid, epi, kw["qscale"], kw["qcols"])
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5), (out - ref).abs().max()


RING_SHAPES = [(192, 1280, 1280), (192, 3840, 1280), (384, 1280, 5120), (64, 64, 32), (64, 64, 256), (200, 72, 512), (37, 31, 1024),
               (576, 5120, 1280)]


@pytest.mark.parametrize("variant", ["ring4", "ring8", "ring4/k2", "ring8/k4", "ring4/k8", "ring8/k16"])
@pytest.mark.parametrize("shape", RING_SHAPES)
def test_gemm_ring_shapes(built_lib, cuda_dev, shape, variant):
    """Small-M kernel: every ring depth x split-K factor, including K tiles fewer than the ring is deep (K=32: one tile),
    ragged M / N edges and split-K slices of a single K tile."""
    from tokenhmr_amd import ops
    M, N, K = shape
    ks = int(variant.split("/k")[1]) if "/k" in variant else 1
    if K % (32 * ks):
        pytest.skip("K not divisible by 32*ksplit (rejected, see test_gemm_ring_rejects)")
    if ks > 1 and N % 4:
        pytest.skip("split-K reducer needs N % 4 == 0")
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=4)
    for epi in ("bias", "bias_resid", "bias_gelu"):
        out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), r.to(cuda_dev) if epi == "bias_resid" else None,
                       epi=epi, variant=variant).cpu()
        ref = _gemm_ref(a, w, b, r, epi, 1.0, 0)
        assert torch.allclose(out, ref, atol=3e-5, rtol=1e-5), (epi, (out - ref).abs().max())


def test_gemm_ring_deterministic_and_qscale(built_lib, cuda_dev):
    from tokenhmr_amd import ops
    a, w, b = _rand(192, 1280, seed=1).to(cuda_dev), _rand(3840, 1280, seed=2, scale=0.03).to(cuda_dev), _rand(3840, seed=3).to(cuda_dev)
    kw = dict(epi="bias_qscale", qscale=80 ** -0.5, qcols=1280)
    ref = ops.gemm(a, w, b, variant="128x160", **kw)
    for v in ("ring8", "ring4/k4"):
        o1, o2 = ops.gemm(a, w, b, variant=v, **kw), ops.gemm(a, w, b, variant=v, **kw)
        assert torch.equal(o1, o2)                                  # fixed-order split-K reduction
        assert torch.allclose(o1, ref, atol=2e-5, rtol=1e-5)
    # ksplit = 1 accumulates K in the same order as the big-tile kernel: bit-identical
    assert torch.equal(ops.gemm(a, w, b, variant="ring8", **kw), ref)


SPLITK_SHAPES = [(1536, 1280, 5120), (1344, 1280, 1280), (3072, 1280, 5120), (200, 72, 512), (130, 1280, 256)]


@pytest.mark.parametrize("shape", SPLITK_SHAPES)
def test_gemm_big_tile_splitk(built_lib, cuda_dev, shape):
    """Split-K on the big LDS-DMA tiles (proj / fc2 at 7 ... 23 crops): every tile x split factor against fp64, run-to-run
    determinism, and the two invariants the engine's regimes rely on — within a split factor the result does not depend on the
    tile (each K slice is summed in the one order all big tiles share, the slices are added in a fixed order), and a row's
    result does not depend on how many rows share the launch."""
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=4)
    da, dw, db, dr = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), r.to(cuda_dev)
    for ks in (2, 4):
        if K % (32 * ks):
            continue
        outs = {}
        for tile in ("128x128", "128x160", "128x96", "64x128", "128x64", "auto"):
            for epi in ("bias", "bias_resid"):
                o = ops.gemm(da, dw, db, dr if epi == "bias_resid" else None, epi=epi, variant=f"{tile}/k{ks}")
                assert torch.allclose(o.cpu(), _gemm_ref(a, w, b, r, epi, 1.0, 0), atol=3e-5, rtol=1e-5), (tile, ks, epi)
                assert torch.equal(o, ops.gemm(da, dw, db, dr if epi == "bias_resid" else None, epi=epi, variant=f"{tile}/k{ks}"))
                outs[(tile, epi)] = o
        for epi in ("bias", "bias_resid"):
            for tile in ("128x160", "128x96", "64x128", "128x64", "auto"):
                assert torch.equal(outs[(tile, epi)], outs[("128x128", epi)]), (tile, ks, epi)      # tile-independent
        if M > 256:
            half = ops.gemm(da[:M // 2].contiguous(), dw, db, epi="bias", variant=f"auto/k{ks}")
            assert torch.equal(half, outs[("auto", "bias")][:M // 2]), ks                            # batch-independent


def test_gemm_big_tile_splitk_rejects(built_lib, cuda_dev):
    from tokenhmr_amd import ops, _cabi
    a, w = _rand(256, 96, seed=1).to(cuda_dev), _rand(128, 96, seed=2).to(cuda_dev)
    with pytest.raises(_cabi.EngineError):
        ops.gemm(a, w, variant="128x128/k2")          # 96 % 64 != 0


RING16_SHAPES = [(192, 3840, 1280), (384, 3840, 1280), (64, 48, 32), (64, 48, 96), (200, 100, 512), (37, 31, 1024), (130, 1280, 256), (576, 5120, 1280)]


@pytest.mark.parametrize("shape", RING16_SHAPES)
def test_gemm_ring16(built_lib, cuda_dev, shape):
    """Small-M kernel on 16x16x4 MFMA tiles (64 x 48 workgroup tile; the engine's qkv at one and two crops): every epilogue it has
    against fp64, K tiles fewer than the ring is deep, ragged M / N edges (partial 64-row and 48-column tiles, the uneven 4 / 4 / 3 / 3
    split of the copies), determinism and batch-independence of a row's result."""
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3)
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    for epi, kw in (("none", {}), ("bias", {}), ("bias_gelu", {}), ("bias_relu", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        out = ops.gemm(da, dw, None if epi == "none" else db, epi=epi, variant="ring16", **kw)
        ref = _gemm_ref(a, w, b, None, epi, kw.get("qscale", 1.0), kw.get("qcols", 0))
        assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5), (epi, (out.cpu() - ref).abs().max())
        assert torch.equal(out, ops.gemm(da, dw, None if epi == "none" else db, epi=epi, variant="ring16", **kw))
    full = ops.gemm(da, dw, db, epi="bias", variant="ring16")
    for m in (1, M // 2, M - 1):
        if m >= 1:
            assert torch.equal(ops.gemm(da[:m].contiguous(), dw, db, epi="bias", variant="ring16"), full[:m]), m
    close = ops.gemm(da, dw, db, epi="bias", variant="128x160")
    assert torch.allclose(full, close, atol=2e-5, rtol=1e-5)                         # another order of the K sum, same value to fp32 rounding


def _split3_ref(x):
    """numpy restatement of csrc/gemm_split.hip::split3_kernel: h = bf16_rne(x), m = bf16_rne(x - h), l = bf16_rne(x - h - m),
    laid out [R][K/8][3][8] (uint16 bit patterns)."""
    import numpy as np

    def rne(f):
        u = f.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint32)

    def up(b):
        return (b << 16).astype(np.uint32).view(np.float32)

    x = x.numpy().astype(np.float32)
    h = rne(x)
    r1 = (x - up(h)).astype(np.float32)
    m = rne(r1)
    r2 = (r1 - up(m)).astype(np.float32)
    l = rne(r2)
    R, K = x.shape
    out = np.stack([p.reshape(R, K // 8, 8) for p in (h, m, l)], axis=2).astype(np.uint16)         # (R, K/8, 3, 8)
    return out, (up(h).astype(np.float64) + up(m) + up(l))


def test_split3_convert(built_lib, cuda_dev):
    """fp32 -> three bf16 pieces: bit-identical to the numpy restatement, and h + m + l reproduces x to 2^-24 (it is exact unless the
    third piece rounds)."""
    import numpy as np
    from tokenhmr_amd import ops
    x = _rand(200, 256, seed=5) * torch.logspace(-6, 6, 256)[None, :]
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-38, 1e-45, 65504.0])
    got = ops.split3(x.to(cuda_dev)).cpu().numpy().view(np.uint16)
    ref, back = _split3_ref(x)
    assert np.array_equal(got, ref)
    xd = x.numpy().astype(np.float64)
    ok = (np.abs(xd) < 1e38) & (np.abs(xd) > 1e-30)
    assert np.all(np.abs(back - xd)[ok] <= np.abs(xd)[ok] * 2.0 ** -24)


@pytest.mark.parametrize("B", [1, 2, 3, 16, 20, 30])
def test_vit_attention_split3_output(built_lib, cuda_dev, B):
    """Attention with its output written as the proj GEMM's split3 operand (key-split kernel for one and two crops, 64-query kernel up to 10
    and for 17-24 crops, persistent kernel otherwise — with its own vmcnt bookkeeping for 45 instead of 15 stores per item): bit-identical to converting the fp32 output."""
    from tokenhmr_amd import ops
    qkv = _rand(B, 192, 3840, seed=B).to(cuda_dev)
    ref = ops.split3(ops.vit_attention(qkv, variant="keysplit" if B <= 2 else "auto").reshape(B * 192, 1280))
    got = ops.vit_attention_split3(qkv)
    assert torch.equal(got, ref)
    assert torch.equal(got, ops.vit_attention_split3(qkv))


SPLIT3_SHAPES = [(384, 512, 256), (200, 300, 96), (128, 256, 32), (1, 8, 64), (1536, 1280, 1280), (777, 3840, 1280), (260, 1280, 5120)]


@pytest.mark.parametrize("shape", SPLIT3_SHAPES)
def test_gemm_split3(built_lib, cuda_dev, shape):
    """fp32 GEMM on the bf16 matrix pipe (three bf16 pieces per operand, six products, fp32 accumulate) against fp64: every tile
    variant and epilogue, ragged M / N, one K tile, determinism, tile- and batch-independence of a row's result — and its error
    measured beside the exact-fp32 MFMA kernel's on the same operands (it must be of the same size)."""
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=4)
    a[:, ::7] *= 30.0                                                        # outlier channels, as a ViT residual stream has
    da, dw, db, dr = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), r.to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    ref64 = a.double() @ w.double().t()
    bound = (a.double().abs() @ w.double().abs().t())                       # |a| . |w|: what a dot product's rounding error scales with
    e32 = ((ops.gemm(da, dw, variant="128x160").cpu().double() - ref64).abs() / bound).max().item()
    outs = {}
    for variant in ("128x256/w8", "128x256/w4", "128x128/w4", "128x128/w8", "128x128/w4/s3", "128x128/w8/s3", "256x256/w4"):
        o = ops.gemm_split3(sa, sw, variant=variant)
        es = ((o.cpu().double() - ref64).abs() / bound).max().item()
        assert es <= max(2.0 * e32, 2.0 ** -22), (variant, es, e32)
        assert torch.equal(o, ops.gemm_split3(sa, sw, variant=variant))
        outs[variant] = o
        for epi, kw in (("bias", {}), ("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
            o = ops.gemm_split3(sa, sw, db, dr if epi == "bias_resid" else None, epi=epi, variant=variant, **kw)
            ref = _gemm_ref(a, w, b, r, epi, kw.get("qscale", 1.0), kw.get("qcols", 0))
            assert torch.allclose(o.cpu(), ref, atol=2e-4, rtol=1e-5), (variant, epi, (o.cpu() - ref).abs().max())
    # the product kernels (v_mfma_f32_16x16x32_bf16, csrc/gemm_split16.hip): both tiles and the rule give the same bits
    assert torch.equal(outs["128x128/w4"], outs["128x256/w8"]) and torch.equal(ops.gemm_split3(sa, sw, variant="auto"), outs["128x256/w8"])
    # round 6: the 128 x 128 tile on eight waves of 64 x 32, and on four waves with a three-stage K ring
    assert torch.equal(outs["128x128/w8"], outs["128x256/w8"]) and torch.equal(outs["128x128/w4/s3"], outs["128x256/w8"])
    assert torch.equal(outs["128x128/w8/s3"], outs["128x256/w8"])
    for epi, kw in (("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        rr = dr if epi == "bias_resid" else None
        for v in ("128x128/w8", "128x128/w4/s3", "128x128/w8/s3", "128x256/w8/front", "128x128/w4/s3/front"):
            assert torch.equal(ops.gemm_split3(sa, sw, db, rr, epi=epi, variant=v, **kw), ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="128x256/w8", **kw)), (v, epi)
    # the round-3 / first round-4 kernels on 32x32x16 MFMAs (experiments build): bit-identical among themselves — 64x64 and 64x128 wave
    # tiles, the 256x256 tile, the small-M ring kernel without split-K — and equal to the product kernels to fp32 rounding (another
    # grouping of k inside the MFMA)
    old = ops.gemm_split3(sa, sw, variant="old/128x256/w8")
    ring = ops.gemm_split3(sa, sw, variant="ring")
    assert torch.equal(outs["128x256/w4"], old) and torch.equal(outs["256x256/w4"], old) and torch.equal(ring, old)
    assert torch.equal(ops.gemm_split3(sa, sw, variant="old/128x128/w4"), old)
    assert (((old.cpu().double() - ref64).abs() / bound).max().item()) <= max(2.0 * e32, 2.0 ** -22)
    assert torch.allclose(old, outs["128x256/w8"], atol=2e-4, rtol=1e-5)
    for epi, kw in (("bias", {}), ("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        o = ops.gemm_split3(sa, sw, db, dr if epi == "bias_resid" else None, epi=epi, variant="ring", **kw)
        assert torch.equal(o, ops.gemm_split3(sa, sw, db, dr if epi == "bias_resid" else None, epi=epi, variant="old/128x256/w8", **kw)), epi
    for ks, name in ((2, "ring/k2"), (4, "ring/k4"), (2, "auto/k2"), (4, "auto/k4")):
        if K % (32 * ks):
            continue
        o = ops.gemm_split3(sa, sw, db, dr, epi="bias_resid", variant=name)
        assert torch.allclose(o.cpu(), _gemm_ref(a, w, b, r, "bias_resid", 1.0, 0), atol=2e-4, rtol=1e-5), name
        assert torch.equal(o, ops.gemm_split3(sa, sw, db, dr, epi="bias_resid", variant=name))
        if M > 2:
            half = ops.gemm_split3(ops.split3(da[:M // 2].contiguous()), sw, db, dr[:M // 2].contiguous(), epi="bias_resid", variant=name)
            assert torch.equal(half, o[:M // 2]), name
    if N % 8 == 0:      # the epilogue's result as the next GEMM's split3 operand: bit-identical to converting the fp32 result
        for variant in ("128x256/w8", "128x256/w4", "128x128/w4", "128x128/w8", "128x128/w4/s3", "256x256/w4", "ring", "old/128x256/w8"):
            for epi, kw in (("none", {}), ("bias", {}), ("bias_gelu", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
                bb = None if epi == "none" else db
                fused = ops.gemm_split3(sa, sw, bb, epi=epi, variant=variant, out_split=True, **kw)
                assert torch.equal(fused, ops.split3(ops.gemm_split3(sa, sw, bb, epi=epi, variant=variant, **kw))), (variant, epi)
    if M > 2:
        half = ops.gemm_split3(ops.split3(da[:M // 2].contiguous()), sw)
        assert torch.equal(half, outs["128x256/w8"][:M // 2])


# (M, N, K): 256 tiles exactly (every stream lane 8 tiles) / ragged lists (272 tiles: lanes 0-15 hold 9, the rest 8) with an odd number
# of K tiles / the four ViT GEMM shapes of a 64-crop batch (1440, 480, 1920, 480 tiles = 5.625 / 1.875 / 7.5 / 1.875 per CU)
PERSIST_SHAPES = [(2048, 4096, 320), (2176, 4096, 352), (12288, 3840, 1280), (12288, 1280, 1280), (12288, 5120, 1280), (12288, 1280, 5120),
                  (6720, 1280, 5120), (4416, 3840, 1280)]      # round 6, ragged M: fc2 at 35 crops (52.5 row tiles), qkv at 23


@pytest.mark.parametrize("shape", PERSIST_SHAPES)
def test_gemm_split3_persistent(built_lib, cuda_dev, shape):
    """The persistent split3 GEMM (256 workgroups over a tile stream, a ragged last round split along K with the accumulators handed
    from one XCD's workgroup to the next through memory) is BIT-IDENTICAL to the one-workgroup-per-tile kernel — the accumulator chain of
    every element is the same, only who runs which part of it changes — for every epilogue, for the split3 output in both forms (LDS
    transposition / swapped operand roles + permlane swaps), repeatedly (the hand-over flags re-arm themselves), and interleaved with
    other shapes on the same workspace."""
    import torch
    from tokenhmr_amd import ops
    if torch.cuda.get_device_properties(cuda_dev).multi_processor_count != 256:
        pytest.skip("the persistent decomposition is 8 XCDs x 32 CUs")
    M, N, K = shape
    a, w, b = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=1 / math.sqrt(K)), _rand(N, seed=13)
    a[:, ::7] *= 30.0
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    dr = _rand(M, N, seed=14).to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    base = ops.gemm_split3(sa, sw, variant="128x256/w8")
    ref64 = a[:256].double() @ w.double().t()                              # the kernel itself against fp64 on a slice (the rest by identity)
    bound = a[:256].double().abs() @ w.double().abs().t()
    assert ((base[:256].cpu().double() - ref64).abs() / bound).max().item() < 2.0 ** -19      # measured ~1e-6 (test_gemm_split3 compares with the fp32 kernel)
    for rep in range(3):
        o = ops.gemm_split3(sa, sw, variant="persist")
        assert torch.equal(o, base), (rep, int((o != base).sum()), (o - base).abs().max().item())
    for epi, kw in (("bias", {}), ("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        rr = dr if epi == "bias_resid" else None
        want = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="128x256/w8", **kw)
        got = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="persist", **kw)
        assert torch.equal(got, want), (epi, int((got != want).sum()))
    for epi in ("none", "bias", "bias_gelu", "bias_qscale"):
        bb = None if epi == "none" else db
        kw = dict(qscale=80 ** -0.5, qcols=N // 3) if epi == "bias_qscale" else {}
        want = ops.gemm_split3(sa, sw, bb, epi=epi, variant="128x256/w8", out_split=True, **kw)
        got = ops.gemm_split3(sa, sw, bb, epi=epi, variant="persist/swap", out_split=True, **kw)
        assert torch.equal(got, want), (epi, int((got != want).sum()))
    # the first round-4 persistent kernel (32x32x16 MFMAs; experiments build) against ITS per-tile twin: fp32 output, and the split3 output
    # through the LDS transposition / swapped roles + permlane32 swaps
    if M % 128 == 0:                                        # (the 32x32x16 kernels take whole row tiles only)
        old = ops.gemm_split3(sa, sw, variant="old/128x256/w8")
        assert torch.equal(ops.gemm_split3(sa, sw, variant="old/persist"), old) and torch.allclose(old, base, atol=2e-4, rtol=1e-5)
        want = ops.gemm_split3(sa, sw, db, epi="bias_gelu", variant="old/128x256/w8", out_split=True)
        for variant in ("old/persist/lds", "old/persist/swap"):
            assert torch.equal(ops.gemm_split3(sa, sw, db, epi="bias_gelu", variant=variant, out_split=True), want), variant
    # another shape on the same (device, stream) workspace in between, then this one again
    a2, w2 = _rand(2048, 64, seed=21).to(cuda_dev), _rand(4096, 64, seed=22).to(cuda_dev)
    s2a, s2w = ops.split3(a2), ops.split3(w2)
    assert torch.equal(ops.gemm_split3(s2a, s2w, variant="persist"), ops.gemm_split3(s2a, s2w, variant="128x256/w8"))
    assert torch.equal(ops.gemm_split3(sa, sw, variant="persist"), base)


# few-crop shapes whose 128 x 128 grid is more than one round of 256 workgroups: qkv / fc1 at 6, 8 and 16 crops, an odd number of K tiles
# (phases of the three-stage ring carried across segment boundaries), the smallest stream (256 tiles: every lane's ranges exactly one tile)
PERSIST_NARROW_SHAPES = [(1152, 3840, 1280), (1536, 3840, 1280), (1536, 5120, 1280), (3072, 3840, 1280), (2048, 2048, 352), (2176, 4096, 96),
                         (1344, 3840, 1280), (960, 5120, 1280), (2496, 3840, 160)]      # ragged M: 7 / 5 / 13 crops (192 B rows: the last row tile half full)


@pytest.mark.parametrize("shape", PERSIST_NARROW_SHAPES)
def test_gemm_split3_persistent_narrow(built_lib, cuda_dev, shape):
    """Round 6: 256 persistent workgroups over the stream of 128 x 128 tiles with the three-stage K ring — a tile's K
    range shared by two workgroups at every range boundary, the raw accumulators handed over through the slab, the ring's phase carried
    across segments — is BIT-IDENTICAL to one workgroup per tile: every epilogue, the split3 output (row-major and row-blocked),
    repeatedly, and alternating with the 128 x 256 stream on the same workspace."""
    import torch
    from tokenhmr_amd import ops
    if torch.cuda.get_device_properties(cuda_dev).multi_processor_count != 256:
        pytest.skip("the persistent decomposition is 8 XCDs x 32 CUs")
    M, N, K = shape
    a, w, b = _rand(M, K, seed=51), _rand(N, K, seed=52, scale=1 / math.sqrt(K)), _rand(N, seed=53)
    a[:, ::7] *= 30.0
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    dr = _rand(M, N, seed=54).to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    base = ops.gemm_split3(sa, sw, variant="128x256/w8")
    for rep in range(3):
        o = ops.gemm_split3(sa, sw, variant="persist/128x128")
        assert torch.equal(o, base), (rep, int((o != base).sum()), (o - base).abs().max().item())
    for epi, kw in (("bias", {}), ("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        rr = dr if epi == "bias_resid" else None
        want = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="128x256/w8", **kw)
        got = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="persist/128x128", **kw)
        assert torch.equal(got, want), (epi, int((got != want).sum()))
    for epi in ("none", "bias_gelu"):
        bb = None if epi == "none" else db
        for blocked in (False, True):
            want = ops.gemm_split3(sa, sw, bb, epi=epi, variant="128x256/w8", out_split=True, out_blocked=blocked)
            got = ops.gemm_split3(sa, sw, bb, epi=epi, variant="persist/128x128", out_split=True, out_blocked=blocked)
            assert torch.equal(got, want), (epi, blocked, int((got != want).sum()))
    if M % 128 == 0 and M * N // (128 * 256) >= 256 and N % 256 == 0:
        assert torch.equal(ops.gemm_split3(sa, sw, variant="persist"), base)          # the 128 x 256 stream on the same workspace in between
        assert torch.equal(ops.gemm_split3(sa, sw, variant="persist/128x128"), base)
    with pytest.raises(Exception):
        ops.gemm_split3(ops.split3(da[:640].contiguous()), sw[:1280].contiguous(), variant="persist/128x128")      # 5 x 10 tiles: fewer than 256


# split-K launches whose (tile, K slice) units run as the 128 x 128 stream: fc2 at 9 / 18 crops and proj at 10 crops two ways, a four-way case
SPLITK_STREAM_SHAPES = [(1728, 1280, 5120, 2), (3456, 1280, 5120, 2), (1920, 1280, 1280, 2), (1152, 1280, 5120, 4)]


@pytest.mark.parametrize("shape", SPLITK_STREAM_SHAPES)
def test_gemm_split3_splitk_through_the_stream(built_lib, cuda_dev, shape):
    """Round 6: a split-K launch as (tile, K slice) units of the persistent 128 x 128 stream writes the SAME raw partial sums as the
    grid copies of the per-tile kernel — the reduced result (fixed-order reduce + bias + residual) is bit-identical — repeatedly, ragged M
    included; a launch with fewer than 256 units is refused."""
    import torch
    from tokenhmr_amd import ops
    if torch.cuda.get_device_properties(cuda_dev).multi_processor_count != 256:
        pytest.skip("the persistent decomposition is 8 XCDs x 32 CUs")
    M, N, K, ks = shape
    a, w, b = _rand(M, K, seed=61), _rand(N, K, seed=62, scale=1 / math.sqrt(K)), _rand(N, seed=63)
    a[:, ::7] *= 30.0
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    dr = _rand(M, N, seed=64).to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    want = ops.gemm_split3(sa, sw, db, dr, epi="bias_resid", variant=f"auto/k{ks}")
    for rep in range(3):
        got = ops.gemm_split3(sa, sw, db, dr, epi="bias_resid", variant=f"persist/128x128/k{ks}")
        assert torch.equal(got, want), (rep, int((got != want).sum()), (got - want).abs().max().item())
    assert torch.allclose(want.cpu(), _gemm_ref(a, w, b, dr.cpu(), "bias_resid", 1.0, 0), atol=2e-4, rtol=1e-5)
    with pytest.raises(Exception):
        ops.gemm_split3(ops.split3(da[:512].contiguous()), sw, db, dr[:512].contiguous(), epi="bias_resid", variant=f"persist/128x128/k{ks}")      # 4 x 10 x ks units


# (M, N, K) whose 128 x 256 tile count T divides by 8 and leaves, per XCD (q = T / 8 tiles on 32 CUs), 1 ... 16 tiles after the full rounds:
# fc1 of a 64-crop batch (q = 240 = 7 x 32 + 16), qkv / proj of 48 crops (q = 135 = 4 x 32 + 7; q = 45 = 32 + 13), a ragged-M case with 3 K tiles
TAIL_SHAPES = [(12288, 5120, 1280), (9216, 3840, 1280), (9216, 1280, 1280), (12238, 5120, 96)]


@pytest.mark.parametrize("shape", TAIL_SHAPES)
def test_gemm_split3_half_tile_tail(built_lib, cuda_dev, shape):
    """Round 5: the per-tile split3 GEMM whose ragged last round runs as 128 x 128 half tiles on 4 of the workgroup's 8
    waves (csrc/gemm_split16.hip gemm_split16_tail_kernel) is BIT-IDENTICAL to the plain 128 x 256 grid — every element's K sum is the
    same chain, only which workgroup shape computes it changes — for every epilogue, the split3 output (row-major and row-blocked), ragged
    M, repeatedly; the rule ("auto") picks it for these shapes; a shape that does not qualify is refused when asked for by name."""
    import torch
    from tokenhmr_amd import ops, _cabi
    if torch.cuda.get_device_properties(cuda_dev).multi_processor_count != 256:
        pytest.skip("the tail split is laid out for 8 XCDs x 32 CUs")
    M, N, K = shape
    a, w, b = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=1 / math.sqrt(K)), _rand(N, seed=33)
    a[:, ::7] *= 30.0
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    dr = _rand(M, N, seed=34).to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    base = ops.gemm_split3(sa, sw, variant="128x256/w8")
    for rep in range(2):
        for tail in ("tail", "tail/w8"):               # the half tiles on four waves of 64 x 64 (round 5, the rule's form) / on eight waves of 64 x 32 (round 6)
            o = ops.gemm_split3(sa, sw, variant=tail)
            assert torch.equal(o, base), (tail, rep, int((o != base).sum()), (o - base).abs().max().item())
    assert torch.equal(ops.gemm_split3(sa, sw, variant="auto"), base)
    for epi, kw in (("bias", {}), ("bias_gelu", {}), ("bias_resid", {}), ("bias_qscale", dict(qscale=80 ** -0.5, qcols=N // 3))):
        rr = dr if epi == "bias_resid" else None
        want = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="128x256/w8", **kw)
        got = ops.gemm_split3(sa, sw, db, rr, epi=epi, variant="tail", **kw)
        assert torch.equal(got, want), (epi, int((got != want).sum()))
    for epi in ("none", "bias_gelu"):
        bb = None if epi == "none" else db
        for blocked in (False, True):
            want = ops.gemm_split3(sa, sw, bb, epi=epi, variant="128x256/w8", out_split=True, out_blocked=blocked)
            got = ops.gemm_split3(sa, sw, bb, epi=epi, variant="tail", out_split=True, out_blocked=blocked)
            assert torch.equal(got, want), (epi, blocked, int((got != want).sum()))
    # does not qualify: 96 x 15 = 1440 tiles, q = 180 = 5 x 32 + 20 (qkv of a 64-crop batch): refused by name, plain grid by rule
    small_a, small_w = ops.split3(_rand(12288, 64, seed=41).to(cuda_dev)), ops.split3(_rand(3840, 64, seed=42).to(cuda_dev))
    with pytest.raises(_cabi.EngineError):
        ops.gemm_split3(small_a, small_w, variant="tail")
    assert torch.equal(ops.gemm_split3(small_a, small_w, variant="auto"), ops.gemm_split3(small_a, small_w, variant="128x256/w8"))


@pytest.mark.parametrize("M", [192 * 3, 192 * 64, 700])
def test_gemm_split3_pos_embed_epilogue(built_lib, cuda_dev, M):
    """The patch-embed epilogue ((acc + bias) + pos[1 + row % 192]) + pos[0] (vit.py:327) on the split3 GEMM: equal to the exact-fp32 GEMM's
    to the mode's rounding class, both tiles bit-identical, ragged M."""
    from tokenhmr_amd import ops
    N, K = 1280, 768
    a, w, b, pos = _rand(M, K, seed=51), _rand(N, K, seed=52, scale=1 / math.sqrt(K)), _rand(N, seed=53), _rand(193, N, seed=54)
    da, dw, db, dp = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), pos.to(cuda_dev)
    ref = ops.gemm(da, dw, db, dp, epi="bias_pos")
    sa, sw = ops.split3(da), ops.split3(dw)
    o0 = ops.gemm_split3(sa, sw, db, dp, epi="bias_pos", variant="128x256/w8")
    o2 = ops.gemm_split3(sa, sw, db, dp, epi="bias_pos", variant="128x128/w4")
    assert torch.equal(o0, o2) and torch.equal(o0, ops.gemm_split3(sa, sw, db, dp, epi="bias_pos", variant="auto"))
    assert torch.allclose(o0, ref, atol=2e-5, rtol=1e-5), (o0 - ref).abs().max().item()
    want = (a.double() @ w.double().t() + b.double()[None]) + pos[1:].double()[torch.arange(M) % 192] + pos[0].double()[None]
    assert (o0.cpu().double() - want).abs().max().item() < 3e-5


@pytest.mark.parametrize("shape", [(384, 512, 256), (200, 512, 96), (777, 1280, 1280), (2048, 4096, 320), (12288, 1280, 5120)])
def test_gemm_split3_row_blocked_operand(built_lib, cuda_dev, shape):
    """The row-blocked split3 operand ([R/32][K/8][3][32][8]: what the engine's fc1 hands fc2) — written by both split3-output epilogues
    (LDS transposition of the per-tile kernel, swapped roles of the persistent one) and read as A by the per-tile kernel (both tiles,
    split-K) and the persistent kernel: the same bits as the row-major form, ragged M included."""
    import torch
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=1 / math.sqrt(K)), _rand(N, seed=33)
    a[:, ::5] *= 20.0
    da, dw, db = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev)
    dr = _rand(M, N, seed=34).to(cuda_dev)
    sa, sw = ops.split3(da), ops.split3(dw)
    sab = ops.split3_block(sa)
    assert torch.equal(ops.split3_unblock(sab, M), sa)
    persist = M % 32 == 0 and N % 256 == 0 and ((M + 127) // 128) * (N // 256) >= 256 and torch.cuda.get_device_properties(cuda_dev).multi_processor_count == 256
    # A in the blocked form
    for epi in ("none", "bias_resid"):
        bb, rr = (None, None) if epi == "none" else (db, dr)
        want = ops.gemm_split3(sa, sw, bb, rr, epi=epi, variant="128x256/w8")
        for v in ("128x256/w8", "128x128/w4") + (("persist",) if persist else ()):
            got = ops.gemm_split3(sab, sw, bb, rr, epi=epi, variant=v, a_blocked_rows=M)
            assert torch.equal(got, want), (v, epi, int((got != want).sum()))
    if K % 64 == 0:
        want = ops.gemm_split3(sa, sw, db, dr, epi="bias_resid", variant="auto/k2")
        assert torch.equal(ops.gemm_split3(sab, sw, db, dr, epi="bias_resid", variant="auto/k2", a_blocked_rows=M), want)
    # the result in the blocked form
    if N % 8 == 0:
        for epi in ("none", "bias_gelu"):
            bb = None if epi == "none" else db
            want = ops.gemm_split3(sa, sw, bb, epi=epi, variant="128x256/w8", out_split=True)
            for v in ("128x256/w8", "128x128/w4") + (("persist/swap",) if persist else ()):
                got = ops.gemm_split3(sa, sw, bb, epi=epi, variant=v, out_split=True, out_blocked=True)
                assert torch.equal(ops.split3_unblock(got, M), want), (v, epi)
    # and chained as the engine does: blocked out of one product, blocked A of the next
    if N % 32 == 0 and N <= 4096:
        w2 = _rand(256, N, seed=35, scale=1 / math.sqrt(N)).to(cuda_dev)
        s2 = ops.split3(w2)
        hid_b = ops.gemm_split3(sa, sw, db, epi="bias_gelu", variant="128x256/w8", out_split=True, out_blocked=True)
        hid = ops.gemm_split3(sa, sw, db, epi="bias_gelu", variant="128x256/w8", out_split=True)
        assert torch.equal(ops.gemm_split3(hid_b, s2, variant="128x256/w8", a_blocked_rows=M), ops.gemm_split3(hid, s2, variant="128x256/w8"))


def test_gemm_split3_persistent_rejects(built_lib, cuda_dev):
    from tokenhmr_amd import ops, _cabi
    sa, sw = ops.split3(_rand(2048, 64, seed=1).to(cuda_dev)), ops.split3(_rand(3840, 64, seed=2).to(cuda_dev))
    with pytest.raises(_cabi.EngineError):
        ops.gemm_split3(sa, sw, variant="persist")             # 16 x 15 = 240 tiles < 256
    sa2 = ops.split3(_rand(2048 + 24, 64, seed=1).to(cuda_dev))
    with pytest.raises(_cabi.EngineError):
        ops.gemm_split3(sa2, ops.split3(_rand(4096, 64, seed=2).to(cuda_dev)), variant="persist")     # M % 32 != 0 (round 6: a ragged last row tile of whole 32-row blocks is served)


TINY_SHAPES = [(21, 512, 1536), (160, 512, 768), (160, 256, 2048), (126, 6, 1536), (960, 512, 1536), (55, 512, 512), (37, 31, 256)]


@pytest.mark.parametrize("shape", TINY_SHAPES)
def test_gemm_tiny_variant(built_lib, cuda_dev, shape):
    """The tiny-M kernel of the head's small-batch regime (32x32 tiles, K split over the 8 waves of a workgroup, partial tiles added
    in wave order): every epilogue it serves against fp64, ragged M / N edges (21 rows, 6 columns), determinism, and — the property
    the regime relies on — a row's result does not depend on how many rows share the launch."""
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=4)
    da, dw, db, dr = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), r.to(cuda_dev)
    for epi in ("none", "bias", "bias_relu", "bias_resid", "bias_gelu"):
        out = ops.gemm(da, dw, None if epi == "none" else db, dr if epi == "bias_resid" else None, epi=epi, variant="tiny")
        ref = _gemm_ref(a, w, b, r, epi, 1.0, 0)
        assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5), (epi, (out.cpu() - ref).abs().max())
        assert torch.equal(out, ops.gemm(da, dw, None if epi == "none" else db, dr if epi == "bias_resid" else None, epi=epi, variant="tiny"))
    full = ops.gemm(da, dw, db, epi="bias", variant="tiny")
    for m in (1, 21, min(M, 100)):
        assert torch.equal(ops.gemm(da[:m].contiguous(), dw, db, epi="bias", variant="tiny"), full[:m]), m


def test_gemm_tiny_rejects(built_lib, cuda_dev):
    from tokenhmr_amd import ops, _cabi
    a, w = _rand(64, 384, seed=1).to(cuda_dev), _rand(64, 384, seed=2).to(cuda_dev)
    with pytest.raises(_cabi.EngineError):
        ops.gemm(a, w, variant="tiny")                # K must be a multiple of 256 (8 waves x 32-deep slices)


def test_gemm_ring_rejects(built_lib, cuda_dev):
    from tokenhmr_amd import ops, _cabi
    a, w = _rand(64, 96, seed=1).to(cuda_dev), _rand(64, 96, seed=2).to(cuda_dev)
    with pytest.raises(_cabi.EngineError):
        ops.gemm(a, w, variant="ring4/k2")            # 96 % 64 != 0


def test_gelu_epilogue_ulp(built_lib, cuda_dev):
    """The epilogue's branch-free erf (csrc/common.h erf_gelu; erf_fast in the THMR_GELU_IMPL=1/2 builds) against fp64 GELU on a dense grid.  A has x in column 0 and
    W a single 1, so C[m, n] = gelu(x_m) with no accumulation error.  fp32 GELU itself loses bits to the 1+erf cancellation
    for x << 0 (torch's fp32 kernel does too), hence the |x|-scaled absolute term: 2 ulp of (1+erf) * |x|/2."""
    from tokenhmr_amd import ops
    M, N, K = 8192, 32, 32
    x = torch.cat([torch.linspace(-9, 9, M - 512, dtype=torch.float64),
                   torch.linspace(-1.4, -1.2, 256, dtype=torch.float64),      # around the piece boundary |x|/sqrt2 = 0.921875
                   torch.linspace(1.2, 1.4, 256, dtype=torch.float64)]).float()
    a = torch.zeros(M, K)
    a[:, 0] = x
    w = torch.zeros(N, K)
    w[:, 0] = 1.0
    out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), torch.zeros(N).to(cuda_dev), epi="bias_gelu", variant="128x160").cpu()
    ref = F.gelu(x.double())
    err = (out[:, 0].double() - ref).abs()
    tol = 1.2e-7 * x.abs().clamp(min=1.0).double() + 2e-7 * ref.abs()
    assert bool((err <= tol).all()), (err / tol).max()
    assert torch.equal(out[:, 0], out[:, N - 1])
    # and it is no worse than torch's own fp32 GELU (ocml/MKL erff) by more than 1 ulp-class
    t32 = (F.gelu(x).double() - ref).abs().max()
    assert err.max() <= t32 + 2.5e-7, (err.max(), t32)


def test_gemm_asymmetric_identity(built_lib, cuda_dev):
    """A = I with an asymmetric W catches a transposed C write (guide rule: always A=I with asymmetric B)."""
    from tokenhmr_amd import ops
    n = 256
    a = torch.eye(n)
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), epi="none").cpu()
    assert torch.equal(out, w.t().contiguous())


@pytest.mark.parametrize("shape", [(64, 1024, 1024), (64, 512, 1024), (5, 31, 1024), (64, 10240, 1024), (100, 48, 64), (2, 1024, 512)])
@pytest.mark.parametrize("epi", ["none", "bias", "bias_gelu", "bias_resid"])
def test_gemm_skinny(built_lib, cuda_dev, shape, epi):
    from tokenhmr_amd import ops
    M, N, K = shape
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3), _rand(M, N, seed=4)
    out = ops.gemm(a.to(cuda_dev), w.to(cuda_dev), None if epi == "none" else b.to(cuda_dev),
                   r.to(cuda_dev) if epi == "bias_resid" else None, epi=epi, variant="skinny").cpu()
    ref = _gemm_ref(a, w, b, r, epi, 1.0, 0)
    assert torch.allclose(out, ref, atol=3e-5, rtol=1e-5), (out - ref).abs().max()


def test_gemm_deterministic(built_lib, cuda_dev):
    from tokenhmr_amd import ops
    a, w = _rand(384, 1280, seed=1).to(cuda_dev), _rand(1280, 1280, seed=2, scale=0.03).to(cuda_dev)
    assert torch.equal(ops.gemm(a, w), ops.gemm(a, w))


def test_gemm_result_does_not_depend_on_the_tile(built_lib, cuda_dev):
    """Every LDS-DMA tile (and the auto-selector, whatever it picks) sums K in the same order: the tile is a scheduling choice only.
    The engine relies on it — a crop's bits must not change with the tile its batch size selects (7-8 crops: 64x128 on the N = 1280
    GEMMs, 17+ crops: 128x128 / 128x160)."""
    from tokenhmr_amd import ops
    for (M, N, K), epi in (((1536, 1280, 5120), "bias_resid"), ((1344, 1280, 1280), "bias_resid"), ((1536, 3840, 1280), "bias_qscale"),
                           ((530, 200, 96), "bias_gelu")):
        a, w, b, r = _rand(M, K, seed=1).to(cuda_dev), _rand(N, K, seed=2, scale=1 / math.sqrt(K)).to(cuda_dev), _rand(N, seed=3).to(cuda_dev), _rand(M, N, seed=4).to(cuda_dev)
        kw = dict(epi=epi, qscale=80 ** -0.5, qcols=N // 3) if epi == "bias_qscale" else dict(epi=epi)
        ref = ops.gemm(a, w, b, r if epi == "bias_resid" else None, variant="128x160", **kw)
        for v in ("128x128", "128x96", "64x64", "64x128", "128x64", "auto"):
            assert torch.equal(ops.gemm(a, w, b, r if epi == "bias_resid" else None, variant=v, **kw), ref), (M, N, K, v)


@pytest.mark.parametrize("rows,D,eps,relu", [(384, 1280, 1e-6, False), (7, 1280, 1e-6, False), (64, 1024, 1e-5, False),
                                              (2, 10240, 1e-5, True), (320, 64, 1e-5, True), (3, 64, 1e-5, False)])
def test_layernorm(built_lib, cuda_dev, rows, D, eps, relu):
    from tokenhmr_amd import ops
    x = _rand(rows, D, seed=1, scale=3.0) + 0.7
    g, b = 1 + 0.1 * _rand(D, seed=2), 0.1 * _rand(D, seed=3)
    out = ops.layernorm(x.to(cuda_dev), g.to(cuda_dev), b.to(cuda_dev), eps, relu).cpu()
    ref = F.layer_norm(x, (D,), g, b, eps)
    if relu:
        ref = F.relu(ref)
    assert torch.allclose(out, ref, atol=5e-6, rtol=1e-5), (out - ref).abs().max()


@pytest.mark.parametrize("B", [1, 3])
def test_vit_attention(built_lib, cuda_dev, B):
    """vit.py:113-122 on random q,k,v (q pre-scaled as the QKV epilogue would)."""
    from tokenhmr_amd import ops
    qkv = _rand(B, 192, 3840, seed=11)
    qkv[:, :, :1280] *= 80 ** -0.5
    out = ops.vit_attention(qkv.to(cuda_dev)).cpu()
    t = qkv.reshape(B, 192, 3, 16, 80).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    ref = ((q @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, 192, 1280)
    assert torch.allclose(out, ref, atol=5e-6, rtol=1e-5), (out - ref).abs().max()


@pytest.mark.parametrize("B", [1, 2, 6])
def test_vit_attention_keysplit(built_lib, cuda_dev, B):
    """The small-batch regime's attention (16 queries per workgroup, the 192 keys split over its 4 waves, per-wave softmaxes merged
    the flash-attention way): against fp64 with the error class of torch's own fp32, against the 64-query kernel to fp32 rounding,
    deterministic, batch-independent, and finite on peaked scores (one dominant key per query, |scores| ~ 100: a wave whose 48 keys
    are all far below the row maximum contributes 2^(-large) without producing inf / nan)."""
    from tokenhmr_amd import ops
    for scale, tol in ((1.0, 5e-6), (6.0, None)):
        qkv = _rand(B, 192, 3840, seed=21 + B)
        qkv[:, :, :1280] *= 80 ** -0.5
        qkv[:, :, :2560] *= scale
        d = qkv.to(cuda_dev)
        out = ops.vit_attention(d, variant="keysplit")
        assert torch.equal(out, ops.vit_attention(d, variant="keysplit"))
        assert torch.equal(ops.vit_attention(d[:1].contiguous(), variant="keysplit"), out[:1])
        for v in ("keysplit/q16", "keysplit/q32", "keysplit/q48"):                  # queries per workgroup: a scheduling choice only
            assert torch.equal(ops.vit_attention(d, variant=v), out), v
        t = qkv.reshape(B, 192, 3, 16, 80).permute(2, 0, 3, 1, 4)
        ref32 = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(B, 192, 1280)
        t64 = t.double()
        ref64 = ((t64[0] @ t64[1].transpose(-2, -1)).softmax(-1) @ t64[2]).transpose(1, 2).reshape(B, 192, 1280)
        assert torch.isfinite(out).all()
        err_hip = (out.cpu().double() - ref64).abs().max().item()
        err_cpu = (ref32.double() - ref64).abs().max().item()
        assert err_hip <= max(4 * err_cpu, 2e-5), (scale, err_hip, err_cpu)
        other = ops.vit_attention(d, variant="q64")
        if tol is not None:
            assert torch.allclose(out, other, atol=tol, rtol=1e-5), (out - other).abs().max()


def test_vit_attention_peaked(built_lib, cuda_dev):
    """Large score spread (|scores| ~ 100, one dominant key per query) exercises the max-subtraction path.
    Judged against an fp64 statement: the HIP fp32 error must be of the same class as torch's own fp32 error
    (both are dominated by the fp32 rounding of ~1e2-sized scores entering exp)."""
    from tokenhmr_amd import ops
    qkv = _rand(1, 192, 3840, seed=12)
    qkv[:, :, :2560] *= 6.0
    out = ops.vit_attention(qkv.to(cuda_dev)).cpu()
    t = qkv.reshape(1, 192, 3, 16, 80).permute(2, 0, 3, 1, 4)
    ref32 = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(1, 192, 1280)
    t64 = t.double()
    ref64 = ((t64[0] @ t64[1].transpose(-2, -1)).softmax(-1) @ t64[2]).transpose(1, 2).reshape(1, 192, 1280)
    assert torch.isfinite(out).all()
    err_hip = (out.double() - ref64).abs().max().item()
    err_cpu = (ref32.double() - ref64).abs().max().item()
    assert err_hip <= max(4 * err_cpu, 2e-5), (err_hip, err_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [11, 16, 28, 32, 33, 40, 100])
def test_vit_attention_persistent_matches_the_64_query_variant(built_lib, cuda_dev, B):
    """Batches of 11-16 and from 25 crops on run on the persistent kernel (a different LDS image: 84-float rows filled three rows per
    copy; up to 32 crops one item per workgroup, above that 512 workgroups walking 2 ... 4 items each with the next item's K fetched
    under the current P.V).  Every query must come out bit for bit as from the 64-query variant that serves B <= 10 — B = 33 gives a
    grid where only some workgroups have a second item, B = 100 up to four items per workgroup — and repeat runs must agree (missed
    wait / barrier)."""
    from tokenhmr_amd import ops
    qkv = _rand(B, 192, 3840, seed=100 + B)
    qkv[:, :, :1280] *= 80 ** -0.5
    d = qkv.to(cuda_dev)
    out = ops.vit_attention(d)
    for s0 in range(0, B, 10):
        assert torch.equal(ops.vit_attention(d[s0:s0 + 10].contiguous()), out[s0:s0 + 10]), s0
    for _ in range(5):
        assert torch.equal(ops.vit_attention(d), out)
    t = qkv[-2:].reshape(2, 192, 3, 16, 80).permute(2, 0, 3, 1, 4).double()
    ref = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(2, 192, 1280)
    assert (out[-2:].cpu().double() - ref).abs().max() < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 11, 33, 40, 100])
def test_vit_attention_b16(built_lib, cuda_dev, B):
    """The attention on the bf16 matrix pipe (csrc/attention_b16.hip: q, k, v and the un-normalised probabilities as three bf16 pieces, six
    products per pair, fp32 accumulate; three 64-key blocks combined with a running row maximum) against fp64 with the error class of
    torch's own fp32 and of the fp32-MFMA kernel; the two workgroup shapes (64 / 192 queries) bit-identical; deterministic; batch-independent;
    the split3 output = the conversion of the fp32 output; finite and in class on peaked scores (|scores| ~ 100).  At most 512 workgroups
    walk the items with the block pipeline running across item boundaries: B = 33 gives a grid where only some workgroups have a second
    item, B = 100 three or four items per workgroup."""
    from tokenhmr_amd import ops
    for scale in (1.0, 6.0):
        qkv = _rand(B, 192, 3840, seed=300 + B)
        qkv[:, :, :1280] *= 80 ** -0.5
        qkv[:, :, :2560] *= scale
        d = qkv.to(cuda_dev)
        out = ops.vit_attention_b16(d)
        assert torch.isfinite(out).all()
        for qt in (1, 3):
            assert torch.equal(ops.vit_attention_b16(d, qt=qt), out), qt
        for _ in range(3):
            assert torch.equal(ops.vit_attention_b16(d), out)
        assert torch.equal(ops.vit_attention_b16(d[-1:].contiguous()), out[-1:])
        if B > 32:
            assert torch.equal(ops.vit_attention_b16(d[:32].contiguous()), out[:32])          # one item per workgroup vs several
        assert torch.equal(ops.vit_attention_b16(d, out_split=True), ops.split3(out.reshape(B * 192, 1280)))
        assert torch.equal(ops.vit_attention_b16(d, out_split=True, qt=1), ops.vit_attention_b16(d, out_split=True, qt=3))
        n = min(B, 4)
        t = qkv[:n].reshape(n, 192, 3, 16, 80).permute(2, 0, 3, 1, 4)
        ref32 = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(n, 192, 1280)
        t64 = t.double()
        ref64 = ((t64[0] @ t64[1].transpose(-2, -1)).softmax(-1) @ t64[2]).transpose(1, 2).reshape(n, 192, 1280)
        err_hip = (out[:n].cpu().double() - ref64).abs().max().item()
        err_cpu = (ref32.double() - ref64).abs().max().item()
        err_f32k = (ops.vit_attention(d[:n].contiguous()).cpu().double() - ref64).abs().max().item()
        print(f"[attention b16 B={B} scale={scale}] max|err| vs fp64: bf16x3 kernel {err_hip:.2e}, fp32-MFMA kernel {err_f32k:.2e}, torch fp32 {err_cpu:.2e}")
        assert err_hip <= max(4 * err_cpu, 2e-5 if scale > 1 else 5e-6), (scale, err_hip, err_cpu)


def test_vit_attention_b16_stress_determinism(built_lib, cuda_dev):
    """The bf16-pipe attention stages K / V through registers into two LDS images that are rewritten every block while the other waves may
    still be a phase behind, walks two items per workgroup at 64 crops and reuses the V^T image as the staging area of its split3 output: a
    missing barrier shows up as a run-to-run difference.  30 repeats at 64 crops (both outputs) interleaved with launches that move the
    co-resident workgroups' timing (the fp32-MFMA attention kernel, a LayerNorm), then the same on a second stream concurrently."""
    from tokenhmr_amd import ops
    B = 64
    qkv = _rand(B, 192, 3840, seed=77)
    qkv[:, :, :1280] *= 80 ** -0.5
    d = qkv.to(cuda_dev)
    first, first_s = ops.vit_attention_b16(d), ops.vit_attention_b16(d, out_split=True)
    assert torch.equal(first_s, ops.split3(first.reshape(B * 192, 1280)))
    x = _rand(4096, 1280, seed=3).to(cuda_dev)
    gam, bet = torch.ones(1280, device=cuda_dev), torch.zeros(1280, device=cuda_dev)
    for i in range(30):
        if i % 3 == 1:
            ops.vit_attention(d[:16].contiguous())
        if i % 3 == 2:
            ops.layernorm(x, gam, bet, 1e-6)
        assert torch.equal(ops.vit_attention_b16(d, out_split=True), first_s), i
        assert torch.equal(ops.vit_attention_b16(d), first), i
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(6):
        a = ops.vit_attention_b16(d, out_split=True)
        with torch.cuda.stream(side):
            b = ops.vit_attention_b16(d[:40].contiguous(), out_split=True)
        outs.append((a, b))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for a, b in outs:
        assert torch.equal(a, first_s) and torch.equal(b, first_s[:40 * 192])


def test_rot6d(built_lib, cuda_dev):
    from tokenhmr_amd import ops
    from oracle import tokenhmr_oracle as O
    x = _rand(4, 144, seed=5)
    out = ops.rot6d_to_rotmat(x.to(cuda_dev)).cpu()
    ref = O.rot6d_to_rotmat(x)
    assert torch.allclose(out, ref, atol=1e-6), (out - ref).abs().max()
    eye = torch.eye(3).expand_as(out)
    assert torch.allclose(out @ out.transpose(1, 2), eye, atol=1e-5)      # orthonormal (size-independent property)
    assert torch.allclose(torch.linalg.det(out), torch.ones(out.shape[0]), atol=1e-5)


def test_aa_to_rotmat_vs_reference_golden(built_lib, cuda_dev):
    """thmr_op_aa_to_rotmat vs the reference's own aa_to_rotmat (golden fixture) and vs fp64: tolerance 2e-6 (sin/cos/sqrt
    of ocml vs ATen), zero and tiny rotations included."""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    from oracle import tokenhmr_oracle as O
    from tokenhmr_amd import ops
    g = np.load(os.path.join(GOLDEN_DIR, "geometry_small.npz"))
    th = torch.from_numpy(g["theta"])
    R = ops.aa_to_rotmat(th.to(cuda_dev)).cpu()
    assert (R - torch.from_numpy(g["rotmat"])).abs().max() < 2e-6
    big = 3.0 * _rand(4096, 3, seed=5)
    R2 = ops.aa_to_rotmat(big.to(cuda_dev)).cpu()
    assert (R2.double() - O.aa_to_rotmat(big.double())).abs().max() < 2e-6


def test_vit_attention_stress_both_variants_and_determinism(built_lib, cuda_dev):
    """The attention kernel stages K / V by asynchronous LDS-DMA tracked with s_waitcnt vmcnt(N): a missed wait or barrier shows
    up as a wrong or run-to-run varying result.  Many workgroups (B = 64: four per CU), both query-tile variants (B <= 10 uses
    64-query workgroups, larger batches 192-query ones; the first 10 crops must agree bit for bit between them), ten repeats."""
    from tokenhmr_amd import ops
    B = 64
    qkv = _rand(B, 192, 3840, seed=21)
    qkv[:, :, :1280] *= 80 ** -0.5
    t = qkv.reshape(B, 192, 3, 16, 80).permute(2, 0, 3, 1, 4).double()
    ref = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(B, 192, 1280)
    d = qkv.to(cuda_dev)
    first = ops.vit_attention(d)
    assert (first.cpu().double() - ref).abs().max() < 5e-6
    small = ops.vit_attention(d[:10].contiguous())                 # 64-query variant
    assert torch.equal(small, first[:10])
    for _ in range(10):
        assert torch.equal(ops.vit_attention(d), first)
        assert torch.equal(ops.vit_attention(d[:10].contiguous()), small)
