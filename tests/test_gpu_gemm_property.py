"""Property-based parity of the fp32 MFMA GEMM family (-m gpu): random shapes (ragged M / N edges, K = 32 .. 2048), every
tile variant, small-M ring kernel with split-K, random epilogues, against an fp64 reference; plus the invariant that every
variant with an unsplit K accumulates in the same order (bit-identical across tile shapes)."""
import math

import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

VARIANTS = ["128x128", "128x160", "128x96", "64x64", "ring4", "ring8", "auto"]
EPIS = ["none", "bias", "bias_gelu", "bias_relu", "bias_resid", "bias_qscale"]


def _ref(a, w, b, r, epi, qscale, qcols):
    c = a.double() @ w.double().t()
    if epi != "none":
        c = c + b.double()
    if epi == "bias_gelu":
        c = F.gelu(c)
    elif epi == "bias_relu":
        c = F.relu(c)
    elif epi == "bias_resid":
        c = r.double() + c
    elif epi == "bias_qscale":
        c[:, :qcols] = c[:, :qcols] * qscale
    return c.float()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(M=st.integers(1, 700), N=st.integers(1, 700), kt=st.integers(1, 64), epi=st.sampled_from(EPIS), seed=st.integers(0, 2 ** 16))
def test_gemm_random_shapes_all_variants(built_lib, cuda_dev, M, N, kt, epi, seed):
    from tokenhmr_amd import ops
    K = 32 * kt
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    qcols = N // 2
    ref = _ref(a, w, b, r, epi, 0.25, qcols)
    ad, wd, bd, rd = a.to(cuda_dev), w.to(cuda_dev), b.to(cuda_dev), r.to(cuda_dev)
    outs = {}
    for v in VARIANTS:
        out = ops.gemm(ad, wd, None if epi == "none" else bd, rd if epi == "bias_resid" else None, epi=epi, qscale=0.25,
                       qcols=qcols, variant=v)
        outs[v] = out
        assert torch.allclose(out.cpu(), ref, atol=4e-5, rtol=1e-5), (v, (out.cpu() - ref).abs().max())
    for v in VARIANTS[1:]:
        assert torch.equal(outs[v], outs[VARIANTS[0]]), v          # same K order in every unsplit variant
    # split-K (fixed-order reduction): needs K % (32 * ksplit) == 0 and N % 4 == 0
    for ks in (2, 4):
        if kt % ks == 0 and N % 4 == 0:
            out = ops.gemm(ad, wd, None if epi == "none" else bd, rd if epi == "bias_resid" else None, epi=epi, qscale=0.25,
                           qcols=qcols, variant=f"ring4/k{ks}")
            assert torch.allclose(out.cpu(), ref, atol=4e-5, rtol=1e-5), (ks, (out.cpu() - ref).abs().max())
    # tiny-M kernel (K split over the 8 waves of a workgroup): K % 256 == 0; its own association, fixed -> run twice, bit-equal
    if kt % 8 == 0:
        kw = dict(epi=epi, qscale=0.25, qcols=qcols, variant="tiny")
        o1 = ops.gemm(ad, wd, None if epi == "none" else bd, rd if epi == "bias_resid" else None, **kw)
        o2 = ops.gemm(ad, wd, None if epi == "none" else bd, rd if epi == "bias_resid" else None, **kw)
        assert torch.allclose(o1.cpu(), ref, atol=4e-5, rtol=1e-5), ("tiny", (o1.cpu() - ref).abs().max())
        assert torch.equal(o1, o2)
