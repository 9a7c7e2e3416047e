"""CPU (-m "not gpu"): the arithmetic claim behind the "split3" GEMM mode (the engine default since round 5; tokenhmr_amd/csrc/gemm_split16.hip), checked in numpy:
an fp32 number is the sum of three bf16 pieces to 2^-24, products of pieces are exact in fp32, and keeping the six piece pairs down to
2^-16 leaves an error below one fp32 rounding of the product — so a dot product accumulated in fp32 from them is as accurate as the
fp32 dot product itself."""
import numpy as np


def _rne_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _split3(x):
    h = _rne_bf16(x)
    r1 = (x - h).astype(np.float32)
    m = _rne_bf16(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, _rne_bf16(r2), r1, r2


def test_three_bf16_pieces_carry_an_fp32_number():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    h, m, l, r1, r2 = _split3(x)
    # the two subtractions are exact (computed in fp32 == computed in fp64)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    back = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.all(np.abs(back - x.astype(np.float64)) <= np.abs(x.astype(np.float64)) * 2.0 ** -24)
    # each piece is a bf16 number: its low 16 bits are zero
    for p in (h, m, l):
        assert not np.any(p.view(np.uint32) & 0xFFFF)


def test_six_products_are_an_fp32_grade_product():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    ah, am, al, _, _ = _split3(a)
    bh, bm, bl, _, _ = _split3(b)
    # a product of two bf16 numbers is exact in fp32 (8 x 8 significand bits)
    for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        assert np.array_equal((p * q).astype(np.float64), p.astype(np.float64) * q.astype(np.float64))
    six = sum(p.astype(np.float64) * q.astype(np.float64) for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)))
    exact = a.astype(np.float64) * b.astype(np.float64)
    assert np.all(np.abs(six - exact) <= np.abs(exact) * 2.0 ** -22)          # the three dropped pairs: < 3 * 2^-24 + representation
    assert np.sqrt(np.mean(((six - exact) / exact) ** 2)) < 2.0 ** -24


def test_split_dot_product_is_no_worse_than_the_fp32_one():
    rng = np.random.default_rng(2)
    K, n = 1280, 400
    a = rng.standard_normal((n, K)).astype(np.float32)
    a[:, ::97] *= 40.0
    w = (rng.standard_normal((n, K)) / np.sqrt(K)).astype(np.float32)
    exact = np.einsum("ik,ik->i", a.astype(np.float64), w.astype(np.float64))
    bound = np.einsum("ik,ik->i", np.abs(a).astype(np.float64), np.abs(w).astype(np.float64))
    # fp32 fmaf chain (what v_mfma_f32_32x32x2_f32 computes, one k at a time)
    acc = np.zeros(n, np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)
    # split3: per 16 k and piece pair, exact products summed (the MFMA's internal sum, idealised) then ONE fp32 accumulate
    ah, am, al, _, _ = _split3(a)
    wh, wm, wl, _, _ = _split3(w)
    acc3 = np.zeros(n, np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        for p, q in ((al, wh), (ah, wl), (am, wm), (am, wh), (ah, wm), (ah, wh)):
            acc3 = (acc3.astype(np.float64) + np.einsum("ik,ik->i", p[:, s].astype(np.float64), q[:, s].astype(np.float64))).astype(np.float32)
    e32 = np.abs(acc.astype(np.float64) - exact) / bound
    e3 = np.abs(acc3.astype(np.float64) - exact) / bound
    assert e3.max() <= 2.0 * e32.max() and np.sqrt(np.mean(e3 ** 2)) <= 1.5 * np.sqrt(np.mean(e32 ** 2)), (e3.max(), e32.max())
