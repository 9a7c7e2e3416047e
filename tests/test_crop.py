"""Crop preprocessing.  CPU: the oracle restatement against the fixtures produced by the reference's own
dataset code (oracle/gen_golden_crop.py), warp properties, and the host-side box -> affine logic of the product mirror.
GPU (-m gpu): thmr_cropper_run through the C ABI against the oracle and the fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import crop_oracle as CO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crop_small.npz")


class _Node(dict):
    __getattr__ = dict.__getitem__


def _cfg(bbox_shape):
    m = _Node(IMAGE_SIZE=256, IMAGE_MEAN=[0.485, 0.456, 0.406], IMAGE_STD=[0.229, 0.224, 0.225])
    if bbox_shape:
        m["BBOX_SHAPE"] = bbox_shape
    return _Node(MODEL=m)


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def big_frame(H=1080, W=1920, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 90 * np.sin(xx / 51.0 + c) * np.cos(yy / 29.0 - c) for c in range(3)], -1)
    img += 50 * (((xx // 9) + (yy // 13)) % 2)[..., None] + rng.normal(0, 10, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("tag,shape", [("ar", [192, 256]), ("sq", None)])
def test_oracle_equals_reference_fixture(gold, tag, shape):
    """crop_oracle.vitdet_item vs crops the reference's ViTDetDataset produced (sub-grid).  numpy1=False is the expression as
    this image's numpy evaluates it -> bit-equal; the default (numpy 1.23 float32 semantics, what the HIP kernel does) <= 1 ulp."""
    for i, box in enumerate(gold["boxes"]):
        a = CO.vitdet_item(gold["frame"], box, 256, shape, numpy1=False)
        assert np.array_equal(a["img"][:, ::4, ::4], gold[f"img_{tag}"][i])
        assert a["box_size"] == gold[f"box_size_{tag}"][i] and a["sigma"] == gold[f"sigma_{tag}"][i]
        b = CO.vitdet_item(gold["frame"], box, 256, shape)
        assert np.abs(b["img"][:, ::4, ::4] - gold[f"img_{tag}"][i]).max() < 5e-7
    for (cx, cy, w), ref in zip(gold["example_params"], gold["example_img"]):
        assert np.array_equal(CO.example_item(gold["frame"], cx, cy, w, w, numpy1=False)["img"][:, ::4, ::4], ref)


def test_warp_properties():
    """Size-independent properties of the restated cv2.warpAffine: identity, integer shifts with zero border, half-pixel
    averaging with OpenCV's rounding, constant images under scaling (uint8 and float64 paths)."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(CO.warp_affine(img, I, (56, 40)), img)
    assert np.array_equal(CO.warp_affine(img.astype(np.float64), I, (56, 40)), img.astype(np.float64))
    T = np.array([[1.0, 0, 5], [0, 1.0, -3]])            # dst(x, y) = src(x - 5, y + 3)
    out = CO.warp_affine(img, T, (56, 40))
    assert np.array_equal(out[:37, 5:], img[3:, :51]) and not out[:, :5].any() and not out[37:].any()
    Hh = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])          # half-pixel shift: (a + b + 1) >> 1 in OpenCV's fixed point
    out = CO.warp_affine(img, Hh, (56, 40)).astype(np.int64)
    exp = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + 16384) >> 15
    assert np.array_equal(out[:, 1:], exp)
    const = np.full((30, 30, 3), 200, np.uint8)
    S = np.array([[3.7, 0, -20.0], [0, 3.7, -20.0]])
    assert (CO.warp_affine(const, S, (64, 64))[8:56, 8:56] == 200).all()
    assert np.allclose(CO.warp_affine(const.astype(np.float64), S, (64, 64))[8:56, 8:56], 200.0, atol=1e-12)


def test_affine_through_three_points():
    M = CO.gen_trans_from_patch_cv(100.5, 80.25, 300.0, 300.0, 256, 256, 1.0, 0)
    s = 256 / 300.0
    assert np.allclose(M, [[s, 0, 128 - s * 100.5], [0, s, 128 - s * 80.25]], atol=1e-9)
    M = CO.gen_trans_from_patch_cv(10, 20, 50.0, 50.0, 256, 256, 1.0, 90)    # rotation is part of the restated function
    assert np.allclose(M @ np.array([10, 20, 1.0]), [128, 128], atol=1e-4)


@pytest.mark.parametrize("tag,shape", [("ar", [192, 256]), ("sq", None)])
def test_product_host_logic_matches_fixture(gold, tag, shape):
    """tokenhmr_amd.preprocess (product path, no oracle import): box -> bbox size / sigma / affine equal to what the reference's
    code produced.  No GPU is touched: the cropper handle is created lazily."""
    from tokenhmr_amd.preprocess import ViTDetDataset
    ds = ViTDetDataset(_cfg(shape), gold["frame"], gold["boxes"], device="cuda:0")
    assert len(ds) == len(gold["boxes"])
    for i in range(len(ds)):
        size, sigma, trans = ds._params(i)
        o1 = CO.vitdet_item(gold["frame"], gold["boxes"][i], 256, shape)
        # the crop_small fixture ran under numpy 2 (float32 box arithmetic); the product follows the pinned numpy 1.23 (float64)
        assert float(size) == float(o1["box_size"]) and np.float32(size) == np.float32(gold[f"box_size_{tag}"][i])
        assert np.array_equal(trans, o1["trans"]) and np.abs(trans - gold[f"trans_{tag}"][i]).max() < 1e-4
        # sigma: the fixture ran under numpy 2 (float32 arithmetic), the product follows the pinned numpy 1.23 (float64)
        assert abs(sigma - gold[f"sigma_{tag}"][i]) <= 1e-6 * sigma and (sigma > 0) == (gold[f"sigma_{tag}"][i] > 0)
        assert sigma == CO.vitdet_item(gold["frame"], gold["boxes"][i], 256, shape)["sigma"]


def test_cropper_without_gpu_fails_loudly(built_lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tokenhmr_amd.preprocess import Cropper
    with pytest.raises(Exception):
        Cropper("cuda:0")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Cropper("cpu")


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,shape", [("ar", [192, 256]), ("sq", None)])
def test_gpu_vitdet_vs_fixture_and_oracle(built_lib, cuda_dev, gold, tag, shape):
    from tokenhmr_amd.preprocess import ViTDetDataset
    ds = ViTDetDataset(_cfg(shape), gold["frame"], gold["boxes"], device=cuda_dev)
    batch = ds.batch()
    img = batch["img"].cpu().numpy()
    assert img.shape == (5, 3, 256, 256) and batch["box_size"].dtype == torch.float64
    assert np.abs(img[:, :, ::4, ::4] - gold[f"img_{tag}"]).max() < 1e-6          # reference-produced fixture (numpy-2 rounding: 1 ulp)
    for i, box in enumerate(gold["boxes"]):
        o = CO.vitdet_item(gold["frame"], box, 256, shape)
        if o["sigma"] == 0:
            assert np.array_equal(img[i], o["img"]), i                           # integer bilinear + float32 normalisation: bit-exact
        else:
            d = np.abs(img[i] - o["img"])
            assert d.max() < 1e-6 and (d > 0).mean() < 1e-4, (d.max(), (d > 0).mean())     # fp64 blur in scipy's own operation order
        assert batch["box_size"][i].item() == float(o["box_size"])
    one = ds[3]
    assert torch.equal(one["img"], batch["img"][3]) and one["personid"] == 3


@pytest.mark.gpu
def test_gpu_large_frame_blurred_crops(built_lib, cuda_dev):
    """1080p frame, boxes large enough for the anti-alias branch (sigma 0.5 .. 1.9, kernel radius 2 .. 8), touching every frame
    edge so that the blurred region is clipped and the 'nearest' edge rule is exercised."""
    from tokenhmr_amd.preprocess import ViTDetDataset
    frame = big_frame()
    boxes = np.array([[400, 100, 1100, 1000], [-200, -100, 900, 1200], [900, -50, 2100, 1150], [0, 0, 1920, 1080],
                      [1500, 700, 1900, 1079], [-3000, -3000, 5000, 5000.0]])
    ds = ViTDetDataset(_cfg([192, 256]), frame, boxes, device=cuda_dev)
    img = ds.batch()["img"].cpu().numpy()
    sig = []
    for i, box in enumerate(boxes):
        o = CO.vitdet_item(frame, box, 256, [192, 256])
        sig.append(o["sigma"])
        d = np.abs(img[i] - o["img"])
        assert d.max() < 2e-6, (i, o["sigma"], d.max())
    assert sum(s > 0 for s in sig) >= 4 and min(sig) == 0.0 and max(sig) > 1.5


@pytest.mark.gpu
def test_gpu_eval_crops(built_lib, cuda_dev, gold):
    from tokenhmr_amd.preprocess import Cropper, crop_examples
    cr = Cropper(cuda_dev)
    p = gold["example_params"]
    img, trans = crop_examples(cr, gold["frame"], p[:, :2], np.stack([p[:, 2], p[:, 2]], 1))
    assert np.abs(img.cpu().numpy()[:, :, ::4, ::4] - gold["example_img"]).max() < 1e-6
    for i, (cx, cy, w) in enumerate(p):
        o = CO.example_item(gold["frame"], cx, cy, w, w)
        assert np.array_equal(img[i].cpu().numpy(), o["img"]) and np.array_equal(trans[i], o["trans"])
    # get_example's anti-alias rule (utils.py:583-587, truncate 3.0): only fires when the box is SMALLER than the patch
    img2, _ = crop_examples(cr, gold["frame"], [[300.0, 200.0]], [[90.0, 90.0]], use_skimage_antialias=True)
    o = CO.example_item(gold["frame"], 300.0, 200.0, 90.0, 90.0, use_skimage_antialias=True)
    assert o["sigma"] > 0 and np.abs(img2[0].cpu().numpy() - o["img"]).max() < 2e-6


@pytest.mark.gpu
def test_gpu_cropper_errors_and_rgb(built_lib, cuda_dev, gold):
    from tokenhmr_amd import _cabi
    from tokenhmr_amd.preprocess import Cropper
    cr = Cropper(cuda_dev)
    with pytest.raises(_cabi.EngineError):
        cr.warp(gold["frame"], np.full((1, 2, 3), np.nan))              # non-finite affine
    with pytest.raises(_cabi.EngineError):
        cr.warp(gold["frame"], np.eye(2, 3)[None], sigmas=[-1.0])
    with pytest.raises(ValueError):
        cr.warp(gold["frame"].astype(np.float32), np.eye(2, 3)[None])
    I = np.array([[[1.0, 0, 0], [0, 1.0, 0]]])
    a = cr.warp(gold["frame"], I, patch=64, mean=(0, 0, 0), std=(1 / 255.0,) * 3, is_bgr=False).cpu().numpy()
    assert np.array_equal(a[0], gold["frame"][:64, :64].transpose(2, 0, 1).astype(np.float32))     # identity, no flip
    b = cr.warp(gold["frame"], I, patch=64, mean=(0, 0, 0), std=(1 / 255.0,) * 3, is_bgr=True).cpu().numpy()
    assert np.array_equal(b[0], a[0][::-1])


@pytest.mark.gpu
def test_gpu_crops_feed_the_model(built_lib, cuda_dev, gold):
    """demo.py's loop body with both pieces swapped in: ViTDetDataset(...).batch() -> model(batch)."""
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.model import TokenHMR
    from tokenhmr_amd.preprocess import ViTDetDataset
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    model = TokenHMR.from_state(cfg, W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0),
                                max_batch=8, device=cuda_dev)
    batch = ViTDetDataset(_cfg([192, 256]), gold["frame"], gold["boxes"], device=cuda_dev).batch()
    out = model(batch)
    assert out["pred_vertices"].shape == (5, 6890, 3) and torch.isfinite(out["pred_vertices"]).all()


# ------------------------------------------------------------------------------------------------ numpy-1 / real-skimage fixture
GOLD1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crop_numpy1.npz")


def _numpy1_cases():
    from oracle.gen_golden_crop_numpy1 import big_frame
    g1, small = dict(np.load(GOLD1)), dict(np.load(GOLD))
    big = big_frame()
    assert int(big.astype(np.int64).sum()) == int(g1["big_frame_checksum"][0])
    return g1, (("small", small["frame"], small["boxes"]), ("big", big, g1["big_boxes"]))


def test_oracle_equals_numpy1_realskimage_fixture():
    """tests/golden/crop_numpy1.npz was produced by the reference's ViTDetDataset under numpy 1.26 (the promotion rules of the
    pinned numpy 1.23) with the REAL scikit-image gaussian (oracle/gen_golden_crop_numpy1.py).  The oracle's numpy1=True
    restatement, evaluated HERE under numpy 2 / a newer scipy, must reproduce it bit for bit: this pins the float32
    normalisation, the float64 sigma and 'skimage gaussian == scipy.ndimage.gaussian_filter'."""
    g1, cases = _numpy1_cases()
    assert str(g1["versions"][0]).startswith("1.")
    nblur = 0
    for tag, frame, boxes in cases:
        for i, box in enumerate(boxes):
            o = CO.vitdet_item(frame, box, 256, [192, 256])
            assert np.array_equal(o["img"][:, ::4, ::4], g1[f"img_{tag}"][i]), (tag, i)
            assert float(o["box_size"]) == float(g1[f"box_size_{tag}"][i])
            nblur += o["sigma"] > 0
    assert nblur >= 4


@pytest.mark.gpu
def test_gpu_vs_numpy1_realskimage_fixture(built_lib, cuda_dev):
    """The HIP crops against the same fixture: bit-exact where no blur is involved, < 1e-6 (a handful of float32 roundings of
    the fp64 blur) where it is."""
    from tokenhmr_amd.preprocess import ViTDetDataset
    g1, cases = _numpy1_cases()
    for tag, frame, boxes in cases:
        ds = ViTDetDataset(_cfg([192, 256]), frame, boxes, device=cuda_dev)
        img = ds.batch()["img"].cpu().numpy()[:, :, ::4, ::4]
        for i in range(len(boxes)):
            d = np.abs(img[i] - g1[f"img_{tag}"][i])
            if ds._params(i)[1] == 0:
                assert d.max() == 0, (tag, i, d.max())
            else:
                assert d.max() < 1e-6 and (d > 0).mean() < 1e-3, (tag, i, d.max(), (d > 0).mean())


def test_restated_warp_is_close_to_an_independent_bilinear_warp():
    """cv2.warpAffine itself cannot be pinned here (opencv is absent), but its restatement can be bounded: against
    scikit-image's floating-point bilinear affine warp (fixture produced by the real skimage, see
    oracle/gen_golden_crop_numpy1.py) the 5-bit fixed-point result differs by well under one grey level on average.  A wrong
    matrix convention, half-pixel offset or axis swap would show tens of levels."""
    g1 = dict(np.load(GOLD1))
    small = dict(np.load(GOLD))
    for i, box in enumerate(small["boxes"]):
        o = CO.vitdet_item(small["frame"], box, 256, [192, 256])
        mine = CO.warp_affine(small["frame"], o["trans"], (256, 256)).astype(np.float64)[::4, ::4]
        d = np.abs(mine - g1["skimage_bilinear_small"][i])
        assert d.mean() < 0.5 and np.percentile(d, 99) < 1.6 and d.max() < 6.0, (i, d.mean(), d.max())
