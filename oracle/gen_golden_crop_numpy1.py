"""TEST INFRASTRUCTURE ONLY — golden crops from the reference's OWN dataset code executed under the environment the reference
pins: a numpy 1.x interpreter (legacy value-based promotion, requirements.txt:1 numpy==1.23.1) with the REAL scikit-image.

This image happens to carry such an interpreter besides the system one:  /opt/conda/bin/python3.9  with numpy 1.26.4,
scipy 1.7.1 and scikit-image 0.18.3 (no torch, no cv2, no yacs).  Run

    /opt/conda/bin/python3.9 oracle/gen_golden_crop_numpy1.py        ->  tests/golden/crop_numpy1.npz

What is real and what is stubbed in that run:
  * numpy 1.26 promotion rules                      REAL  (float32 normalisation, float64 sigma — see crop_oracle.finish_patch)
  * skimage.filters.gaussian                        REAL  (0.18.3 spells the argument multichannel=True instead of
                                                           channel_axis=2, which arrived in 0.19; the shim below only renames it)
  * tokenhmr/lib/datasets/vitdet_dataset.py, utils.py   REAL, executed in place
  * cv2.getAffineTransform / cv2.warpAffine         STUB -> oracle/crop_oracle.py restatement (opencv is absent: UNPINNED)
  * torch (only `torch.utils.data.Dataset` as a base class), yacs.config.CfgNode      trivial stubs
The system-python tests then require crop_oracle.vitdet_item(numpy1=True) — computed under numpy 2.2 / scipy 1.15 — to equal
these tensors BIT FOR BIT, which pins (a) the numpy-1 restatement and (b) "skimage gaussian == scipy gaussian_filter".
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import crop_oracle as CO  # noqa: E402  (numpy + scipy only: importable without the repo's torch-based packages)

REF = "/root/reference/tokenhmr/lib/datasets"


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def load_reference_datasets():
    assert int(np.__version__.split(".")[0]) == 1, "run me with a numpy 1.x interpreter (see the module docstring)"
    import skimage
    import skimage.filters as real_filters
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = 0, 1
    cv2.getAffineTransform = CO.get_affine_transform
    cv2.warpAffine = lambda img, M, dsize, flags=1, borderMode=0, borderValue=0: CO.warp_affine(img, M, dsize)
    sys.modules["cv2"] = cv2
    real_gaussian = real_filters.gaussian
    if "channel_axis" not in real_gaussian.__code__.co_varnames:          # scikit-image < 0.19

        def gaussian(image, sigma=1, channel_axis=None, **kw):
            return real_gaussian(image, sigma=sigma, multichannel=(channel_axis is not None), **kw)

        real_filters.gaussian = gaussian
    torch = types.ModuleType("torch")
    tu = types.ModuleType("torch.utils")
    tud = types.ModuleType("torch.utils.data")
    tud.Dataset = object
    tu.data, torch.utils = tud, tu
    sys.modules.update({"torch": torch, "torch.utils": tu, "torch.utils.data": tud})
    yacs = types.ModuleType("yacs")
    yc = types.ModuleType("yacs.config")
    yc.CfgNode = dict
    yacs.config = yc
    sys.modules.update({"yacs": yacs, "yacs.config": yc})
    pkg = types.ModuleType("_ref_ds")
    pkg.__path__ = [REF]
    sys.modules["_ref_ds"] = pkg
    mods = {}
    for name in ("utils", "vitdet_dataset"):
        spec = importlib.util.spec_from_file_location(f"_ref_ds.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "_ref_ds"
        sys.modules[f"_ref_ds.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods, skimage.__version__


def big_frame(H=900, W=1400):
    """A second, larger frame so that several crops take the anti-alias branch with different sigmas (kernel radius 2..4).
    Integer arithmetic + the legacy RandomState stream only, so every interpreter / numpy version rebuilds the same bytes
    (the fixture does not have to carry the 3.8 MB frame)."""
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.RandomState(12)
    base = np.stack([(xx * (3 + c) + yy * (5 - c)) // 9 % 200 for c in range(3)], -1)
    img = base + 40 * (((xx // 11) + (yy // 7)) % 2)[..., None] + rng.randint(0, 16, size=(H, W, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    mods, skv = load_reference_datasets()
    import scipy
    small = np.load(os.path.join(ROOT, "tests", "golden", "crop_small.npz"))
    frame, boxes = small["frame"], small["boxes"]
    big = big_frame()
    big_boxes = np.array([[50.0, 20.0, 1350.0, 880.0], [-100.0, 100.0, 700.0, 1000.0], [600.0, -200.0, 1500.0, 850.0],
                          [300.0, 300.0, 500.0, 700.0]])
    out = {"big_boxes": big_boxes, "big_frame_checksum": np.array([int(big.astype(np.int64).sum())]),
           "versions": np.array([np.__version__, scipy.__version__, skv])}
    for tag, fr, bx in (("small", frame, boxes), ("big", big, big_boxes)):
        cfg = _Cfg(MODEL=_Cfg(IMAGE_SIZE=256, IMAGE_MEAN=[0.485, 0.456, 0.406], IMAGE_STD=[0.229, 0.224, 0.225], BBOX_SHAPE=[192, 256]))
        ds = mods["vitdet_dataset"].ViTDetDataset(cfg, fr, bx)
        items = [ds[i] for i in range(len(ds))]
        img = np.stack([it["img"] for it in items])
        assert img.dtype == np.float32
        mine = np.stack([CO.vitdet_item(fr, b, 256, [192, 256])["img"] for b in bx])       # numpy1=True restatement, same interpreter
        print(tag, "reference (numpy", np.__version__, "+ skimage", skv, ") vs crop_oracle numpy1=True here: max|diff| =",
              float(np.abs(img - mine).max()))
        out[f"img_{tag}"] = img[:, :, ::4, ::4]
        out[f"box_size_{tag}"] = np.array([it["box_size"] for it in items], dtype=np.float64)
    # Plausibility bound for the UNPINNED primitive: the restated cv2.warpAffine (5-bit fixed-point bilinear) against
    # scikit-image's independent floating-point bilinear affine warp on the un-blurred crops.  They are different algorithms
    # (coordinates quantised to 1/32 px, integer weights and rounding in OpenCV), so only closeness is expected: a wrong
    # matrix convention, a half-pixel offset or swapped axes would show up as tens of grey levels.
    from skimage.transform import AffineTransform, warp
    sk = []
    for b in boxes:
        o = CO.vitdet_item(frame, b, 256, [192, 256])
        T = AffineTransform(matrix=np.vstack([o["trans"], [0, 0, 1]]))
        ref = warp(frame, T.inverse, output_shape=(256, 256), order=1, mode="constant", cval=0, preserve_range=True)
        mine = CO.warp_affine(frame, o["trans"], (256, 256)).astype(np.float64)
        d = np.abs(mine - ref)
        print("restated cv2.warpAffine vs skimage bilinear warp: mean |diff| %.3f  p99 %.3f  max %.3f grey levels" %
              (d.mean(), np.percentile(d, 99), d.max()))
        assert d.mean() < 0.5 and np.percentile(d, 99) < 1.5 and d.max() < 6.0
        sk.append(ref[::4, ::4].astype(np.float32))
    out["skimage_bilinear_small"] = np.stack(sk)
    path = os.path.join(ROOT, "tests", "golden", "crop_numpy1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
