"""TEST INFRASTRUCTURE ONLY — golden crops from the reference's OWN dataset code (tokenhmr/lib/datasets/vitdet_dataset.py and
lib/datasets/utils.py executed in place).  cv2 and skimage are absent from this image, so they are stubbed:
  cv2.getAffineTransform / cv2.warpAffine -> oracle/crop_oracle.py restatements (UNPINNED, see its header)
  skimage.filters.gaussian                -> scipy.ndimage.gaussian_filter, the function skimage itself calls
so these fixtures pin the reference's Python logic around the two cv2 primitives (box -> centre/scale/bbox size, the
anti-alias rule and sigma, the float32 point arithmetic of gen_trans_from_patch_cv, flip / CHW / normalisation), not the
primitives themselves.    python oracle/gen_golden_crop.py   ->   tests/golden/crop_small.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import crop_oracle as CO  # noqa: E402

REF = "/root/reference/tokenhmr/lib/datasets"


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def load_reference_datasets():
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = 0, 1
    cv2.getAffineTransform = CO.get_affine_transform

    def warpAffine(img, M, dsize, flags=1, borderMode=0, borderValue=0):
        assert flags == 1 and borderMode == 0 and borderValue == 0
        return CO.warp_affine(img, M, dsize)

    cv2.warpAffine = warpAffine
    sys.modules["cv2"] = cv2
    sk = types.ModuleType("skimage")
    skf = types.ModuleType("skimage.filters")
    skt = types.ModuleType("skimage.transform")

    def gaussian(image, sigma=1, channel_axis=None, preserve_range=False, truncate=4.0, **kw):
        assert channel_axis == 2 and preserve_range
        return CO.gaussian_antialias(image, sigma, truncate)

    skf.gaussian = gaussian
    skt.rotate = skt.resize = None
    sk.filters, sk.transform = skf, skt
    sys.modules.update({"skimage": sk, "skimage.filters": skf, "skimage.transform": skt})
    yacs = types.ModuleType("yacs")
    yc = types.ModuleType("yacs.config")
    yc.CfgNode = dict
    yacs.config = yc
    sys.modules.update({"yacs": yacs, "yacs.config": yc})
    for name in ("webdataset", "braceexpand"):
        sys.modules.setdefault(name, types.ModuleType(name))
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
    return mods


def synthetic_frame(H=540, W=720, seed=0):
    """A frame with structure at several scales (smooth gradients + blocks + noise), BGR uint8."""
    rng = np.random.default_rng(7000 + seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 100 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], -1)
    img += 40 * (((xx // 16) + (yy // 16)) % 2)[..., None]
    img += rng.normal(0, 12, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


BOXES = np.array([[100.3, 50.2, 260.9, 400.7],      # tall box, no blur (bbox 350 -> f = 0.68)
                  [-40.0, -30.0, 200.0, 300.0],     # sticks out top-left: zero border
                  [10.0, 5.0, 715.0, 535.0],        # nearly the whole frame: f = 1.38 -> blurred, float64 warp
                  [600.5, 400.25, 735.0, 560.0],    # sticks out bottom-right
                  [300.0, 200.0, 330.0, 260.0]],    # tiny box: 8x up-sampling
                 dtype=np.float64)


def main():
    mods = load_reference_datasets()
    frame = synthetic_frame()
    out = {"frame": frame, "boxes": BOXES}
    for tag, bbox_shape in (("ar", [192, 256]), ("sq", None)):
        cfg = _Cfg(MODEL=_Cfg(IMAGE_SIZE=256, IMAGE_MEAN=[0.485, 0.456, 0.406], IMAGE_STD=[0.229, 0.224, 0.225],
                              **({"BBOX_SHAPE": bbox_shape} if bbox_shape else {})))
        ds = mods["vitdet_dataset"].ViTDetDataset(cfg, frame, BOXES)
        items = [ds[i] for i in range(len(ds))]
        ours = [CO.vitdet_item(frame, b, 256, bbox_shape, numpy1=False) for b in BOXES]
        np1 = [CO.vitdet_item(frame, b, 256, bbox_shape) for b in BOXES]
        d = max(np.abs(a["img"] - b["img"]).max() for a, b in zip(ours, np1))
        assert d < 5e-7, d          # numpy-1.23 float32 normalisation vs this image's numpy: <= 1 ulp
        for k in ("img", "box_center", "box_size", "img_size"):
            ref = np.stack([np.asarray(it[k]) for it in items])
            mine = np.stack([np.asarray(it[k]) for it in ours])
            assert ref.dtype == mine.dtype or k != "img", (k, ref.dtype, mine.dtype)
            assert np.array_equal(ref, mine), (tag, k, np.abs(ref.astype(np.float64) - mine).max())
        print(tag, "reference ViTDetDataset == crop_oracle.vitdet_item bit for bit; sigmas", [round(o["sigma"], 3) for o in ours])
        out[f"img_{tag}"] = np.stack([it["img"] for it in items]).astype(np.float32)[:, :, ::4, ::4]     # 64x64 sub-grid
        out[f"box_size_{tag}"] = np.array([it["box_size"] for it in items], dtype=np.float64)
        out[f"trans_{tag}"] = np.stack([o["trans"] for o in ours])
        out[f"sigma_{tag}"] = np.array([o["sigma"] for o in ours])
    # eval.py crop: generate_image_patch_cv2 + the tail of get_example (no augmentation), via the reference's own functions
    U = mods["utils"]
    mean, std = 255.0 * np.array([0.485, 0.456, 0.406]), 255.0 * np.array([0.229, 0.224, 0.225])
    ex = []
    for (cx, cy, w) in ((360.2, 270.1, 380.0), (80.0, 500.0, 300.0)):
        patch, trans = U.generate_image_patch_cv2(frame, cx, cy, w, w, 256, 256, False, 1.0, 0, border_mode=0)
        img = U.convert_cvimg_to_tensor(patch[:, :, ::-1])
        for c in range(3):
            img[c] = (np.clip(img[c] * 1.0, 0, 255) - mean[c]) / std[c]
        mine = CO.example_item(frame, cx, cy, w, w, numpy1=False)
        assert np.array_equal(img, mine["img"]) and np.array_equal(trans, mine["trans"])
        ex.append(img[:, ::4, ::4])
    out["example_params"] = np.array([(360.2, 270.1, 380.0), (80.0, 500.0, 300.0)])
    out["example_img"] = np.stack(ex).astype(np.float32)
    print("reference generate_image_patch_cv2 + get_example tail == crop_oracle.example_item bit for bit")
    path = os.path.join(ROOT, "tests", "golden", "crop_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
