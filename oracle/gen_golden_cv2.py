"""TEST INFRASTRUCTURE ONLY — the STAGED pin of the OpenCV boundary of crop preprocessing.

    python oracle/gen_golden_cv2.py      -> tests/golden/cv2_warp.npz        (needs the REAL `cv2`, opencv-python 4.x)

`cv2.getAffineTransform` and `cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT)` (tokenhmr/lib/datasets/utils.py:81-128,351-356;
tokenhmr/requirements.txt:5) are third-party code that is absent from this image: `oracle/crop_oracle.py` restates them from
OpenCV 4.x imgwarp.cpp and parity there is UNPINNED (bounded against scikit-image's independent bilinear warp only).  The day
opencv-python is importable this script freezes, from the real library:
  * getAffineTransform on seeded float32 point triples (incl. the triples gen_trans_from_patch_cv produces);
  * warpAffine of the synthetic uint8 frame of gen_golden_crop.py, and of its float64 blurred version (the anti-alias
    branch hands cv2 a float64 image), under the affines of five boxes that cross every frame edge and zoom 0.3x ... 8x;
  * the reference's own generate_image_patch_cv2 executed in place over the real cv2.
tests/test_cv2_pin.py then holds crop_oracle (CPU) and crop.hip (GPU) to it bit for bit; while the fixture is absent those
tests SKIP with the reason "parity unpinned".
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import crop_oracle as CO  # noqa: E402
from oracle.gen_golden_crop import synthetic_frame, BOXES  # noqa: E402


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("cv2 is not importable here: the OpenCV boundary stays UNPINNED (this script is the staged pin)")
    if not hasattr(cv2, "__version__") or not hasattr(cv2, "remap"):
        sys.exit("`cv2` resolves to a stub, not opencv-python")
    rng = np.random.default_rng(9000)
    out = {"cv2_version": np.array(cv2.__version__)}
    # ---- getAffineTransform
    src = (rng.standard_normal((32, 3, 2)) * 200).astype(np.float32)
    dst = (rng.standard_normal((32, 3, 2)) * 100 + 128).astype(np.float32)
    out["gat_src"], out["gat_dst"] = src, dst
    out["gat_M"] = np.stack([cv2.getAffineTransform(s, d) for s, d in zip(src, dst)])
    # ---- warpAffine under the affines the dataset code builds for BOXES (utils.py:81-128 via the restated helper)
    frame = synthetic_frame()
    blurred = CO.gaussian_antialias(frame.astype(np.float64), 1.2, 4.0)      # float64 input = the anti-alias branch
    Ms = []
    for b in BOXES:
        cx, cy = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
        w = max(b[2] - b[0], b[3] - b[1])
        Ms.append(CO.gen_trans_from_patch_cv(cx, cy, w, w, 256, 256, 1.0, 0))
    Ms = np.stack(Ms)
    out["warp_M"] = Ms
    u8 = np.stack([cv2.warpAffine(frame, M, (256, 256), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT) for M in Ms])
    f64 = np.stack([cv2.warpAffine(blurred, M, (256, 256), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT) for M in Ms])
    out["warp_u8"] = u8                                                       # full crops: 5 x 196 KB, compressible
    out["warp_f64_sub"] = f64[:, ::4, ::4]
    out["warp_f64_sum"] = f64.reshape(len(Ms), -1).sum(1)
    # how far the restatement is from the real thing, reported at generation time
    mine_u8 = np.stack([CO.warp_affine(frame, M, (256, 256)) for M in Ms])
    mine_f64 = np.stack([CO.warp_affine(blurred, M, (256, 256)) for M in Ms])
    print("getAffineTransform: max|diff| =", np.abs(out["gat_M"] - np.stack([CO.get_affine_transform(s, d) for s, d in zip(src, dst)])).max())
    print("warpAffine uint8:   differing pixels =", int((mine_u8 != u8).sum()), "of", u8.size)
    print("warpAffine float64: max|diff| =", np.abs(mine_f64 - f64).max())
    path = os.path.join(ROOT, "tests", "golden", "cv2_warp.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
