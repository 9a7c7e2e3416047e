"""Crop preprocessing on the GPU — drop-in for the reference's per-crop CPU code.

    from tokenhmr_amd.preprocess import ViTDetDataset          # instead of lib.datasets.vitdet_dataset.ViTDetDataset (demo.py:71)
    batch = ViTDetDataset(model_cfg, img_cv2, boxes, device="cuda:0").batch()      # == next(iter(DataLoader(dataset, ...)))
    out = model(batch)

`ViTDetDataset` mirrors tokenhmr/lib/datasets/vitdet_dataset.py:16-88: same constructor arguments, `len()`, `ds[i]` items with
the same keys ('img', 'personid', 'box_center', 'box_size', 'img_size'), plus `.batch()` which produces all crops of the frame
in ONE GPU call, already collated and resident on the device.  `crop_examples` is the same for the eval.py crop
(`get_example` without augmentation, lib/datasets/utils.py:501-638).

The box -> affine arithmetic (3 points, float32, lib/datasets/utils.py:81-128 + cv2.getAffineTransform) runs here on the host
in numpy; warp, anti-alias blur, channel flip and normalisation run in csrc/crop.hip through the C ABI (thmr_cropper_run).
torch is used only to hold device memory.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _cabi

DEFAULT_MEAN = (0.485, 0.456, 0.406)
DEFAULT_STD = (0.229, 0.224, 0.225)


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the affine map through three point pairs (6x6 system, LU, double)."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        A[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i]
    return np.linalg.solve(A, b).reshape(2, 3)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale=1.0, rot=0.0):
    """lib/datasets/utils.py:81-128 — centre, centre+down, centre+right of the box mapped onto the patch (float32 points)."""
    f32 = np.float32
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)

    def rot2(v):
        return np.array([v[0] * cs - v[1] * sn, v[0] * sn + v[1] * cs], dtype=f32)

    center = np.array([c_x, c_y], dtype=np.float64)
    down = rot2(np.array([0, src_height * scale * 0.5], dtype=f32))
    right = rot2(np.array([src_width * scale * 0.5, 0], dtype=f32))
    src = np.stack([center, center + down, center + right]).astype(f32)
    dc = np.array([dst_width * 0.5, dst_height * 0.5], dtype=f32)
    dst = np.stack([dc, dc + np.array([0, dst_height * 0.5], dtype=f32), dc + np.array([dst_width * 0.5, 0], dtype=f32)]).astype(f32)
    return get_affine_transform(src, dst)


def expand_to_aspect_ratio(input_shape, target_aspect_ratio=None):
    """lib/datasets/utils.py:14-33.  w, h are numpy float32 scalars and w_t, h_t Python ints; under the numpy the reference
    pins (1.23.1, legacy scalar promotion) `w * h_t / w_t` is evaluated in float64, so the expanded side — and with it the
    bbox size, the anti-alias sigma and the affine — is a double-precision function of the float32 box.  Python floats
    reproduce exactly that (verified against the reference's code run under numpy 1.26: tests/golden/crop_numpy1.npz)."""
    if target_aspect_ratio is None:
        return input_shape
    w, h = input_shape
    w_t, h_t = target_aspect_ratio
    if h / w < h_t / w_t:
        return np.array([w, max(float(w) * h_t / w_t, h)])
    return np.array([max(float(h) * w_t / h_t, w), h])


class Cropper:
    """Owns a thmr_cropper handle (device scratch for blurred regions)."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Cropper needs a GPU device: the crop kernels have no CPU fallback")
        # 'cuda' without an index means the CURRENT device (like Engine), not device 0
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.lib = _cabi.load()
        h = C.c_void_p()
        rc = self.lib.thmr_cropper_create(idx, C.byref(h))
        if rc != 0:
            raise _cabi.EngineError(f"thmr_cropper_create: {self.lib.thmr_cropper_last_error(None).decode()}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.thmr_cropper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_device(self, frame):
        """(H, W, 3) uint8 numpy array or tensor -> contiguous device tensor (one H2D copy per frame, shared by its crops)."""
        t = torch.as_tensor(np.ascontiguousarray(frame) if isinstance(frame, np.ndarray) else frame)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("frame must be (H, W, 3) uint8")
        return t.to(self.device).contiguous()

    def warp(self, frame, trans, sigmas=None, truncate=4.0, patch=256, mean=DEFAULT_MEAN, std=DEFAULT_STD, is_bgr=True, out=None):
        """frame (H,W,3) uint8; trans (n,2,3) forward affines as given to cv2.warpAffine; sigmas (n,) anti-alias sigma or 0.
        Returns (n,3,patch,patch) float32 on the device: gaussian -> warpAffine -> [::-1] -> CHW -> (x - 255 mean)/(255 std)."""
        fr = self.to_device(frame)
        trans = np.asarray(trans, dtype=np.float64).reshape(-1, 6)
        n = trans.shape[0]
        sig = np.zeros(n) if sigmas is None else np.asarray(sigmas, dtype=np.float64).reshape(n)
        descs = (_cabi.CropDesc * n)()
        for i in range(n):
            descs[i].M[:] = trans[i].tolist()
            descs[i].sigma, descs[i].truncate = float(sig[i]), float(truncate)
        m = (C.c_float * 3)(*[np.float32(255.0 * v) for v in mean])
        s = (C.c_float * 3)(*[np.float32(255.0 * v) for v in std])
        if out is None:
            out = torch.empty(n, 3, patch, patch, device=self.device, dtype=torch.float32)
        elif out.shape != (n, 3, patch, patch) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous (n,3,patch,patch) float32 tensor on the cropper's device")
        H, W = int(fr.shape[0]), int(fr.shape[1])
        with torch.cuda.device(self.device):
            rc = self.lib.thmr_cropper_run(self.h, C.c_void_p(fr.data_ptr()), H, W, W * 3, descs, n, int(patch), int(bool(is_bgr)), m, s,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise _cabi.EngineError(f"thmr_cropper_run error {rc}: {self.lib.thmr_cropper_last_error(self.h).decode()}")
        return out


class ViTDetDataset:
    """Mirror of lib/datasets/vitdet_dataset.py:16-88 (inference only).  cfg needs MODEL.IMAGE_SIZE / IMAGE_MEAN / IMAGE_STD and
    optionally MODEL.BBOX_SHAPE, exactly as the reference reads them."""

    def __init__(self, cfg, img_cv2, boxes, train=False, device="cuda:0", cropper=None, **kwargs):
        assert train is False, "ViTDetDataset is only for inference"
        self.cfg = cfg
        self.img_cv2 = img_cv2
        self.img_size = cfg.MODEL.IMAGE_SIZE
        self.mean_rgb, self.std_rgb = tuple(cfg.MODEL.IMAGE_MEAN), tuple(cfg.MODEL.IMAGE_STD)
        self.mean, self.std = 255.0 * np.array(self.mean_rgb), 255.0 * np.array(self.std_rgb)
        boxes = np.asarray(boxes).astype(np.float32).reshape(-1, 4)
        self.center = (boxes[:, 2:4] + boxes[:, 0:2]) / 2.0
        self.scale = (boxes[:, 2:4] - boxes[:, 0:2]) / 200.0
        self.personid = np.arange(len(boxes), dtype=np.int32)
        self._cropper, self._device = cropper, device       # the GPU handle is created on first use
        self._frame_dev = None
        get = cfg.MODEL.get if hasattr(cfg.MODEL, "get") else (lambda k, d=None: getattr(cfg.MODEL, k, d))
        self.bbox_shape = get("BBOX_SHAPE", None)

    def __len__(self):
        return len(self.personid)

    @property
    def cropper(self):
        if self._cropper is None:
            self._cropper = Cropper(self._device)
        return self._cropper

    def _params(self, idx):
        """vitdet_dataset.py:46-68: bbox size, anti-alias sigma, affine of crop idx."""
        center = self.center[idx].copy()
        bbox_size = expand_to_aspect_ratio(self.scale[idx] * 200, target_aspect_ratio=self.bbox_shape).max()
        # bbox_size is a float32 scalar; the reference's pinned numpy (1.23.1) promotes `bbox_size*1.0` to float64, so the
        # down-sampling factor, sigma and the gaussian weights are double-precision functions of that float32 value
        f = (float(bbox_size) / self.img_size) / 2.0
        sigma = (f - 1) / 2 if f > 1.1 else 0.0
        trans = gen_trans_from_patch_cv(center[0], center[1], bbox_size, bbox_size, self.img_size, self.img_size, 1.0, 0)
        return bbox_size, float(sigma), trans

    def _frame(self):
        if self._frame_dev is None:
            self._frame_dev = self.cropper.to_device(self.img_cv2)
        return self._frame_dev

    def _crops(self, idxs):
        ps = [self._params(i) for i in idxs]
        img = self.cropper.warp(self._frame(), np.stack([p[2] for p in ps]), [p[1] for p in ps], truncate=4.0, patch=self.img_size,
                                mean=self.mean_rgb, std=self.std_rgb, is_bgr=True)
        return img, ps

    def __getitem__(self, idx):
        img, ps = self._crops([idx])
        H, W = self.img_cv2.shape[:2]
        return {"img": img[0], "personid": int(self.personid[idx]), "box_center": self.center[idx].copy(), "box_size": ps[0][0],
                "img_size": 1.0 * np.array([W, H])}

    def batch(self, idxs=None):
        """All (or the given) crops of the frame in one GPU call, collated like torch's default_collate would."""
        idxs = list(range(len(self))) if idxs is None else list(idxs)
        img, ps = self._crops(idxs)
        H, W = self.img_cv2.shape[:2]
        dev = self.cropper.device
        return {"img": img,
                "personid": torch.as_tensor(self.personid[idxs].astype(np.int64), device=dev),
                "box_center": torch.as_tensor(self.center[idxs], device=dev),
                "box_size": torch.as_tensor(np.array([float(p[0]) for p in ps], dtype=np.float64), device=dev),   # float64, as collated under the pinned numpy
                "img_size": torch.as_tensor(np.tile(1.0 * np.array([W, H]), (len(idxs), 1)), device=dev)}


def crop_examples(cropper, cvimg, centers, sizes, patch=256, mean=DEFAULT_MEAN, std=DEFAULT_STD, use_skimage_antialias=False,
                  is_bgr=True):
    """The image part of `get_example` with do_augment=False (lib/datasets/utils.py:501-638) for n boxes of ONE frame:
    centers (n,2), sizes (n,2) = (width, height) of each box.  Returns ((n,3,patch,patch) device tensor, (n,2,3) affines)."""
    centers, sizes = np.asarray(centers, dtype=np.float64).reshape(-1, 2), np.asarray(sizes, dtype=np.float64).reshape(-1, 2)
    trans, sig = [], []
    for (cx, cy), (w, h) in zip(centers, sizes):
        s = 0.0
        if use_skimage_antialias:
            f = patch / (w * 1.0)                      # utils.py:585, as written in the reference
            if f > 1.1:
                s = (f - 1) / 2
        sig.append(s)
        trans.append(gen_trans_from_patch_cv(cx, cy, w, h, patch, patch, 1.0, 0))
    trans = np.stack(trans)
    return cropper.warp(cvimg, trans, sig, truncate=3.0, patch=patch, mean=mean, std=std, is_bgr=is_bgr), trans
