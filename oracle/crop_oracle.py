"""TEST INFRASTRUCTURE ONLY — CPU restatement of the crop preprocessing that feeds the hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path never does.

What the reference does (tokenhmr/lib/datasets/vitdet_dataset.py:44-88 for demo.py; lib/datasets/utils.py:501-638
`get_example` for eval.py) and what it is restated from:

  * box -> affine:  `gen_trans_from_patch_cv`   lib/datasets/utils.py:81-128   (restated line by line, numpy)
                    + cv2.getAffineTransform      THIRD PARTY (opencv-python, requirements: not vendored, absent here)
  * anti-alias:     skimage.filters.gaussian      THIRD PARTY (scikit-image); it is a thin wrapper of
                    scipy.ndimage.gaussian_filter(mode='nearest', truncate=4.0, sigma=(s, s, 0)) on the float64 image.
                    The system python has scipy but not scikit-image; the image's second interpreter
                    (/opt/conda/bin/python3.9: numpy 1.26, scikit-image 0.18.3) has both, and the reference's dataset
                    code run THERE with the real skimage gives tensors this module reproduces bit for bit
                    (oracle/gen_golden_crop_numpy1.py -> tests/golden/crop_numpy1.npz): PINNED.
  * numpy version:  the reference pins numpy==1.23.1 (legacy value-based promotion).  Three expressions of the crop code
                    change precision under numpy >= 2 (float32 box arithmetic in expand_to_aspect_ratio, float32 sigma,
                    float64 normalisation); `numpy1=True` restates the pinned behaviour and is what the fixture above pins.
  * warp:           cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT)   THIRD PARTY, absent here.  `warp_affine` below restates
                    OpenCV 4.x imgwarp.cpp: inverse of the 2x3 matrix in double; fixed-point source coordinates with
                    AB_BITS = 10 and INTER_BITS = 5 (X = (rint((M1*y+M2)*1024) + 16 + rint(M0*x*1024)) >> 5, integer part
                    X >> 5, 5-bit fraction X & 31); uint8 images: bilinear weights 32*a*b (a, b in 0..32, sum 2^15) and
                    (sum + 2^14) >> 15; float64 images (after the blur): float weights a*b/1024, sum in double;
                    BORDER_CONSTANT = zero padding of the four neighbours.
  * layout:         BGR->RGB flip, HWC->CHW float32, (x - 255*mean) / (255*std)   vitdet_dataset.py:75-80

PARITY UNPINNED for the two cv2 primitives: opencv is not installed in this image and the reference holds no golden
crops, so `warp_affine` / `get_affine_transform` are pinned only to the published algorithm.  Everything around them IS
pinned: oracle/gen_golden_crop.py runs the reference's own ViTDetDataset / generate_image_patch_cv2 code in place with a
`cv2` stub that forwards to these two functions and `skimage.filters.gaussian` forwarded to scipy, and the results equal
`vitdet_item` / `example_item` below bit for bit.
"""
import numpy as np
from scipy import ndimage

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
COEF_BITS = 15


def get_affine_transform(src, dst):
    """cv2.getAffineTransform (imgwarp.cpp): the 6x6 system [x y 1 0 0 0; 0 0 0 x y 1] X = [u; v] solved by LU in double."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        A[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def rotate_2d(pt_2d, rot_rad):
    """lib/datasets/utils.py:66-79"""
    x, y = pt_2d[0], pt_2d[1]
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([x * cs - y * sn, x * sn + y * cs], dtype=np.float32)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot):
    """lib/datasets/utils.py:81-128 (same dtypes at every step)"""
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.zeros(2)
    src_center[0], src_center[1] = c_x, c_y
    rot_rad = np.pi * rot / 180
    src_downdir = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    src_rightdir = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst_downdir = np.array([0, dst_height * 0.5], dtype=np.float32)
    dst_rightdir = np.array([dst_width * 0.5, 0], dtype=np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :], src[1, :], src[2, :] = src_center, src_center + src_downdir, src_center + src_rightdir
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :], dst[1, :], dst[2, :] = dst_center, dst_center + dst_downdir, dst_center + dst_rightdir
    return get_affine_transform(np.float32(src), np.float32(dst))


def invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (no WARP_INVERSE_MAP)."""
    M = np.array(M, dtype=np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def source_coords(M, dsize):
    """Fixed-point source coordinates of every destination pixel (WarpAffineInvoker): integer parts and 5-bit fractions."""
    w, h = dsize
    Mi = invert_affine(M)
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    adelta = np.rint(Mi[0] * x * AB_SCALE).astype(np.int64)
    bdelta = np.rint(Mi[3] * x * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = np.rint((Mi[1] * y + Mi[2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((Mi[4] * y + Mi[5]) * AB_SCALE).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)          # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    return sx, sy, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def warp_affine(img, M, dsize):
    """cv2.warpAffine(img, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for (H, W, C) uint8 or
    float64 images."""
    H, W = img.shape[:2]
    sx, sy, fx, fy = source_coords(M, dsize)
    pad = np.zeros((H + 2, W + 2) + img.shape[2:], dtype=img.dtype)        # zero border: index -1 and H / W
    pad[1:-1, 1:-1] = img

    def at(yy, xx):
        inside = (yy >= -1) & (yy <= H) & (xx >= -1) & (xx <= W)
        v = pad[np.clip(yy, -1, H) + 1, np.clip(xx, -1, W) + 1]
        return np.where(inside[..., None], v, 0)

    v00, v01, v10, v11 = at(sy, sx), at(sy, sx + 1), at(sy + 1, sx), at(sy + 1, sx + 1)
    ax1, ay1 = fx[..., None], fy[..., None]
    ax0, ay0 = INTER_TAB_SIZE - ax1, INTER_TAB_SIZE - ay1
    if img.dtype == np.uint8:
        w00, w01, w10, w11 = (32 * ay0 * ax0, 32 * ay0 * ax1, 32 * ay1 * ax0, 32 * ay1 * ax1)     # shorts, sum 2^15
        acc = v00.astype(np.int64) * w00 + v01.astype(np.int64) * w01 + v10.astype(np.int64) * w10 + v11.astype(np.int64) * w11
        return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)
    # float / double images: float table (1-fy)(1-fx) ..., products exact in float32; accumulation in double, left to right
    t = np.float32(1.0 / INTER_TAB_SIZE)
    fx1, fy1 = (ax1.astype(np.float32) * t), (ay1.astype(np.float32) * t)
    fx0, fy0 = np.float32(1) - fx1, np.float32(1) - fy1
    w00, w01, w10, w11 = [(a * b).astype(np.float64) for a, b in ((fy0, fx0), (fy0, fx1), (fy1, fx0), (fy1, fx1))]
    return ((v00.astype(np.float64) * w00 + v01 * w01) + v10 * w10) + v11 * w11


def gaussian_antialias(img, sigma, truncate=4.0):
    """skimage.filters.gaussian(img, sigma=sigma, channel_axis=2, preserve_range=True[, truncate]) == scipy on float64."""
    return ndimage.gaussian_filter(img.astype(np.float64), [sigma, sigma, 0], mode="nearest", cval=0, truncate=truncate)


def expand_to_aspect_ratio(input_shape, target_aspect_ratio=None, numpy1=False):
    """lib/datasets/utils.py:14-33.  w, h arrive as numpy float32 scalars and w_t, h_t as Python ints: the pinned numpy 1.23
    promotes `w * h_t / w_t` to float64 (legacy scalar promotion), numpy >= 2 keeps float32.  numpy1=True restates the pinned
    behaviour with Python floats (same IEEE double operations)."""
    if target_aspect_ratio is None:
        return input_shape
    w, h = input_shape
    w_t, h_t = target_aspect_ratio
    wd, hd = (float(w), float(h)) if numpy1 else (w, h)
    if h / w < h_t / w_t:
        h_new, w_new = max(wd * h_t / w_t, h), w
    else:
        h_new, w_new = h, max(hd * w_t / h_t, w)
    return np.array([w_new, h_new])


def finish_patch(patch_cv, mean, std, is_bgr=True, numpy1=True, clip=False):
    """flip, CHW float32, normalise (vitdet_dataset.py:75-80 / utils.py:606-617).  The reference pins numpy==1.23.1
    (requirements.txt:1), whose value-based promotion keeps `float32_array - float64_scalar` in float32; numpy >= 2 (this
    image) computes it in float64 and rounds once on assignment.  numpy1=True restates the pinned behaviour (what the HIP
    kernel implements); numpy1=False is the expression as this image's numpy evaluates it (used to check the restatement
    against the reference code executed here).  The two differ by at most 1 float32 ulp."""
    p = patch_cv[:, :, ::-1] if is_bgr else patch_cv
    img = np.transpose(p.copy(), (2, 0, 1)).astype(np.float32)
    for c in range(min(patch_cv.shape[2], 3)):
        if clip:
            img[c, :, :] = np.clip(img[c, :, :] * 1.0, 0, 255)
        if numpy1:
            img[c, :, :] = (img[c, :, :] - np.float32(mean[c])) / np.float32(std[c])
        else:
            img[c, :, :] = (img[c, :, :] - mean[c]) / std[c]
    return img


def vitdet_item(img_cv2, box, img_size=256, bbox_shape=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), numpy1=True):
    """One item of ViTDetDataset (vitdet_dataset.py:16-88) for one xyxy box.  Returns the dict the DataLoader collates."""
    mean, std = 255.0 * np.array(mean), 255.0 * np.array(std)
    box = np.asarray(box).astype(np.float32)
    center = (box[2:4] + box[0:2]) / 2.0
    scale = (box[2:4] - box[0:2]) / 200.0
    bbox_size = expand_to_aspect_ratio(scale * 200, target_aspect_ratio=bbox_shape, numpy1=numpy1).max()
    cvimg = img_cv2.copy()
    # vitdet_dataset.py:64-65.  bbox_size is a numpy float32 scalar: under the pinned numpy 1.23 `bbox_size*1.0` is a float64
    # and sigma / the gaussian weights are computed in double; numpy >= 2 keeps float32 (and scipy then squares sigma in
    # float32), which moves the weights by ~1e-8 relative.  numpy1=True restates the pinned behaviour.
    f = (float(bbox_size) if numpy1 else bbox_size * 1.0) / img_size / 2.0
    sigma = 0.0
    if f > 1.1:
        sigma = (f - 1) / 2
        cvimg = gaussian_antialias(cvimg, sigma)
    trans = gen_trans_from_patch_cv(center[0], center[1], bbox_size, bbox_size, img_size, img_size, 1.0, 0)
    patch = warp_affine(cvimg, trans, (int(img_size), int(img_size)))
    return {"img": finish_patch(patch, mean, std, numpy1=numpy1), "box_center": center.copy(), "box_size": bbox_size,
            "img_size": 1.0 * np.array([cvimg.shape[1], cvimg.shape[0]]), "trans": trans, "sigma": float(sigma)}


def example_item(cvimg, center_x, center_y, width, height, patch=256, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                 use_skimage_antialias=False, is_bgr=True, numpy1=True):
    """`get_example` without augmentation (lib/datasets/utils.py:501-638 with do_augment=False): the eval.py crop."""
    mean, std = 255.0 * np.array(mean), 255.0 * np.array(std)
    sigma = 0.0
    if use_skimage_antialias:
        f = patch / (width * 1.0)                      # utils.py:585 (as written in the reference)
        if f > 1.1:
            sigma = (f - 1) / 2
            cvimg = gaussian_antialias(cvimg, sigma, truncate=3.0)
    trans = gen_trans_from_patch_cv(center_x, center_y, width, height, patch, patch, 1.0, 0)
    p = warp_affine(cvimg, trans, (int(patch), int(patch)))
    img = finish_patch(p, mean, std, is_bgr, numpy1=numpy1, clip=True)
    return {"img": img, "trans": trans, "sigma": float(sigma)}
