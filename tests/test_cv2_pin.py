"""Staged pin of the OpenCV boundary of crop preprocessing: cv2.getAffineTransform / cv2.warpAffine are
restated in oracle/crop_oracle.py (opencv-python is absent from this image => parity UNPINNED, bounded against an independent
bilinear warp in tests/test_crop.py).  oracle/gen_golden_cv2.py freezes the real library's outputs the day cv2 is importable;
these tests then hold the restatement (CPU) and crop.hip (GPU) to them bit for bit, and SKIP until then."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def _fixture():
    p = os.path.join(GOLDEN_DIR, "cv2_warp.npz")
    if not os.path.exists(p):
        pytest.skip("parity unpinned: tests/golden/cv2_warp.npz is absent (no opencv-python in this image; "
                    "oracle/gen_golden_cv2.py writes it the day one exists)")
    return np.load(p)


def test_staged_pin_script_refuses_a_stub():
    """Without the real cv2 the generator must exit non-zero and write nothing (a stubbed cv2 would 'pin' the restatement
    to itself)."""
    import subprocess
    import sys
    from conftest import ROOT
    try:
        import cv2  # noqa: F401
        pytest.skip("cv2 is importable here: run oracle/gen_golden_cv2.py and commit the fixture")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_cv2.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "UNPINNED" in (r.stdout + r.stderr)
    assert not os.path.exists(os.path.join(GOLDEN_DIR, "cv2_warp.npz"))


def test_oracle_vs_real_cv2_fixture():
    from oracle import crop_oracle as CO
    from oracle.gen_golden_crop import synthetic_frame
    g = _fixture()
    M = np.stack([CO.get_affine_transform(s, d) for s, d in zip(g["gat_src"], g["gat_dst"])])
    assert np.array_equal(M, g["gat_M"])
    frame = synthetic_frame()
    mine = np.stack([CO.warp_affine(frame, m, (256, 256)) for m in g["warp_M"]])
    assert np.array_equal(mine, g["warp_u8"])
    blurred = CO.gaussian_antialias(frame.astype(np.float64), 1.2, 4.0)
    f64 = np.stack([CO.warp_affine(blurred, m, (256, 256)) for m in g["warp_M"]])
    assert np.abs(f64[:, ::4, ::4] - g["warp_f64_sub"]).max() < 1e-9


@pytest.mark.gpu
def test_gpu_vs_real_cv2_fixture(built_lib, cuda_dev):
    import ctypes as C
    import torch
    from oracle.gen_golden_crop import synthetic_frame
    from tokenhmr_amd import _cabi
    from tokenhmr_amd.preprocess import Cropper
    g = _fixture()
    frame = torch.from_numpy(synthetic_frame()).to(cuda_dev)
    cr = Cropper(cuda_dev)
    n = len(g["warp_M"])
    descs = (_cabi.CropDesc * n)()
    for i, m in enumerate(g["warp_M"]):
        for k in range(6):
            descs[i].M[k] = float(m.reshape(-1)[k])
        descs[i].sigma, descs[i].truncate = 0.0, 4.0
    out = torch.empty(n, 3, 256, 256, device=cuda_dev)
    mean = (C.c_float * 3)(0, 0, 0)
    std = (C.c_float * 3)(1, 1, 1)
    H, W = frame.shape[:2]
    rc = cr.lib.thmr_cropper_run(cr.h, C.c_void_p(frame.data_ptr()), H, W, W * 3, descs, n, 256, 0, mean, std,
                                 C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    got = out.permute(0, 2, 3, 1).round().to(torch.uint8).cpu().numpy()
    assert np.array_equal(got, g["warp_u8"])
