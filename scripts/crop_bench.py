"""Throughput of the GPU crop preprocessing (N2) on a 1080p frame vs the CPU restatement of the reference's per-crop code.
    python scripts/crop_bench.py [n_boxes]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tokenhmr_amd.preprocess import ViTDetDataset, Cropper


class N(dict):
    __getattr__ = dict.__getitem__


cfg = N(MODEL=N(IMAGE_SIZE=256, IMAGE_MEAN=[0.485, 0.456, 0.406], IMAGE_STD=[0.229, 0.224, 0.225], BBOX_SHAPE=[192, 256]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
H, W = 1080, 1920
frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
res = {}
for name, (lo, hi) in {"small boxes (no blur)": (150, 500), "large boxes (anti-alias blur)": (600, 1000)}.items():
    hgt = rng.uniform(lo, hi, n)
    cx, cy = rng.uniform(300, W - 300, n), rng.uniform(300, H - 300, n)
    boxes = np.stack([cx - hgt * 0.2, cy - hgt / 2, cx + hgt * 0.2, cy + hgt / 2], 1)
    cr = Cropper("cuda:0")
    ds = ViTDetDataset(cfg, frame, boxes, device="cuda:0", cropper=cr)
    ds.batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ds._frame_dev = None            # include the H2D copy of the frame (6.2 MB) every time
        ds.batch()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ds.batch()
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / reps
    from oracle import crop_oracle as CO
    t0 = time.perf_counter()
    k = min(n, 3)
    for b in boxes[:k]:
        CO.vitdet_item(frame, b, 256, [192, 256])
    t_cpu = (time.perf_counter() - t0) / k
    res[name] = {"n_crops": n, "gpu_ms_per_frame_incl_h2d": round(t_all * 1e3, 3), "gpu_ms_per_frame_resident": round(t_res * 1e3, 3),
                 "gpu_crops_per_s_incl_h2d": round(n / t_all, 1), "cpu_oracle_ms_per_crop": round(t_cpu * 1e3, 1),
                 "cpu_crops_per_s_1core": round(1 / t_cpu, 2)}
    print(name, res[name], flush=True)
print(json.dumps(res))
