"""GPU-side mirror of the reference's `Evaluator` (tokenhmr/lib/utils/pose_utils.py:145-275).

Same constructor arguments, `__call__(output, batch)`, `log()`, `get_metrics_dict()`, `get_imgnames()` and the same
metric arrays (`mode_mpjpe`, `mode_re`, `mode_pve`, millimetres) — but pelvis alignment, MPJPE, the batched 3x3-SVD
Procrustes and PVE run in libtokenhmr_hip.so (thmr_eval_pose / thmr_regress_joints) on the tensors the hot path just
produced, so only 3 floats per crop are copied to the host instead of (B,6890,3) vertices (pose_utils.py:139-143,246).
"""
import ctypes as C

from collections.abc import Mapping

import numpy as np
import torch

from . import _cabi


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_KP_CACHE = {}


def eval_pose_gpu(pred_joints, gt_joints, keypoint_list, pelvis_ind, pelvis_mode=0, pred_vertices=None, gt_vertices=None):
    """Returns (mpjpe_mm, re_mm, pve_mm or None) as CUDA float32 tensors of shape (B,)."""
    dev = pred_joints.device
    if dev.type != "cuda":
        raise RuntimeError("evaluator kernels run on the GPU only (no CPU fallback)")
    pj = pred_joints.detach().float().contiguous()
    gj = gt_joints.detach().float().contiguous()
    B, nj = pj.shape[0], pj.shape[1]
    if gj.shape[:2] != (B, nj) or pj.shape[2] != 3 or gj.shape[2] not in (3, 4):
        raise ValueError(f"bad joint shapes {tuple(pj.shape)} / {tuple(gj.shape)}")
    kpl = [int(k) for k in keypoint_list]
    if not kpl or min(kpl) < 0 or max(kpl) >= nj:
        # the reference indexes pred[:, keypoint_list] and raises here (pose_utils.py:225-226); the kernel must never read past nj
        raise IndexError(f"keypoint_list indices must lie in [0, {nj}) for {nj}-joint inputs, got min {min(kpl, default=None)} max {max(kpl, default=None)}")
    if not 0 <= int(pelvis_ind) < nj:
        raise IndexError(f"pelvis_ind {pelvis_ind} outside [0, {nj})")
    kp = _KP_CACHE.get((tuple(kpl), dev))               # the index list lives on the device once: no H2D copy per batch
    if kp is None:
        kp = _KP_CACHE[(tuple(kpl), dev)] = torch.as_tensor(kpl, dtype=torch.int32, device=dev)
    mp = torch.empty(B, device=dev, dtype=torch.float32)
    re = torch.empty(B, device=dev, dtype=torch.float32)
    pelv = torch.empty(B, 6, device=dev, dtype=torch.float32)
    pv = gv = pve = None
    nv = 0
    if pred_vertices is not None and gt_vertices is not None:
        pv = pred_vertices.detach().float().contiguous()
        gv = gt_vertices.detach().float().contiguous()
        nv = pv.shape[1]
        pve = torch.empty(B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _cabi.check(_cabi.load().thmr_eval_pose(_p(pj), _p(gj), nj, gj.shape[2], _p(kp), kp.numel(), int(pelvis_ind),
                                               int(pelvis_mode), _p(pv), _p(gv), nv, B, _p(mp), _p(re), _p(pve), _p(pelv), st))
    return mp, re, pve


def regress_joints_gpu(J, verts):
    """(nj,nv) @ (B,nv,3) -> (B,nj,3) on the GPU."""
    J = J.detach().float().contiguous()
    v = verts.detach().float().contiguous()
    B, nv = v.shape[0], v.shape[1]
    out = torch.empty(B, J.shape[0], 3, device=v.device, dtype=torch.float32)
    with torch.cuda.device(v.device):
        st = C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
        _cabi.check(_cabi.load().thmr_regress_joints(_p(J), _p(v), J.shape[0], nv, B, _p(out), st))
    return out


class _BatchMetrics(Mapping):
    """What `Evaluator.__call__` returns by default: the reference's {'mode_mpjpe': (B,) array, ...} — filled from the device on first
    access (eval.py:149 ignores the return value, so a batch normally never pays a device synchronisation for it).  A read-only Mapping,
    not a dict subclass (CPython's fast paths for dict subclasses — `dict(res)`, `res.copy()`, `{**res}`, `==` — bypass an overridden
    `__getitem__` and would hand out placeholder values).  It holds THIS batch's own (3, B) device result — not the evaluator — so its
    values are the batch's whatever `merge_evaluator` does to the evaluator's arrays later, and it keeps nothing else alive; every access
    returns a fresh copy, as the reference does (`pose_utils.py:246`).  `to_dict()` gives the reference's plain dict; an evaluator built
    with `eager_results=True` returns that plain dict from every call (one device synchronisation per batch, like the reference)."""

    _ROWS = ("mode_mpjpe", "mode_re", "mode_pve")

    def __init__(self, dev3, keys):
        self._dev3, self._keys, self._host = dev3, tuple(keys), None

    def _materialise(self):
        if self._host is None:
            self._host = self._dev3.cpu().numpy().astype(np.float64)       # the evaluator's arrays are float64 (np.zeros), like the reference's
            self._dev3 = None
        return self._host

    def __getitem__(self, k):
        if k not in self._keys:
            raise KeyError(k)
        return self._materialise()[self._ROWS.index(k)].copy()

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def to_dict(self):
        """a plain dict of fresh arrays: what the reference's Evaluator.__call__ returns (isinstance(..., dict), item assignment, pickling)"""
        return {k: self[k] for k in self._keys}

    def __repr__(self):
        return repr(self.to_dict())


class Evaluator:
    """The per-sample metric arrays (`mode_mpjpe`, `mode_re`, `mode_pve`; `metrics` names them) live on the host as in the reference, but
    they are FILLED lazily: `__call__` only enqueues the kernels and keeps the three (B,) device results; the copy to the host happens when
    somebody looks — `log()`, `get_metrics_dict()`, reading a metric array, a returned batch dict — or after `max_pending` batches.  The
    reference's loop (eval.py:144-149) therefore runs without a device synchronisation per batch (its own evaluator copies (B,) arrays
    to the host every batch, pose_utils.py:139-143,246)."""

    def __init__(self, dataset_length, keypoint_list, pelvis_ind, metrics=("mode_mpjpe", "mode_re", "model_pve"),
                 J_regressor_24_SMPL=None, dataset="", max_pending=64, eager_results=False):
        self.dataset_length = dataset_length
        self.keypoint_list = list(keypoint_list)
        self.pelvis_ind = pelvis_ind
        self.metrics = list(metrics)
        self.J_regressor_24_SMPL = J_regressor_24_SMPL
        self.dataset = dataset
        self._arrays = {m: np.zeros((dataset_length,)) for m in self.metrics}
        self._pending = []                       # (first sample, stacked (3, B) device tensor)
        self._max_pending = max(1, int(max_pending))
        self._eager = bool(eager_results)        # True: __call__ returns the reference's plain dict (a device synchronisation per batch)
        self.counter = 0
        self.imgnames = []

    def __getattr__(self, name):
        # the metric arrays as attributes (`evaluator.mode_mpjpe`, `hasattr(evaluator, 'mode_pve')`), up to date when read
        arrays = self.__dict__.get("_arrays")
        if arrays is not None and name in arrays:
            self._flush()
            return arrays[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        arrays = self.__dict__.get("_arrays")
        if arrays is not None and name in arrays:          # merge_evaluator (dist.py) replaces the arrays wholesale
            self._flush()
            arrays[name] = value
        else:
            object.__setattr__(self, name, value)

    def _flush(self):
        pending, self._pending = self._pending, []
        for lo, dev3 in pending:
            host = dev3.cpu().numpy()
            n = host.shape[1]
            for row, m in enumerate(("mode_mpjpe", "mode_re", "mode_pve")):
                if m in self._arrays:
                    self._arrays[m][lo:lo + n] = host[row]

    def log(self):
        if self.counter == 0:
            print("Evaluation has not started")
            return
        print(f"{self.counter} / {self.dataset_length} samples")
        for m in self.metrics:
            unit = "mm" if m in ("mode_mpjpe", "mode_re", "mode_pve") else ""
            print(f"{m}: {getattr(self, m)[:self.counter].mean(0)} {unit}")
        print("***")

    def get_metrics_dict(self):
        return {m: getattr(self, m)[:self.counter].mean() for m in self.metrics}

    def get_imgnames(self):
        return self.imgnames

    def __call__(self, output, batch):
        self.imgnames += list(batch.get("imgname", []))
        want_pve = "mode_pve" in self._arrays
        if "EMDB" in self.dataset:                                  # pose_utils.py:209-222
            gt_v, pred_v = batch["vertices"], output["pred_vertices"]
            gt_j = regress_joints_gpu(self.J_regressor_24_SMPL, gt_v)
            pred_j = regress_joints_gpu(self.J_regressor_24_SMPL, pred_v)
            mp, re, pve = eval_pose_gpu(pred_j, gt_j, self.keypoint_list, 0, 1, pred_v if want_pve else None,
                                        gt_v if want_pve else None)
        else:                                                        # pose_utils.py:223-243
            mp, re, pve = eval_pose_gpu(output["pred_keypoints_3d"], batch["keypoints_3d"], self.keypoint_list,
                                        self.pelvis_ind, 0, output["pred_vertices"] if want_pve else None,
                                        batch["vertices"] if want_pve else None)
        B = mp.shape[0]
        dev3 = torch.stack([mp, re, pve if pve is not None else torch.zeros_like(mp)], 0)
        self._pending.append((self.counter, dev3))
        res = _BatchMetrics(dev3, [m for m in ("mode_mpjpe", "mode_re", "mode_pve") if m in self._arrays])
        if self._eager:
            res = res.to_dict()
        self.counter += B
        if len(self._pending) >= self._max_pending:
            self._flush()
        return res
