"""Stand-alone SMPL model on the GPU.

Mirrors what the reference's dataset code does per sample on the CPU with `smplx.SMPL(gender=...)`
(tokenhmr/lib/datasets/image_dataset.py:151-164,254-270; emdb_dataset.py:184-199): axis-angle
`global_orient` (B,3) + `body_pose` (B,69) + `betas` (B,10) -> GT vertices — but batched, through the same LBS
kernels as the hot path, with the male / female constants held in their own `thmr_smpl` handle.
"""
import ctypes as C
import types

import torch

from . import _cabi


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class SMPL:
    def __init__(self, constants, max_batch=64, device="cuda:0"):
        self.lib = _cabi.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _cabi.EngineError("tokenhmr_amd.smpl.SMPL runs on a HIP device only")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.max_batch = int(max_batch)
        keys = ["v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J19_regressor"]
        ikeys = ["parents", "extra_verts", "joint_map"]
        ts = {k: constants[k].detach().float().contiguous().cpu() for k in keys}
        ts.update({k: constants[k].detach().to(torch.int32).contiguous().cpu() for k in ikeys})
        d = _cabi.SmplDesc(**{k: ts[k].data_ptr() for k in keys + ikeys}, on_device=0,
                           update_hips=1 if constants.get("update_hips", False) else 0)
        h = C.c_void_p(0)
        _cabi.check(self.lib.thmr_smpl_create(C.byref(d), self.max_batch, idx, C.byref(h)))
        self.h = h
        self.faces = constants.get("faces")

    def close(self):
        if getattr(self, "h", None):
            self.lib.thmr_smpl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, global_orient, body_pose, betas, pose2rot=True):
        """pose2rot=True: axis-angle (B,3)+(B,69) (smplx.SMPL); False: rotation matrices (B,1,3,3)+(B,23,3,3) (SMPLLayer)."""
        B = betas.shape[0]
        if pose2rot:
            pose = torch.cat([global_orient.reshape(B, 3), body_pose.reshape(B, 69)], dim=1)
        else:
            pose = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], dim=1)
        pose = pose.to(self.device, torch.float32).contiguous()
        betas = betas.to(self.device, torch.float32).contiguous()
        verts = torch.empty(B, 6890, 3, device=self.device, dtype=torch.float32)
        joints = torch.empty(B, 44, 3, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _cabi.check(self.lib.thmr_smpl_forward(self.h, _p(pose), 1 if pose2rot else 0, _p(betas), B, _p(verts), _p(joints), st))
        return types.SimpleNamespace(vertices=verts, joints=joints)

    __call__ = forward
