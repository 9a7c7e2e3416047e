"""TEST INFRASTRUCTURE ONLY — golden metrics from the reference's OWN Evaluator (pose_utils.py imported in place
with a cv2 stub).  Run in the build container:  python oracle/gen_golden_eval.py"""
import importlib.util, os, sys, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/tokenhmr/lib/utils"


def load_pose_utils():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pkg = types.ModuleType("_ref_utils"); pkg.__path__ = [REF]; sys.modules["_ref_utils"] = pkg
    for name in ("rotation_utils", "pose_utils"):
        spec = importlib.util.spec_from_file_location(f"_ref_utils.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec); mod.__package__ = "_ref_utils"
        sys.modules[f"_ref_utils.{name}"] = mod; spec.loader.exec_module(mod)
    return sys.modules["_ref_utils.pose_utils"]


def make_case(B=6, seed=0):
    g = torch.Generator().manual_seed(5000 + seed)
    gt_v = 0.3 * torch.randn(B, 6890, 3, generator=g)
    pred_v = gt_v + 0.02 * torch.randn(B, 6890, 3, generator=g)
    gt_k = torch.cat([0.3 * torch.randn(B, 44, 3, generator=g), torch.ones(B, 44, 1)], -1)
    # predictions = rotated/scaled/translated GT + noise (so PA-MPJPE << MPJPE), one crop mirrored (det < 0 branch)
    from oracle.tokenhmr_oracle import rot6d_to_rotmat
    R = rot6d_to_rotmat(torch.randn(B, 6, generator=g))
    pred_k = 1.1 * torch.einsum("bij,bkj->bki", R, gt_k[:, :, :3]) + 0.05 + 0.01 * torch.randn(B, 44, 3, generator=g)
    pred_k[-1, :, 0] *= -1.0
    J24 = torch.softmax(4 * torch.randn(24, 6890, generator=g), dim=1)
    return pred_k, pred_v, gt_k, gt_v, J24


def main():
    pu = load_pose_utils()
    kp = [25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 43]       # datasets_eval.yaml:12 (3DPW-TEST)
    pred_k, pred_v, gt_k, gt_v, J24 = make_case()
    ev = pu.Evaluator(dataset_length=100, keypoint_list=kp, pelvis_ind=39, metrics=["mode_re", "mode_mpjpe", "mode_pve"], dataset="3DPW-TEST")
    ev({"pred_keypoints_3d": pred_k.clone(), "pred_vertices": pred_v.clone()},
       {"imgname": ["x"] * pred_k.shape[0], "keypoints_3d": gt_k.clone(), "vertices": gt_v.clone()})
    n = pred_k.shape[0]
    out = {"mpjpe": ev.mode_mpjpe[:n], "re": ev.mode_re[:n], "pve": ev.mode_pve[:n]}
    ev2 = pu.Evaluator(dataset_length=100, keypoint_list=list(range(24)), pelvis_ind=39, metrics=["mode_re", "mode_mpjpe", "mode_pve"],
                       J_regressor_24_SMPL=J24, dataset="EMDB")
    ev2({"pred_vertices": pred_v.clone()}, {"imgname": ["x"] * n, "vertices": gt_v.clone()})
    out.update({"emdb_mpjpe": ev2.mode_mpjpe[:n], "emdb_re": ev2.mode_re[:n], "emdb_pve": ev2.mode_pve[:n]})
    from oracle import eval_oracle as E
    mp, re, pve = E.evaluate_batch(pred_k, pred_v, gt_k, gt_v, kp, 39)
    print("oracle vs reference:", abs(mp.numpy() - out["mpjpe"]).max(), abs(re.numpy() - out["re"]).max(), abs(pve.numpy() - out["pve"]).max())
    mp, re, pve = E.evaluate_batch_emdb(pred_v, gt_v, J24, list(range(24)))
    print("oracle vs reference (EMDB):", abs(mp.numpy() - out["emdb_mpjpe"]).max(), abs(re.numpy() - out["emdb_re"]).max(), abs(pve.numpy() - out["emdb_pve"]).max())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eval_small.npz"), **{k: np.asarray(v, dtype=np.float64) for k, v in out.items()})
    print({k: np.round(v, 3) for k, v in out.items()})


if __name__ == "__main__":
    main()
