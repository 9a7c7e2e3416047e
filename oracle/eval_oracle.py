"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's evaluation arithmetic
(tokenhmr/lib/utils/pose_utils.py), the checker for tokenhmr_amd/evaluator.py and csrc/eval.hip.
Pinned against the reference's own functions via tests/golden/eval_small.npz (oracle/gen_golden_eval.py)."""
import torch


def compute_similarity_transform(S1, S2):
    """pose_utils.py:61-114 (torch.svd Procrustes, sign fix on the last singular vector)."""
    S1, S2 = S1.to(torch.float32), S2.to(torch.float32)
    B = S1.shape[0]
    S1, S2 = S1.permute(0, 2, 1), S2.permute(0, 2, 1)
    mu1, mu2 = S1.mean(dim=2, keepdim=True), S2.mean(dim=2, keepdim=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = (X1 ** 2).sum(dim=(1, 2))
    K = torch.matmul(X1, X2.permute(0, 2, 1))
    U, s, V = torch.svd(K)
    Vh = V.permute(0, 2, 1)
    Z = torch.eye(3).unsqueeze(0).repeat(B, 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.linalg.det(torch.matmul(U, Vh)))
    R = torch.matmul(torch.matmul(V, Z), U.permute(0, 2, 1))
    trace = torch.matmul(R, K).diagonal(offset=0, dim1=-1, dim2=-2).sum(dim=-1)
    scale = (trace / var1).unsqueeze(-1).unsqueeze(-1)
    t = mu2 - scale * torch.matmul(R, mu1)
    return (scale * torch.matmul(R, S1) + t).permute(0, 2, 1)


def eval_pose(pred_joints, gt_joints):
    """pose_utils.py:129-143 -> (mpjpe_mm, re_mm)."""
    mpjpe = torch.sqrt(((pred_joints - gt_joints) ** 2).sum(dim=-1)).mean(dim=-1)
    hat = compute_similarity_transform(pred_joints, gt_joints)
    re = torch.sqrt(((hat - gt_joints) ** 2).sum(dim=-1)).mean(dim=-1)
    return 1000 * mpjpe, 1000 * re


def evaluate_batch(pred_keypoints_3d, pred_vertices, gt_keypoints_3d4, gt_vertices, keypoint_list, pelvis_ind):
    """Evaluator.__call__, non-EMDB branch (pose_utils.py:223-247) -> (mpjpe, re, pve) in mm."""
    pk = pred_keypoints_3d.clone()
    gk = gt_keypoints_3d4[:, :, :-1].clone()
    pp, gp = pk[:, [pelvis_ind]], gk[:, [pelvis_ind]]
    pk, gk = pk - pp, gk - gp
    mp, re = eval_pose(pk[:, keypoint_list], gk[:, keypoint_list])
    pve = torch.sqrt((((pred_vertices - pp) - (gt_vertices - gp)) ** 2).sum(dim=-1)).mean(dim=-1) * 1000.0
    return mp, re, pve


def evaluate_batch_emdb(pred_vertices, gt_vertices, J24, keypoint_list):
    """Evaluator.__call__, EMDB branch (pose_utils.py:209-222)."""
    gk = torch.matmul(J24, gt_vertices)
    gp = (gk[:, [1]] + gk[:, [2]]) / 2.0
    pk = torch.matmul(J24, pred_vertices)
    pp = (pk[:, [1]] + pk[:, [2]]) / 2.0
    mp, re = eval_pose((pk - pp)[:, keypoint_list], (gk - gp)[:, keypoint_list])
    pve = torch.sqrt((((pred_vertices - pp) - (gt_vertices - gp)) ** 2).sum(dim=-1)).mean(dim=-1) * 1000.0
    return mp, re, pve
