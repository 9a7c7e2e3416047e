"""TEST INFRASTRUCTURE ONLY — a SECOND, independently derived SMPL forward (fp64 numpy), used to bound the
unpinned SMPL stage (smplx==0.1.28 is absent; `oracle.tokenhmr_oracle.smpl_forward` restates smplx's lbs.py).

It shares no code and no algebra with `oracle.tokenhmr_oracle.smpl_forward` or `csrc/lbs.hip`.  Those two follow smplx's
formulation: relative transforms chained in array order, `A_j = G_j - [0 | G_j J_j]` (the rest joint removed by a
subtraction), joints through precomputed `J_template / J_shapedirs`.  This file follows the SMPL paper instead
(Loper et al. 2015, eq. 2-4):

    t'_v = sum_j  w_vj * G_j(theta, J) * G_j(theta*, J)^-1 * [ T_v + B_S(beta)_v + B_P(theta)_v ; 1 ]
    G_j(theta, J) = prod over the ancestors a of j, root first, of [ R_a | J_a - J_parent(a) ; 0 1 ]

  * every world transform is built from its own explicit ancestor PATH (root -> j), found by walking `parents` upwards —
    not by a loop that relies on parents preceding children in the array;
  * the rest-pose chain G_j(theta*) is built the same way with identity rotations and inverted with `numpy.linalg.inv`
    (an explicit 4x4 inverse, not "subtract the joint");
  * joints are `J_regressor @ v_shaped`, un-precomputed, per sample;
  * skinning is a per-vertex weighted sum of 4x4 matrices applied to homogeneous points.

Output convention = the reference wrapper, tokenhmr/lib/models/smpl_wrapper.py:27-41: 25 mapped joints (24 posed joints +
21 extra vertices, permuted by joint_map), optional update_hips, then the 19 regressed extra joints.
"""
import numpy as np


def _ancestor_path(parents, j):
    path = [j]
    while parents[path[-1]] >= 0:
        path.append(int(parents[path[-1]]))
    return path[::-1]                                        # root first


def _world(parents, R, J, j):
    """4x4 world transform of joint j for ONE sample: product of the local transforms along root -> j."""
    G = np.eye(4)
    for a in _ancestor_path(parents, j):
        L = np.eye(4)
        L[:3, :3] = R[a]
        p = int(parents[a])
        L[:3, 3] = J[a] - (J[p] if p >= 0 else 0.0)
        G = G @ L
    return G


def smpl_forward_independent(rotmat, betas, smpl):
    """rotmat (B,24,3,3), betas (B,10), smpl: dict of constants (tokenhmr_amd.smpl_assets layout) -> verts (B,6890,3),
    joints (B,44,3), float64."""
    f = lambda k: np.asarray(smpl[k], dtype=np.float64)   # noqa: E731
    vt, sd, pd = f("v_template"), f("shapedirs"), f("posedirs")
    Jreg, W, J19 = f("J_regressor"), f("lbs_weights"), f("J19_regressor")
    parents = np.asarray(smpl["parents"], dtype=np.int64)
    extra = np.asarray(smpl["extra_verts"], dtype=np.int64)
    jmap = np.asarray(smpl["joint_map"], dtype=np.int64)
    R_all = np.asarray(rotmat, dtype=np.float64)
    betas = np.asarray(betas, dtype=np.float64)
    B, NJ, V = betas.shape[0], parents.shape[0], vt.shape[0]
    verts = np.zeros((B, V, 3))
    joints = np.zeros((B, 25 + J19.shape[0], 3))
    I3 = np.eye(3)
    for b in range(B):
        R = R_all[b]
        v_shaped = vt + sd @ betas[b]                                        # T + B_S(beta)
        J = Jreg @ v_shaped                                                  # rest joints of THIS shape
        theta_feat = np.concatenate([(R[k] - I3).reshape(-1) for k in range(1, NJ)])
        v_posed = v_shaped + (theta_feat @ pd).reshape(V, 3)                 # + B_P(theta)
        Gp = np.stack([_world(parents, R, J, j) for j in range(NJ)])         # posed chain
        Gr = np.stack([_world(parents, np.broadcast_to(I3, (NJ, 3, 3)), J, j) for j in range(NJ)])   # rest chain
        Gprime = np.stack([Gp[j] @ np.linalg.inv(Gr[j]) for j in range(NJ)])                      # eq. (3)
        Tv = np.einsum("vj,jrc->vrc", W, Gprime)                              # per-vertex blended 4x4
        vh = np.concatenate([v_posed, np.ones((V, 1))], axis=1)
        out = np.einsum("vrc,vc->vr", Tv, vh)
        verts[b] = out[:, :3] / out[:, 3:4]                                   # w == sum_j w_vj (== 1 for real weights)
        j45 = np.concatenate([Gp[:, :3, 3], verts[b][extra]], axis=0)
        jm = j45[jmap].copy()
        if smpl.get("update_hips", False):                                    # smpl_wrapper.py:33-36
            a, c, m = jm[9].copy(), jm[12].copy(), jm[8].copy()
            jm[9] = a + 0.25 * (a - c) + 0.5 * (m - 0.5 * (a + c))
            jm[12] = c + 0.25 * (c - a) + 0.5 * (m - 0.5 * (c + a))
        joints[b] = np.concatenate([jm, J19 @ verts[b]], axis=0)
    return verts, joints


def random_rotations(n, seed=0, scale=1.0):
    """(n,3,3) proper rotations from axis-angle vectors ~ scale * N(0,1), via the matrix exponential series in fp64
    (yet another route than Rodrigues' closed form, so it shares nothing with the kernels under test)."""
    rng = np.random.default_rng(seed)
    aa = scale * rng.standard_normal((n, 3))
    out = np.zeros((n, 3, 3))
    for i, (x, y, z) in enumerate(aa):
        K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
        term, acc = np.eye(3), np.eye(3)
        for k in range(1, 40):
            term = term @ K / k
            acc = acc + term
        out[i] = acc
    return out
