"""SMPL model constants for the LBS stage.

No SMPL .pkl exists offline (license-gated), so tests and the bench use seeded
*synthetic, shape- and structure-faithful* constants:
  v_template (6890,3), shapedirs (6890,3,10), posedirs (207,20670),
  J_regressor (24,6890) and joint_regressor_extra "J19" (19,6890) with non-negative
  rows summing to 1, lbs_weights (6890,24) rows summing to 1, the real SMPL
  kinematic tree and the real smplx extra-joint vertex ids.
`load_smpl_pkl` reads a real SMPL_NEUTRAL.pkl / SMPL_to_J19.pkl when a user has
them (call sites it replaces: tokenhmr/lib/models/smpl_wrapper.py:11-25).
"""
import pickle
import numpy as np
import torch

from .config import HMRConfig, RELEASE, SMPL_PARENTS, SMPL_EXTRA_VERTS, SMPL_TO_OPENPOSE


def make_synthetic_smpl(cfg: HMRConfig = RELEASE, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(3000 + seed)
    V, J = cfg.n_verts, cfg.n_joints

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32)

    a = {}
    a["v_template"] = 0.3 * randn(V, 3)
    a["shapedirs"] = 0.01 * randn(V, 3, cfg.n_betas)
    a["posedirs"] = 0.001 * randn(cfg.n_posedirs, V * 3)
    a["J_regressor"] = torch.softmax(4.0 * randn(J, V), dim=1)
    a["lbs_weights"] = torch.softmax(4.0 * randn(V, J), dim=1)
    a["J19_regressor"] = torch.softmax(4.0 * randn(cfg.n_j19, V), dim=1)
    a["parents"] = torch.tensor(SMPL_PARENTS, dtype=torch.int32)
    a["extra_verts"] = torch.tensor(SMPL_EXTRA_VERTS, dtype=torch.int32)
    a["joint_map"] = torch.tensor(SMPL_TO_OPENPOSE, dtype=torch.int32)
    a["faces"] = torch.zeros(13776, 3, dtype=torch.int64)  # placeholder topology (renderer-only)
    return a


class _ChumpyStub:
    """Unpickle chumpy arrays without chumpy: keep only the ndarray payload."""
    def __setstate__(self, state):
        self.__dict__.update(state)

    def to_numpy(self):
        for key in ("x", "a", "r"):
            if key in self.__dict__:
                return np.asarray(self.__dict__[key])
        raise ValueError("unrecognised chumpy payload")


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChumpyStub
        return super().find_class(module, name)


def _np(x):
    if isinstance(x, _ChumpyStub):
        x = x.to_numpy()
    if hasattr(x, "toarray"):      # scipy sparse J_regressor
        x = x.toarray()
    return np.asarray(x)


def load_smpl_pkl(model_pkl: str, j19_pkl: str, cfg: HMRConfig = RELEASE):
    """Real SMPL constants -> the same dict layout as make_synthetic_smpl()."""
    with open(model_pkl, "rb") as f:
        d = _Unpickler(f, encoding="latin1").load()
    with open(j19_pkl, "rb") as f:
        j19 = pickle.load(f, encoding="latin1")
    V = cfg.n_verts
    a = {}
    a["v_template"] = torch.from_numpy(_np(d["v_template"]).astype(np.float32))
    a["shapedirs"] = torch.from_numpy(_np(d["shapedirs"])[:, :, :cfg.n_betas].astype(np.float32))
    pd = _np(d["posedirs"]).astype(np.float32)             # (6890,3,207)
    a["posedirs"] = torch.from_numpy(pd.reshape(V * 3, -1).T.copy())   # smplx: (207, 20670)
    a["J_regressor"] = torch.from_numpy(_np(d["J_regressor"]).astype(np.float32))
    a["lbs_weights"] = torch.from_numpy(_np(d["weights"]).astype(np.float32))
    a["J19_regressor"] = torch.from_numpy(np.asarray(j19, dtype=np.float32))
    kt = _np(d["kintree_table"]).astype(np.int64)
    parents = kt[0].copy()
    parents[0] = -1
    a["parents"] = torch.from_numpy(parents.astype(np.int32))
    a["extra_verts"] = torch.tensor(SMPL_EXTRA_VERTS, dtype=torch.int32)
    a["joint_map"] = torch.tensor(SMPL_TO_OPENPOSE, dtype=torch.int32)
    a["faces"] = torch.from_numpy(_np(d["f"]).astype(np.int64))
    return a
