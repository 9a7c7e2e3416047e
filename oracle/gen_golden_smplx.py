"""TEST INFRASTRUCTURE ONLY — the STAGED pin of the SMPL boundary.

    python oracle/gen_golden_smplx.py [--smpl-pkl SMPL_NEUTRAL.pkl --j19-pkl SMPL_to_J19.pkl]
        -> tests/golden/smplx_lbs.npz          (needs `import smplx` to be the REAL package, smplx==0.1.28)

The arithmetic of this stage lives in the un-vendored third-party package smplx (tokenhmr/requirements.txt:3); it is absent
from this image, so `oracle.tokenhmr_oracle.smpl_forward` / `batch_rodrigues` are restatements and parity there is
UNPINNED.  This script is what flips it: the day a smplx wheel is importable it
  1. writes the seeded SYNTHETIC constants of tokenhmr_amd.smpl_assets.make_synthetic_smpl as a temporary SMPL_NEUTRAL.pkl /
     SMPL_to_J19.pkl (the file layout smplx.SMPLLayer reads) — so no licence-gated file is needed and the fixture is
     self-contained (constants regenerate from the seed); with --smpl-pkl/--j19-pkl the real model is used as well and only
     inputs / outputs / a checksum of the constants are stored (`smplx_lbs_real.npz`, constants are never committed);
  2. imports the reference's OWN wrapper tokenhmr/lib/models/smpl_wrapper.py in place and runs
     SMPL(model_path, joint_regressor_extra=..., update_hips=False/True)(global_orient, body_pose, betas, pose2rot=False)
     exactly as tokenhmr.py:176 does, plus smplx.lbs.batch_rodrigues on seeded axis-angles (the GT-side path, N3);
  3. freezes inputs and outputs.
tests/test_smplx_pin.py then holds the oracle (CPU) and the HIP kernels (GPU) to that fixture; while the fixture is
absent those tests SKIP with the reason "parity unpinned".

`--plumbing-check` runs the same script against a stand-in `smplx` module built from the oracle's restatement, which proves
the script and the reference wrapper's part (joint_map, update_hips, vertices2joints concat; smpl_wrapper.py:27-41) without
pinning lbs itself; it writes nothing under tests/golden/.
"""
import argparse
import importlib.util
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("TOKENHMR_REFERENCE", "/root/reference")

from tokenhmr_amd.smpl_assets import make_synthetic_smpl, load_smpl_pkl  # noqa: E402
from oracle.lbs_independent import random_rotations  # noqa: E402


def write_synthetic_pkls(smpl, d):
    """The on-disk layout smplx reads: SMPL_NEUTRAL.pkl with v_template, shapedirs (6890,3,10), posedirs (6890,3,207),
    J_regressor, weights, kintree_table (2,24), f; and the J19 regressor pickle."""
    V = smpl["v_template"].shape[0]
    parents = smpl["parents"].numpy().astype(np.int64)
    kt = np.stack([parents, np.arange(len(parents))]).astype(np.int64)
    kt[0, 0] = 2 ** 32 - 1                                   # how the released files mark the root
    model = {"v_template": smpl["v_template"].numpy().astype(np.float64),
             "shapedirs": smpl["shapedirs"].numpy().astype(np.float64),
             "posedirs": smpl["posedirs"].numpy().T.reshape(V, 3, -1).astype(np.float64),
             "J_regressor": smpl["J_regressor"].numpy().astype(np.float64),
             "weights": smpl["lbs_weights"].numpy().astype(np.float64),
             "kintree_table": kt, "f": np.zeros((13776, 3), dtype=np.int64)}
    os.makedirs(os.path.join(d, "smpl"), exist_ok=True)
    mp = os.path.join(d, "smpl", "SMPL_NEUTRAL.pkl")
    with open(mp, "wb") as f:
        pickle.dump(model, f)
    jp = os.path.join(d, "SMPL_to_J19.pkl")
    with open(jp, "wb") as f:
        pickle.dump(smpl["J19_regressor"].numpy().astype(np.float64), f)
    return mp, jp


def install_standin_smplx():
    """--plumbing-check only: a module with the three names smpl_wrapper.py imports, backed by the ORACLE's restated lbs."""
    from oracle import tokenhmr_oracle as O
    from tokenhmr_amd.config import SMPL_EXTRA_VERTS
    smplx = types.ModuleType("smplx")
    lbs = types.ModuleType("smplx.lbs")
    utils = types.ModuleType("smplx.utils")

    class SMPLOutput(types.SimpleNamespace):
        pass

    class SMPLLayer(torch.nn.Module):
        def __init__(self, model_path, gender="neutral", **kw):
            super().__init__()
            pkl = model_path if model_path.endswith(".pkl") else os.path.join(model_path, f"SMPL_{gender.upper()}.pkl")
            with open(pkl, "rb") as f:
                d = pickle.load(f, encoding="latin1")
            V = d["v_template"].shape[0]
            self.c = {"v_template": torch.tensor(d["v_template"], dtype=torch.float32),
                      "shapedirs": torch.tensor(d["shapedirs"][:, :, :10], dtype=torch.float32),
                      "posedirs": torch.tensor(d["posedirs"].reshape(V * 3, -1).T.copy(), dtype=torch.float32),
                      "J_regressor": torch.tensor(d["J_regressor"], dtype=torch.float32),
                      "lbs_weights": torch.tensor(d["weights"], dtype=torch.float32),
                      "parents": torch.tensor(np.where(np.arange(24) == 0, -1, d["kintree_table"][0].astype(np.int64))),
                      "extra_verts": torch.tensor(SMPL_EXTRA_VERTS), "joint_map": torch.arange(45),
                      "J19_regressor": torch.zeros(1, V)}

        def forward(self, global_orient=None, body_pose=None, betas=None, pose2rot=False, **kw):
            assert not pose2rot
            verts, joints = O.smpl_forward(global_orient, body_pose, betas, self.c)
            return SMPLOutput(vertices=verts, joints=joints[:, :45].clone())

    lbs.vertices2joints = lambda J, v: torch.einsum("bik,ji->bjk", [v, J])
    lbs.batch_rodrigues = O.batch_rodrigues
    utils.SMPLOutput = SMPLOutput
    smplx.SMPLLayer, smplx.lbs, smplx.utils, smplx.__standin__ = SMPLLayer, lbs, utils, True
    sys.modules.update({"smplx": smplx, "smplx.lbs": lbs, "smplx.utils": utils})


def load_reference_wrapper():
    path = os.path.join(REF, "tokenhmr", "lib", "models", "smpl_wrapper.py")
    spec = importlib.util.spec_from_file_location("_ref_smpl_wrapper", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SMPL


def run(model_dir, j19_pkl, B=6, seed=0):
    """Reference wrapper over smplx on seeded inputs -> dict of arrays."""
    import smplx
    SMPL = load_reference_wrapper()
    R = torch.from_numpy(random_rotations(B * 24, seed=100 + seed).reshape(B, 24, 3, 3)).float()
    R[0] = torch.eye(3)                                                      # identity pose: vertices == shaped template
    betas = torch.from_numpy(np.random.default_rng(200 + seed).standard_normal((B, 10))).float()
    out = {"rotmat": R.numpy(), "betas": betas.numpy()}
    for hips in (False, True):
        m = SMPL(model_path=model_dir, gender="neutral", num_body_joints=23, joint_regressor_extra=j19_pkl, update_hips=hips)
        with torch.no_grad():
            o = m(global_orient=R[:, [0]], body_pose=R[:, 1:], betas=betas, pose2rot=False)   # tokenhmr.py:176
        tag = "hips" if hips else "plain"
        out[f"joints_{tag}"] = o.joints.numpy()
        if not hips:
            out["vertices"] = o.vertices.numpy()
    aa = torch.from_numpy(np.random.default_rng(300 + seed).standard_normal((64, 3))).float()
    aa[0] = 0.0                                                              # the epsilon branch of batch_rodrigues
    out["aa"] = aa.numpy()
    out["rodrigues"] = smplx.lbs.batch_rodrigues(aa).numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--smpl-pkl")
    ap.add_argument("--j19-pkl")
    ap.add_argument("--plumbing-check", action="store_true")
    a = ap.parse_args()
    if a.plumbing_check:
        install_standin_smplx()
    try:
        import smplx
    except ImportError:
        sys.exit("smplx is not importable here: the SMPL boundary stays UNPINNED (this script is the staged pin)")
    if getattr(smplx, "__standin__", False) != bool(a.plumbing_check) or not hasattr(smplx, "SMPLLayer"):
        sys.exit("`smplx` resolves to a stub, not the real package")
    smpl = make_synthetic_smpl(seed=0)
    with tempfile.TemporaryDirectory() as d:
        mp, jp = write_synthetic_pkls(smpl, d)
        # the loader the product uses must read the file smplx reads, to the same constants
        back = load_smpl_pkl(mp, jp)
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J19_regressor", "parents"):
            assert torch.equal(back[k].float(), smpl[k].float()), k
        out = run(os.path.dirname(mp), jp)
    out["smplx_version"] = np.array(getattr(smplx, "__version__", "standin" if a.plumbing_check else "unknown"))
    out["constants"] = np.array("tokenhmr_amd.smpl_assets.make_synthetic_smpl(seed=0)")
    if a.plumbing_check:
        from oracle import tokenhmr_oracle as O
        R, betas = torch.from_numpy(out["rotmat"]), torch.from_numpy(out["betas"])
        for hips, tag in ((False, "plain"), (True, "hips")):
            c = dict(smpl, update_hips=hips)
            v, j = O.smpl_forward(R[:, :1], R[:, 1:], betas, c)
            dj = np.abs(j.numpy() - out[f"joints_{tag}"]).max()
            print(f"reference smpl_wrapper.SMPL(update_hips={hips}) over the stand-in == oracle.smpl_forward: max|diff| joints = {dj:.1e}")
            assert dj == 0.0
        assert np.abs(v.numpy() - out["vertices"]).max() == 0.0
        print("plumbing OK (nothing written: lbs itself is NOT pinned by a stand-in)")
        return
    path = os.path.join(ROOT, "tests", "golden", "smplx_lbs.npz")
    out["vertices"] = out["vertices"][:, ::7]                                 # every 7th vertex: ~70 KB
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    if a.smpl_pkl and a.j19_pkl:
        real = run(os.path.dirname(a.smpl_pkl), a.j19_pkl, seed=1)
        real["vertices"] = real["vertices"][:, ::7]
        c = load_smpl_pkl(a.smpl_pkl, a.j19_pkl)
        real["constants_checksum"] = np.array([float(c[k].double().abs().sum()) for k in ("v_template", "shapedirs", "posedirs", "lbs_weights")])
        path = os.path.join(ROOT, "tests", "golden", "smplx_lbs_real.npz")
        np.savez_compressed(path, **real)
        print("wrote", path, "(inputs/outputs only; run the test with THMR_SMPL_PKL / THMR_J19_PKL set)")


if __name__ == "__main__":
    main()
