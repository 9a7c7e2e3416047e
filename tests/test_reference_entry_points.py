"""The reference's own entry points EXECUTED against the facade (not-gpu; runs where /root/reference exists).

tests/test_reference_call_sites.py proves by static analysis that everything eval.py / demo.py do with the model is served
by tokenhmr_amd.model; here the scripts themselves run, in place, to completion:

  * tokenhmr/eval.py  `run_eval` (:116-158)  — DataLoader -> recursive_to -> model(batch) -> Evaluator -> save_eval_result CSV
  * tokenhmr/demo.py  `main` (:17-140)       — load_tokenhmr -> .to(device) -> .eval() -> model.smpl.faces -> ViTDetDataset ->
                                                DataLoader(batch_size=8) -> model(batch) -> cam_crop_to_full -> renderer calls

with `load_tokenhmr` / `Evaluator` swapped exactly as INTEGRATION.md §1 says (`from tokenhmr_amd.model import load_tokenhmr`,
`from tokenhmr_amd.evaluator import Evaluator`).  There is no GPU here and the product has no CPU path, so the model is the
real facade class (`TokenHMR.from_engine`) over a stand-in engine whose arithmetic is the CPU oracle at depth 1, and the
evaluator's two kernel calls are served by oracle/eval_oracle.py — both are test infrastructure.  What is under test is the
CONTRACT: attribute surface, dict keys, shapes, dtypes, collated batch types, call order — any mismatch makes the reference's
own code raise.  Third-party packages those scripts import and this image lacks (cv2, smplx, detectron2, pyrender ...) are
stubbed for the duration of a test and removed afterwards.  The metric values written by the reference's run_eval are compared
with the reference's OWN Evaluator on the same outputs."""
import ast
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference/tokenhmr"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box): its entry points run in the build container")


@contextlib.contextmanager
def _modules(stubs):
    """Install stub modules, restore sys.modules exactly on exit (other tests probe for a REAL cv2 / smplx)."""
    saved = {k: sys.modules.get(k) for k in stubs}
    before = set(sys.modules)
    sys.modules.update(stubs)
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in set(sys.modules) - before:
            if k.startswith(("_ref_entry", "lib.", "detectron2")) or k in ("lib",):
                sys.modules.pop(k, None)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _exec_reference_script(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _reference_function(rel, fname, glb):
    """Compile ONE function of a reference file in place (its module cannot be imported: pyrender / trimesh at the top)."""
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fname)
    code = compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, rel), "exec")
    ns = dict(glb)
    exec(code, ns)
    return ns[fname]


class _Ns(dict):
    """attribute + mapping access, like the yacs node the reference scripts index both ways (dict(model_cfg.SMPL), cfg.MODEL.get)"""
    __getattr__ = dict.__getitem__


# ------------------------------------------------------------------------------------------------ stand-in engine
class OracleEngine:
    """Engine surface (tokenhmr_amd/engine.py) over the CPU oracle — TEST ONLY; lets the real facade class run without a GPU."""

    def __init__(self, cfg, sd, tok, smpl, max_batch=8):
        self.cfg, self.sd, self.tok, self.smpl_c = cfg, sd, tok, smpl
        self.max_batch, self.device = max_batch, torch.device("cpu")
        self.calls = []

    def forward(self, img, taps=False, want_probs=True, outputs=None):
        from oracle import tokenhmr_oracle as O
        assert img.dtype == torch.float32 and tuple(img.shape[1:]) == (3, 256, 256) and img.shape[0] <= self.max_batch
        self.calls.append(int(img.shape[0]))
        with torch.no_grad():
            r = O.forward(img, self.sd, self.tok, self.smpl_c, self.cfg)
        R = torch.cat([r["pred_smpl_params"]["global_orient"], r["pred_smpl_params"]["body_pose"]], 1)
        return {"pred_cam": r["pred_cam"], "rotmat": R, "betas": r["pred_smpl_params"]["betas"], "cls_logits_softmax": r["cls_logits_softmax"],
                "pred_cam_t": r["pred_cam_t"], "focal_length": r["focal_length"], "pred_keypoints_3d": r["pred_keypoints_3d"],
                "pred_vertices": r["pred_vertices"], "pred_keypoints_2d": r["pred_keypoints_2d"], "token_idx": r["token_idx"].to(torch.int32)}


@pytest.fixture(scope="module")
def facade():
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.model import TokenHMR
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    sd, tok, smpl = W.make_synthetic_state(cfg, 5), W.make_synthetic_tokenizer(cfg, 5), make_synthetic_smpl(cfg, 5)
    eng = OracleEngine(cfg, sd, tok, smpl)
    model_cfg = _Ns(MODEL=_Ns(IMAGE_SIZE=256, IMAGE_MEAN=[0.485, 0.456, 0.406], IMAGE_STD=[0.229, 0.224, 0.225], BBOX_SHAPE=[192, 256],
                              BACKBONE=_Ns(TYPE="vit")),
                    EXTRA=_Ns(FOCAL_LENGTH=5000, PELVIS_IND=39), SMPL=_Ns(MODEL_PATH="data/body_models/smpl", GENDER="neutral"))
    model = TokenHMR.from_engine(eng, faces=smpl["faces"], model_cfg=model_cfg)
    return model, model_cfg, eng


def _cpu_evaluator_kernels(monkeypatch):
    """tokenhmr_amd.evaluator's two library calls served by the CPU oracle (oracle/eval_oracle.py, pinned to the reference)."""
    from oracle import eval_oracle as EO
    import tokenhmr_amd.evaluator as E

    def eval_pose_cpu(pred_joints, gt_joints, keypoint_list, pelvis_ind, pelvis_mode=0, pred_vertices=None, gt_vertices=None):
        pk, gk = pred_joints.detach().float().clone(), gt_joints.detach().float()[:, :, :3].clone()
        if pelvis_mode == 0:
            pp, gp = pk[:, [pelvis_ind]], gk[:, [pelvis_ind]]
        else:
            pp, gp = (pk[:, [1]] + pk[:, [2]]) / 2.0, (gk[:, [1]] + gk[:, [2]]) / 2.0
        mp, re = EO.eval_pose((pk - pp)[:, list(keypoint_list)], (gk - gp)[:, list(keypoint_list)])
        pve = None
        if pred_vertices is not None:
            pve = torch.sqrt((((pred_vertices - pp) - (gt_vertices - gp)) ** 2).sum(-1)).mean(-1) * 1000.0
        return mp, re, pve

    monkeypatch.setattr(E, "eval_pose_gpu", eval_pose_cpu)
    monkeypatch.setattr(E, "regress_joints_gpu", lambda J, v: torch.matmul(J.float(), v.float()))
    return E.Evaluator


class _EvalSet(torch.utils.data.Dataset):
    """Items with the keys the reference's eval datasets hand to the Evaluator (image_dataset.py / emdb_dataset.py)."""

    def __init__(self, n, smpl, seed=0):
        g = torch.Generator().manual_seed(8000 + seed)
        self.img = torch.randn(n, 3, 256, 256, generator=g)
        self.k3d = torch.cat([0.3 * torch.randn(n, 44, 3, generator=g), torch.ones(n, 44, 1)], -1)
        self.k2d = torch.cat([torch.randn(n, 44, 2, generator=g), torch.ones(n, 44, 1)], -1)
        self.verts = 0.3 * torch.randn(n, 6890, 3, generator=g)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return {"img": self.img[i], "keypoints_3d": self.k3d[i], "keypoints_2d": self.k2d[i], "vertices": self.verts[i],
                "imgname": f"seq/img_{i:05d}.jpg", "personid": i}


@pytest.mark.parametrize("dataset_name", ["3DPW-TEST", "EMDB"])
def test_reference_run_eval_executes_against_the_facade(facade, monkeypatch, tmp_path, dataset_name, capsys):
    import pandas as pd
    from oracle.gen_golden_eval import load_pose_utils
    model, model_cfg, eng = facade
    Evaluator = _cpu_evaluator_kernels(monkeypatch)
    from tokenhmr_amd.model import load_tokenhmr
    recursive_to = _reference_function("lib/utils/__init__.py", "recursive_to", {"torch": torch, "Any": object})
    J24 = torch.softmax(4 * torch.randn(24, 6890, generator=torch.Generator().manual_seed(3)), dim=1)
    data = _EvalSet(5, None)

    class _Reg:                                     # smplx.SMPL(model_path=...).J_regressor.cuda().float()  (eval.py:130)
        def __init__(self, t):
            self.t = t

        def cuda(self):
            return self

        def float(self):
            return self.t

    stubs = {
        "cv2": _mod("cv2"),
        "smplx": _mod("smplx", SMPL=lambda model_path=None, **k: types.SimpleNamespace(J_regressor=_Reg(J24))),
        "lib": _mod("lib"),
        "lib.utils": _mod("lib.utils", Evaluator=Evaluator, recursive_to=recursive_to, MeshRenderer=object),
        "lib.configs": _mod("lib.configs", dataset_eval_config=lambda: {}),
        "lib.datasets": _mod("lib.datasets", create_dataset=lambda model_cfg, dataset_cfg, train=False: data),
        "lib.models": _mod("lib.models", load_tokenhmr=load_tokenhmr),
        "lib.models.smpl_wrapper": _mod("lib.models.smpl_wrapper", SMPL=object),
    }
    kp = [25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 43] if dataset_name == "3DPW-TEST" else list(range(24))
    args = types.SimpleNamespace(render=False, batch_size=2, shuffle=False, num_workers=0, dataset=dataset_name, log_freq=2,
                                 results_file=str(tmp_path / "eval_regression.csv"), checkpoint="logs/run1/checkpoints/last.ckpt", exp_name=None)
    eng.calls.clear()
    with _modules(stubs):
        ref_eval = _exec_reference_script("_ref_entry_eval", "eval.py")
        ref_eval.run_eval(model, model_cfg, _Ns(KEYPOINT_LIST=kp), torch.device("cpu"), args)
    assert eng.calls == [2, 2, 1]                                      # 5 samples at --batch_size 2, every batch went through the facade
    df = pd.read_csv(args.results_file)
    assert list(df["metric_name"]) == ["mode_re", "mode_mpjpe", "mode_pve"] and list(df["dataset"]) == [dataset_name] * 3
    assert list(df["exp_name"]) == ["run1"] * 3 and list(df["iters_done"]) == [2] * 3
    assert "5 / 100000000 samples" in capsys.readouterr().out          # Evaluator.log() of the swapped class, called by the reference
    # the same numbers from the reference's OWN Evaluator over the same model outputs
    with _modules({"cv2": _mod("cv2")}):
        pu = load_pose_utils()
        ev = pu.Evaluator(dataset_length=100, keypoint_list=kp, pelvis_ind=39, metrics=["mode_re", "mode_mpjpe", "mode_pve"],
                          J_regressor_24_SMPL=J24, dataset=dataset_name)
        for batch in torch.utils.data.DataLoader(data, 2, shuffle=False, num_workers=0):
            with torch.no_grad():
                ev(model(batch), batch)
        want = ev.get_metrics_dict()
        sys.modules.pop("_ref_utils", None), sys.modules.pop("_ref_utils.pose_utils", None), sys.modules.pop("_ref_utils.rotation_utils", None)
    for name, val in zip(df["metric_name"], df["metric_value"]):
        assert abs(val - float("{:.2f}".format(want[name]))) <= 0.011, (name, val, want[name])


def test_reference_demo_main_executes_against_the_facade(facade, monkeypatch, tmp_path):
    """demo.py main(): every line of the per-frame loop (:62-121, side view and mesh export included) runs against the facade; the
    detector, the renderer and cv2's file I/O are stubs that record what they were handed."""
    from oracle.gen_golden_crop import load_reference_datasets, synthetic_frame, BOXES
    import tokenhmr_amd.model as M
    model, model_cfg, eng = facade
    frame = synthetic_frame()
    boxes = BOXES[[0, 1, 3]]
    seen = {"imwrite": [], "render": [], "render_multi": 0, "export": [], "load": None, "faces": None}

    def fake_load_tokenhmr(checkpoint_path="", model_cfg="", **kw):
        assert set(kw) <= set(__import__("inspect").signature(M.load_tokenhmr).parameters)      # is_train_state / is_demo are accepted
        seen["load"] = (checkpoint_path, model_cfg, kw)
        return model, facade[1]

    class Renderer:                                          # lib/utils/renderer.py Renderer: called as in demo.py:52,94-109,127
        def __init__(self, cfg, faces):
            seen["faces"] = np.asarray(faces)

        def __call__(self, vertices, camera_translation, image, **kw):
            assert vertices.shape == (6890, 3) and vertices.dtype == np.float32 and camera_translation.shape == (3,)
            assert torch.is_tensor(image) and tuple(image.shape) == (3, 256, 256)
            seen["render"].append(kw.get("side_view", False))
            return np.zeros((256, 256, 3), dtype=np.float32)

        def vertices_to_trimesh(self, verts, cam_t, color):
            assert verts.shape == (6890, 3) and cam_t.shape == (3,)
            return types.SimpleNamespace(export=lambda p: seen["export"].append(os.path.basename(p)))

        def render_rgba_multiple(self, all_verts, cam_t=None, render_res=None, **kw):
            assert len(all_verts) == len(cam_t) == 3 and "focal_length" in kw
            seen["render_multi"] += 1
            return np.zeros((frame.shape[0], frame.shape[1], 4), dtype=np.float32)

    class Predictor:                                         # DefaultPredictor_Lazy(cfg)(img) -> {'instances': ...}
        def __init__(self, cfg):
            pass

        def __call__(self, img):
            inst = types.SimpleNamespace(pred_classes=torch.zeros(3, dtype=torch.int64), scores=torch.tensor([0.9, 0.8, 0.7]),
                                         pred_boxes=types.SimpleNamespace(tensor=torch.from_numpy(boxes.astype(np.float32))))
            return {"instances": inst}

    lazy_cfg = types.SimpleNamespace(train=types.SimpleNamespace(init_checkpoint=None),
                                     model=types.SimpleNamespace(roi_heads=types.SimpleNamespace(box_predictors=[types.SimpleNamespace(test_score_thresh=0) for _ in range(3)])))
    cam_crop_to_full = _reference_function("lib/utils/renderer.py", "cam_crop_to_full", {"torch": torch})
    recursive_to = _reference_function("lib/utils/__init__.py", "recursive_to", {"torch": torch, "Any": object})
    saved = {k: sys.modules.get(k) for k in ("cv2", "skimage", "skimage.filters", "skimage.transform", "yacs", "yacs.config", "webdataset", "braceexpand")}
    ds = load_reference_datasets()                           # the reference's OWN ViTDetDataset, cv2 primitives from oracle/crop_oracle.py
    cv2 = sys.modules["cv2"]
    cv2.imread = lambda p: frame.copy()
    cv2.imwrite = lambda p, im: seen["imwrite"].append((os.path.basename(p), tuple(np.asarray(im).shape))) or True
    (tmp_path / "imgs").mkdir()
    (tmp_path / "imgs" / "frame_000.jpg").write_bytes(b"")
    lib = _mod("lib")
    lib.__file__ = os.path.join(REF, "lib", "__init__.py")
    stubs = {
        "cv2": cv2, "lib": lib,
        "lib.models": _mod("lib.models", load_tokenhmr=fake_load_tokenhmr),
        "lib.utils": _mod("lib.utils", recursive_to=recursive_to),
        "lib.utils.renderer": _mod("lib.utils.renderer", Renderer=Renderer, cam_crop_to_full=cam_crop_to_full),
        "lib.utils.utils_detectron2": _mod("lib.utils.utils_detectron2", DefaultPredictor_Lazy=Predictor),
        "lib.datasets": _mod("lib.datasets"),
        "lib.datasets.vitdet_dataset": ds["vitdet_dataset"],
        "detectron2": _mod("detectron2"), "detectron2.engine": _mod("detectron2.engine"),
        "detectron2.engine.defaults": _mod("detectron2.engine.defaults", DefaultPredictor=Predictor),
        "detectron2.config": _mod("detectron2.config", LazyConfig=types.SimpleNamespace(load=lambda p: lazy_cfg)),
    }
    monkeypatch.setattr(sys, "argv", ["demo.py", "--checkpoint", "ckpt/tokenhmr_model.ckpt", "--model_config", "ckpt/model_config.yaml",
                                      "--img_folder", str(tmp_path / "imgs"), "--out_folder", str(tmp_path / "out"),
                                      "--side_view", "--full_frame", "--save_mesh"])
    eng.calls.clear()
    try:
        with _modules(stubs):
            demo = _exec_reference_script("_ref_entry_demo", "demo.py")
            demo.main()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in [k for k in sys.modules if k.startswith("_ref_ds")]:
            sys.modules.pop(k, None)
    assert seen["load"] == ("ckpt/tokenhmr_model.ckpt", "ckpt/model_config.yaml", {"is_train_state": False, "is_demo": True})
    assert seen["faces"].shape == (13776, 3)                                   # model.smpl.faces (demo.py:52)
    assert eng.calls == [3]                                                    # the 3 detected persons in ONE batch of <= 8 (demo.py:70)
    assert seen["render"] == [False, True] * 3 and seen["render_multi"] == 1
    assert [n for n, _ in seen["imwrite"]] == ["frame_000_0.png", "frame_000_1.png", "frame_000_2.png", "frame_000_all.png"]
    assert all(s == (256, 768, 3) for _, s in seen["imwrite"][:3]) and seen["imwrite"][3][1] == frame.shape
    assert seen["export"] == ["frame_000_0.obj", "frame_000_1.obj", "frame_000_2.obj"]
    assert lazy_cfg.model.roi_heads.box_predictors[0].test_score_thresh == 0.25
