"""load_tokenhmr() against files in the reference's on-disk formats (-m gpu): a Lightning-style checkpoint with unrelated
keys, the tokenizer checkpoint with a pickled config object, the SMPL model pickle with a chumpy-typed array and a scipy
sparse J_regressor, SMPL_to_J19.pkl, smpl_mean_params.npz and model_config.yaml — read exactly where the reference reads
them (lib/models/__init__.py:3-26, lib/utils/misc.py:242-256, vanilla_pose_vqvae.py:299-301, smpl_wrapper.py:11-25).
The model loaded from disk must equal the model built from the same tensors in memory, bit for bit."""
import pickle
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _PickledCfg:                      # stands in for the yacs CfgNode the real tokenizer.pth carries
    def __init__(self):
        self.nb_code = 2048


def test_load_tokenhmr_from_reference_style_files(built_lib, cuda_dev, tmp_path):
    import scipy.sparse as sp
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.model import TokenHMR, load_tokenhmr

    cfg = HMRConfig(vit_depth=1, dec_depth=2)
    sd, tok, smpl = W.make_synthetic_state(cfg, 3), W.make_synthetic_tokenizer(cfg, 3), make_synthetic_smpl(cfg, 3)

    # --- Lightning checkpoint: ['state_dict'] with extra modules; init_cam left to smpl_mean_params.npz
    ck = {k: v for k, v in sd.items() if k != "smpl_head.init_cam"}
    ck.update({"discriminator.fc.weight": torch.zeros(4, 4), "smpl.faces_tensor": torch.zeros(8, 3, dtype=torch.int64)})
    torch.save({"state_dict": ck, "epoch": 7, "hyper_parameters": {"cfg": _PickledCfg()}}, tmp_path / "tokenhmr_model.ckpt")
    # --- tokenizer checkpoint: ['net'] + a pickled config object (the reason weights_only=False is needed)
    net = dict(tok)
    net["encoder.some_unused.weight"] = torch.zeros(3)
    torch.save({"net": net, "hparams": _PickledCfg()}, tmp_path / "tokenizer.pth")
    # --- SMPL_NEUTRAL.pkl in the official layout: chumpy array, scipy sparse regressor, uint32 kintree with 2^32-1 root
    fake = types.ModuleType("chumpy")
    fake_ch = types.ModuleType("chumpy.ch")

    class Ch:                                        # pickled by module path "chumpy.ch.Ch", like the official file
        def __init__(self, x):
            self.x = x

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    fake_ch.Ch = Ch
    fake.ch = fake_ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = fake, fake_ch
    try:
        kt = np.stack([np.array([2 ** 32 - 1] + [int(p) for p in smpl["parents"][1:]], dtype=np.uint32), np.arange(24, dtype=np.uint32)])
        d = {"v_template": smpl["v_template"].numpy(), "shapedirs": Ch(smpl["shapedirs"].numpy().astype(np.float64)),
             "posedirs": smpl["posedirs"].numpy().T.reshape(6890, 3, 207).astype(np.float64),
             "J_regressor": sp.csc_matrix(smpl["J_regressor"].numpy().astype(np.float64)), "weights": smpl["lbs_weights"].numpy(),
             "kintree_table": kt, "f": np.zeros((13776, 3), dtype=np.uint32)}
        (tmp_path / "smpl").mkdir()
        with open(tmp_path / "smpl" / "SMPL_NEUTRAL.pkl", "wb") as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    with open(tmp_path / "SMPL_to_J19.pkl", "wb") as f:
        pickle.dump(smpl["J19_regressor"].numpy(), f, protocol=2)
    np.savez(tmp_path / "smpl_mean_params.npz", pose=sd["smpl_head.init_body_pose"][0].numpy(), shape=sd["smpl_head.init_betas"][0].numpy(),
             cam=sd["smpl_head.init_cam"][0].numpy())
    (tmp_path / "model_config.yaml").write_text(f"""
MODEL:
  IMAGE_SIZE: 256
  IMAGE_MEAN: [0.485, 0.456, 0.406]
  IMAGE_STD: [0.229, 0.224, 0.225]
  TOKENIZER_CHECKPOINT_PATH: {tmp_path}/tokenizer.pth
  BACKBONE:
    TYPE: vit
  SMPL_HEAD:
    TYPE: token
    TRANSFORMER_DECODER:
      depth: 2
      heads: 8
SMPL:
  MODEL_PATH: {tmp_path}/smpl
  GENDER: neutral
  JOINT_REGRESSOR_EXTRA: {tmp_path}/SMPL_to_J19.pkl
  MEAN_PARAMS: {tmp_path}/smpl_mean_params.npz
EXTRA:
  FOCAL_LENGTH: 5000
DATASETS:
  DATASET_DIR: none
""")

    model, mcfg = load_tokenhmr(str(tmp_path / "tokenhmr_model.ckpt"), str(tmp_path / "model_config.yaml"), max_batch=2, device=cuda_dev)
    assert mcfg.MODEL.BBOX_SHAPE == [192, 256] and model.engine.cfg.vit_depth == 1 and model.engine.cfg.dec_depth == 2
    ref = TokenHMR.from_state(cfg, sd, tok, smpl, max_batch=2, device=cuda_dev)
    img = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(cuda_dev)
    a, b = model({"img": img}), ref({"img": img})
    for k in ("pred_vertices", "pred_keypoints_3d", "pred_keypoints_2d", "pred_cam", "cls_logits_softmax"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["pred_smpl_params"]["body_pose"], b["pred_smpl_params"]["body_pose"])

    # strict-load failures are loud (load_state_dict(strict=True) semantics, misc.py:246-250)
    bad = dict(ck)
    bad.pop("backbone.blocks.0.attn.qkv.bias")
    torch.save({"state_dict": bad}, tmp_path / "bad.ckpt")
    with pytest.raises((KeyError, ValueError, RuntimeError)):
        load_tokenhmr(str(tmp_path / "bad.ckpt"), str(tmp_path / "model_config.yaml"), max_batch=2, device=cuda_dev)
