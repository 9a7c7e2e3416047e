"""Writers of files in the REFERENCE's on-disk formats, for the loader tests (CPU and GPU).

The released tokenizer.pth pickles a `yacs.config.CfgNode` (tokenization/utils/eval_poseVQ.py:118-125) and the Lightning
checkpoint's 'hyper_parameters' hold config nodes too (tokenhmr.py:42).  None of yacs / omegaconf / pytorch_lightning is
installed in this image, so the writers below register look-alike classes under exactly those module paths while the file is
written and REMOVE them before anything is read back — the same thing the chumpy part of the SMPL pickle test does.  What
ends up in the pickle stream is then byte-for-byte what the real packages would put there for these objects: the GLOBAL
opcode with the real module path, the dict-subclass SETITEMS for a CfgNode's keys and the BUILD state of its instance dict."""
import contextlib
import pickle
import sys
import types

import numpy as np
import torch

FOREIGN = ("yacs", "yacs.config", "omegaconf", "omegaconf.dictconfig", "omegaconf.base", "pytorch_lightning",
           "pytorch_lightning.utilities", "pytorch_lightning.utilities.enums", "pytorch_lightning.callbacks",
           "pytorch_lightning.callbacks.model_checkpoint", "chumpy", "chumpy.ch")


@contextlib.contextmanager
def foreign_modules():
    """Install look-alikes of the classes the reference's checkpoints pickle; remove them (and anything imported under those
    names) on exit, so that reading the files back happens in an interpreter where `import yacs` fails."""
    assert not any(m in sys.modules for m in FOREIGN), "a real yacs/omegaconf/pytorch_lightning is installed: test premise gone"
    mods = {name: types.ModuleType(name) for name in FOREIGN}

    class CfgNode(dict):                       # yacs.config.CfgNode: a dict subclass with attribute access + instance state
        IMMUTABLE, DEPRECATED_KEYS, RENAMED_KEYS, NEW_ALLOWED = "__immutable__", "__deprecated_keys__", "__renamed_keys__", "__new_allowed__"

        def __init__(self, init_dict=None, key_list=None, new_allowed=False):
            super().__init__({} if init_dict is None else {k: (CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)
                                                           for k, v in init_dict.items()})
            self.__dict__[CfgNode.IMMUTABLE] = False
            self.__dict__[CfgNode.DEPRECATED_KEYS] = set()
            self.__dict__[CfgNode.RENAMED_KEYS] = {}
            self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed

        def __getattr__(self, name):
            if name in self:
                return self[name]
            raise AttributeError(name)

    class DictConfig:                          # omegaconf.dictconfig.DictConfig: plain object, state dict with _content/_metadata
        def __init__(self, content):
            self.__dict__["_content"] = content
            self.__dict__["_metadata"] = Metadata()
            self.__dict__["_parent"] = None

    class Metadata:                            # omegaconf.base.ContainerMetadata
        def __init__(self):
            self.ref_type, self.object_type, self.optional, self.key = dict, dict, True, None

    import enum

    class LightningEnum(str, enum.Enum):       # pytorch_lightning.utilities.enums.*: pickled by value through REDUCE
        FITTING = "fit"

    class ModelCheckpoint:                     # a callback object some Lightning versions pickle under 'callbacks'
        def __init__(self):
            self.monitor, self.best_model_score = "val/loss", torch.tensor(1.5)

    class Ch:                                  # chumpy.ch.Ch of the official SMPL pickle
        def __init__(self, x):
            self.x = x

    for cls, mod in ((CfgNode, "yacs.config"), (DictConfig, "omegaconf.dictconfig"), (Metadata, "omegaconf.base"),
                     (LightningEnum, "pytorch_lightning.utilities.enums"), (ModelCheckpoint, "pytorch_lightning.callbacks.model_checkpoint"),
                     (Ch, "chumpy.ch")):
        cls.__module__, cls.__qualname__ = mod, cls.__name__
        setattr(mods[mod], cls.__name__, cls)
    sys.modules.update(mods)
    try:
        yield types.SimpleNamespace(CfgNode=CfgNode, DictConfig=DictConfig, LightningEnum=LightningEnum,
                                    ModelCheckpoint=ModelCheckpoint, Ch=Ch)
    finally:
        for name in FOREIGN:
            sys.modules.pop(name, None)


RELEASE_ARCH_YAML = {          # tokenization/configs/tokenizer_amass_moyo.yaml:41-54, lists included
    "MODEL_NAME": "vanilla", "CODE_DIM": [256], "NB_CODE": [2048], "ROT_TYPE": "rot6d", "QUANTIZER": "ema_reset",
    "SMPL_TYPE": "smplh", "DOWN_T": 1, "WIDTH": 512, "DEPTH": 2, "DILATION_RATE": 3, "CB_SCALE_DOWN": 2,
    "TOKEN_SIZE_MUL": 4, "TOKEN_SIZE_DIV": 4,
}


def write_reference_files(tmp_path, cfg, sd, tok, smpl, arch_overrides=None, extra_state=None, legacy_format=False, dec_depth=None):
    """tokenhmr_model.ckpt (Lightning layout), tokenizer.pth (yacs hparams), SMPL_NEUTRAL.pkl (chumpy + scipy sparse),
    SMPL_to_J19.pkl, smpl_mean_params.npz, model_config.yaml under tmp_path.  Returns the two paths load_tokenhmr takes."""
    import scipy.sparse as sp
    arch = dict(RELEASE_ARCH_YAML)
    arch.update(arch_overrides or {})
    with foreign_modules() as F:
        # --- Lightning checkpoint: ['state_dict'] with the other sub-modules' tensors, hyper_parameters with config nodes
        ck = {k: v for k, v in sd.items() if k != "smpl_head.init_cam"}              # init_cam left to smpl_mean_params.npz
        ck.update({"discriminator.fc.weight": torch.zeros(4, 4), "smpl.faces_tensor": torch.zeros(8, 3, dtype=torch.int64)})
        ck.update(extra_state or {})
        hyper = {"cfg": F.CfgNode({"MODEL": {"IMAGE_SIZE": 256, "BACKBONE": {"TYPE": "vit"}}, "TRAIN": {"LR": 1e-5}}),
                 "hydra_cfg": F.DictConfig({"trainer": {"devices": 8}}), "init_renderer": False}
        light = {"epoch": 7, "global_step": 12345, "pytorch-lightning_version": "2.0.2", "state_dict": ck,
                 "hyper_parameters": hyper, "callbacks": {"ModelCheckpoint{'monitor': 'val/loss'}": {"best": F.ModelCheckpoint()}},
                 "loops": {"state": F.LightningEnum.FITTING}, "optimizer_states": [{"state": {0: {"exp_avg": torch.zeros(3)}}}],
                 "np_scalar": np.float64(2.5), "np_array": np.arange(4)}
        torch.save(light, tmp_path / "tokenhmr_model.ckpt", _use_new_zipfile_serialization=not legacy_format)
        # --- tokenizer checkpoint exactly as eval_poseVQ.py:118-125 writes it: {'net', 'hparams': CfgNode}
        net = dict(tok)
        net["body_model.shapedirs"] = torch.zeros(3)
        hparams = F.CfgNode({"DATA": {"BATCH_SIZE": 256}, "ARCH": arch, "EXP_NAME": "release"})
        torch.save({"net": net, "hparams": hparams}, tmp_path / "tokenizer.pth", _use_new_zipfile_serialization=not legacy_format)
        # --- SMPL_NEUTRAL.pkl in the official layout: chumpy array, scipy sparse regressor, uint32 kintree with 2^32-1 root
        kt = np.stack([np.array([2 ** 32 - 1] + [int(p) for p in smpl["parents"][1:]], dtype=np.uint32), np.arange(24, dtype=np.uint32)])
        d = {"v_template": smpl["v_template"].numpy(), "shapedirs": F.Ch(smpl["shapedirs"].numpy().astype(np.float64)),
             "posedirs": smpl["posedirs"].numpy().T.reshape(6890, 3, 207).astype(np.float64),
             "J_regressor": sp.csc_matrix(smpl["J_regressor"].numpy().astype(np.float64)), "weights": smpl["lbs_weights"].numpy(),
             "kintree_table": kt, "f": np.zeros((13776, 3), dtype=np.uint32)}
        (tmp_path / "smpl").mkdir(exist_ok=True)
        with open(tmp_path / "smpl" / "SMPL_NEUTRAL.pkl", "wb") as f:
            pickle.dump(d, f, protocol=2)
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
      depth: {cfg.dec_depth if dec_depth is None else dec_depth}
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
    return str(tmp_path / "tokenhmr_model.ckpt"), str(tmp_path / "model_config.yaml")
