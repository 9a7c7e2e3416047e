"""TEST INFRASTRUCTURE ONLY — import the *real* reference modules by file path.

Works only where /root/reference exists (the build container, not the GPU box).
Used by oracle/gen_golden.py to run the reference's own nn.Modules on seeded
inputs and freeze the results under tests/golden/, and by the not-gpu test
that pins oracle/tokenhmr_oracle.py against the live reference when present.

The reference package cannot be imported normally (pytorch_lightning, yacs,
smplx, timm ... are absent), so each pure-torch module is loaded
with importlib under a fake parent package, with three tiny stubs:
  * timm.models.layers.{drop_path,to_2tuple,trunc_normal_}   (vit.py:10)
  * smplx.{SMPLHLayer,SMPLXLayer}                            (vanilla_pose_vqvae.py:10-17)
  * torch.Tensor.cuda -> identity on a GPU-less box          (quantize_cnn.py:18)
Nothing from the reference is copied; it is executed in place.
"""
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get("TOKENHMR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "tokenhmr", "lib", "models"))


def _install_stubs():
    if "timm.models.layers" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")

        def drop_path(x, drop_prob=0.0, training=False):
            assert not training, "reference stub: eval only"
            return x

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        def trunc_normal_(t, std=1.0, **kw):
            return torch.nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)

        layers.drop_path, layers.to_2tuple, layers.trunc_normal_ = drop_path, to_2tuple, trunc_normal_
        timm.models, timm_models.layers = timm_models, layers
        sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": layers})
    if "smplx" not in sys.modules:
        smplx = types.ModuleType("smplx")

        class _Layer(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

        smplx.SMPLHLayer = smplx.SMPLXLayer = smplx.SMPLLayer = _Layer
        sys.modules["smplx"] = smplx


def _load(modname, path, package=None):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load():
    """Return a namespace with the reference classes/functions used on the hot path."""
    if _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    _install_stubs()
    m = os.path.join(REF, "tokenhmr", "lib", "models")
    ns = types.SimpleNamespace()

    ns.vit = _load("_ref_vit", os.path.join(m, "backbones", "vit.py"))

    # components package (pose_transformer uses `from .t_cond_mlp import ...`)
    pkg = types.ModuleType("_ref_components")
    pkg.__path__ = [os.path.join(m, "components")]
    sys.modules["_ref_components"] = pkg
    _load("_ref_components.t_cond_mlp", os.path.join(m, "components", "t_cond_mlp.py"), "_ref_components")
    ns.pose_transformer = _load("_ref_components.pose_transformer",
                                os.path.join(m, "components", "pose_transformer.py"), "_ref_components")
    ns.modules = _load("_ref_head_modules", os.path.join(m, "heads", "modules.py"))
    ns.geometry = _load("_ref_geometry", os.path.join(REF, "tokenhmr", "lib", "utils", "geometry.py"))

    # tokenization.models package (relative imports + module-level SMPLH load + .cuda())
    t = os.path.join(REF, "tokenization", "models")
    tpkg = types.ModuleType("_ref_tokmodels")
    tpkg.__path__ = [t]
    sys.modules["_ref_tokmodels"] = tpkg
    orig_cuda = torch.Tensor.cuda
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        _load("_ref_tokmodels.resnet", os.path.join(t, "resnet.py"), "_ref_tokmodels")
        ns.quantize_cnn = _load("_ref_tokmodels.quantize_cnn", os.path.join(t, "quantize_cnn.py"), "_ref_tokmodels")
        ns.rotation_utils = _load("_ref_tokmodels.rotation_utils", os.path.join(t, "rotation_utils.py"), "_ref_tokmodels")
        ns.vqvae = _load("_ref_tokmodels.vanilla_pose_vqvae", os.path.join(t, "vanilla_pose_vqvae.py"), "_ref_tokmodels")
        ns._orig_cuda = orig_cuda
    finally:
        pass  # keep .cuda patched while reference quantizers are constructed (reset_codebook calls it)
    _cache["ns"] = ns
    return ns
