"""Readers for the reference's on-disk checkpoints that need NONE of the packages those files pickle.

The released files are ordinary `torch.save` archives, but their pickles name classes of packages that are not part of
this product's environment (and are absent from the MI355X image):

  * tokenizer.pth          {'net': state_dict, 'hparams': yacs.config.CfgNode}
                           written by tokenization/utils/eval_poseVQ.py:118-125, read by
                           tokenization/models/vanilla_pose_vqvae.py:265-278 (DecodeTokens) and :309-321 (EncodeTokens)
  * tokenhmr_model.ckpt    a pytorch_lightning checkpoint: {'state_dict', 'hyper_parameters': {'cfg': CfgNode | omegaconf
                           DictConfig, ...}, 'callbacks', 'optimizer_states', ...} (tokenhmr.py:42 save_hyperparameters),
                           read by tokenhmr/lib/utils/misc.py:242-256

A plain `torch.load(weights_only=False)` of either raises ModuleNotFoundError wherever yacs / omegaconf /
pytorch_lightning are not importable.  `load_checkpoint` un-pickles them with a RESTRICTED `find_class`: tensors, storages,
numpy arrays and plain containers resolve to the real classes; every other global — whatever its module — becomes an inert,
dict-like stand-in (`InertNode`) that records what the pickle fed it and never imports or executes anything.  This is the
pattern `smpl_assets._Unpickler` uses for chumpy, extended to "any class", and it is also strictly safer than the
reference's unrestricted un-pickling: no `os.system`-style gadget can be resolved, a nested archive
(`torch.storage._load_from_bytes`, which is an unrestricted `torch.load` in torch itself) is read by THIS un-pickler again,
and of numpy only `ndarray`, `dtype`, the scalar types and the `numpy.dtypes` classes resolve (not `memmap`, whose
constructor creates / truncates files).

`tokenizer_arch` / `check_tokenizer_arch` then read `hparams.ARCH.*` from the stand-in exactly where the reference reads it
(vanilla_pose_vqvae.py:266-278) and compare it with the architecture the HIP engine bakes into its kernels.
"""
import io
import pickle
import types
import warnings

import numpy as np
import torch

# globals that are resolved for real: exactly what a tensor archive legitimately needs, by (module, name) — never by module alone
# (numpy.testing and torch.hub hold code-execution gadgets)
_REAL_GLOBALS = {
    "collections": {"OrderedDict", "defaultdict", "deque", "Counter"},
    "_codecs": {"encode"},
    "copyreg": {"_reconstructor", "__newobj__", "__newobj_ex__"},
    "torch._utils": {"_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_tensor_v3", "_rebuild_parameter", "_rebuild_parameter_with_state",
                     "_rebuild_qtensor", "_rebuild_sparse_tensor", "_rebuild_wrapper_subclass", "_rebuild_device_tensor_from_numpy",
                     "_rebuild_meta_tensor_no_storage"},
    "torch._tensor": {"_rebuild_from_type", "_rebuild_from_type_v2", "Tensor"},
    "torch.nn.parameter": {"Parameter", "Buffer", "UninitializedParameter"},
    "torch.storage": {"TypedStorage", "UntypedStorage"},       # _load_from_bytes: see _nested_archive below
    "torch.serialization": {"_get_layout"},
}
_NUMPY_MODULES = {"numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric",
                  "numpy.core._multiarray_umath", "numpy._core._multiarray_umath", "numpy.dtypes"}
_NUMPY_FUNCS = {"_reconstruct", "_frombuffer", "scalar"}
_REAL_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}


class InertNode(dict):
    """Stand-in for an object of a class that is not importable here (yacs CfgNode, omegaconf containers, Lightning
    enums / callbacks, argparse.Namespace, pathlib paths ...).  A dict subclass so the pickle opcodes that fill dict
    subclasses (SETITEM / SETITEMS: how a CfgNode's keys are stored) work; constructor arguments, `__setstate__` payloads and
    list-style APPENDS are kept as data.  Attribute access reads keys first, then the restored instance state, so
    `hparams.ARCH.CODE_DIM` works as it does on the real CfgNode."""

    _inert_origin = ("?", "?")

    def __init__(self, *args, **kwargs):
        dict.__init__(self)
        if len(args) == 1 and isinstance(args[0], dict) and not kwargs:
            dict.update(self, args[0])              # CfgNode(init_dict) / DictConfig(content)
        elif args or kwargs:
            self.__dict__["_inert_args"] = (args, kwargs)

    def __setstate__(self, state):
        slots = None
        if isinstance(state, tuple) and len(state) == 2 and (state[0] is None or isinstance(state[0], dict)):
            state, slots = state
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif state is not None:
            self.__dict__["_inert_state"] = state
        if isinstance(slots, dict):
            self.__dict__.update(slots)

    # list-like pickles (APPEND / APPENDS go through .append / .extend)
    def append(self, x):
        self.__dict__.setdefault("_inert_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_inert_items", []).extend(xs)

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if dict.__contains__(self, name):
            return dict.__getitem__(self, name)
        state = self.__dict__
        if name in state:
            return state[name]
        content = state.get("_content")             # omegaconf DictConfig keeps its mapping in _content
        if isinstance(content, dict) and name in content:
            return content[name]
        raise AttributeError(f"{type(self).__name__} (inert stand-in for {'.'.join(self._inert_origin)}) has no field '{name}'")

    def __call__(self, *args, **kwargs):            # an unknown FUNCTION used with REDUCE: keep its arguments, do nothing
        node = type(self)()
        node.__dict__["_inert_args"] = (args, kwargs)
        return node

    def __reduce__(self):
        raise pickle.PicklingError("InertNode stand-ins are read-only views of a foreign pickle")

    def __repr__(self):
        return f"<inert {'.'.join(self._inert_origin)} {dict.__repr__(self)}>"


def _nested_archive(b):
    """Replacement for torch.storage._load_from_bytes (= torch.load(io.BytesIO(b), weights_only=False) in torch: a nested,
    UNRESTRICTED un-pickle).  Tensors pickled outside torch.save reduce through it; the inner archive is read with the same
    restricted un-pickler, so nesting cannot be used to reach a real global."""
    if not isinstance(b, (bytes, bytearray)):
        raise pickle.UnpicklingError("_load_from_bytes: expected a bytes payload")
    return torch.load(io.BytesIO(bytes(b)), map_location="cpu", weights_only=False, pickle_module=_pickle_module)


def _numpy_data_type(obj):
    """ndarray, dtype, the scalar types (float64, int32 ...) and the numpy.dtypes classes — NOT ndarray subclasses such as memmap
    (its constructor opens / creates / truncates a file), recarray or matrix."""
    if not isinstance(obj, type):
        return False
    return obj is np.ndarray or obj is np.dtype or issubclass(obj, np.generic) or issubclass(obj, np.dtype)


_stub_cache = {}


def _stub_for(module, name):
    key = (module, name)
    if key not in _stub_cache:
        _stub_cache[key] = type(name.rsplit(".", 1)[-1] or "Inert", (InertNode,), {"_inert_origin": key})
    return _stub_cache[key]


class RestrictedUnpickler(pickle.Unpickler):
    """find_class that can only ever return tensor / array / container classes or an InertNode subclass."""

    def find_class(self, module, name):
        if module == "builtins" or module == "__builtin__":
            if name in _REAL_BUILTINS:
                return super().find_class("builtins", name)
            return _stub_for(module, name)
        if module == "torch":
            # storages, dtypes, torch.Size, torch.device — data classes only; functions of the top-level namespace are not needed
            obj = getattr(torch, name, None)
            if isinstance(obj, (type, torch.dtype)) or name.endswith("Storage"):
                return super().find_class(module, name)
            return _stub_for(module, name)
        if module == "torch.storage" and name == "_load_from_bytes":
            return _nested_archive
        if name in _REAL_GLOBALS.get(module, ()):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return _stub_for(module, name)
        if module in _NUMPY_MODULES:
            # array / scalar reconstruction helpers, and the DATA types (ndarray, dtype, float64, dtypes.Float32DType ...) — no other
            # numpy function and no other ndarray subclass
            try:
                obj = super().find_class(module, name)
            except (ImportError, AttributeError):
                return _stub_for(module, name)
            if name in _NUMPY_FUNCS or _numpy_data_type(obj):
                return obj
            return _stub_for(module, name)
        return _stub_for(module, name)


# torch.load(pickle_module=...) wants a module-like object with Unpickler / load / (unused here) dump
_pickle_module = types.SimpleNamespace(
    __name__="tokenhmr_amd.ckpt_io.restricted_pickle",
    Unpickler=RestrictedUnpickler,
    load=lambda f, **kw: RestrictedUnpickler(f, **kw).load(),
    loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps, HIGHEST_PROTOCOL=pickle.HIGHEST_PROTOCOL,
    PicklingError=pickle.PicklingError, UnpicklingError=pickle.UnpicklingError, Pickler=pickle.Pickler,
)


def load_checkpoint(path):
    """torch.load(path, map_location='cpu') for the reference's checkpoints, with the restricted un-pickler above.
    Works for the zip format and the legacy format; needs neither yacs, omegaconf nor pytorch_lightning."""
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_pickle_module)


# ---------------------------------------------------------------------------------------------------------------------
# tokenizer architecture: what vanilla_pose_vqvae.py:266-278 reads from ckpt['hparams'].ARCH, and what the engine bakes in
RELEASE_ARCH = {            # tokenization/configs/tokenizer_amass_moyo.yaml:41-54 == the constants of csrc/engine.hip
    "ROT_TYPE": "rot6d", "CODE_DIM": 256, "NB_CODE": 2048, "DOWN_T": 1, "WIDTH": 512, "DEPTH": 2,
    "DILATION_RATE": 3, "TOKEN_SIZE_DIV": 4, "TOKEN_SIZE_MUL": 4,
}


def _scalar(v):
    """tokenizer_amass_moyo.yaml writes CODE_DIM / NB_CODE as one-element lists (grid-search syntax): unwrap them."""
    if isinstance(v, (list, tuple)) and len(v) == 1:
        v = v[0]
    if isinstance(v, torch.Tensor) and v.numel() == 1:
        v = v.item()
    if hasattr(v, "item") and not isinstance(v, (str, bytes)):
        try:
            v = v.item()
        except Exception:
            pass
    return v


def _field(node, key):
    if isinstance(node, dict) and key in node:
        return node[key]
    try:
        return getattr(node, key)
    except AttributeError:
        raise KeyError(key) from None


def tokenizer_arch(ckpt):
    """{'ROT_TYPE', 'CODE_DIM', ...} from a loaded tokenizer checkpoint (ckpt['hparams'].ARCH.*), or None if the file carries
    no hparams (a bare {'net': ...} as this repo's synthetic fixtures write)."""
    if not isinstance(ckpt, dict) or "hparams" not in ckpt or ckpt["hparams"] is None:
        return None
    try:
        arch = _field(ckpt["hparams"], "ARCH")
    except KeyError:
        raise KeyError("tokenizer checkpoint: ckpt['hparams'] has no ARCH node (vanilla_pose_vqvae.py:266-267)") from None
    out = {}
    for k in RELEASE_ARCH:
        try:
            out[k] = _scalar(_field(arch, k))
        except KeyError:
            raise KeyError(f"tokenizer checkpoint: hparams.ARCH.{k} is missing (vanilla_pose_vqvae.py:268-278 reads it)") from None
    return out


def check_tokenizer_arch(arch, hmr_cfg=None):
    """The HIP engine's VQ decoder / encoder kernels are built for the release ARCH; a tokenizer trained with another one
    would load tensor-for-tensor only by accident.  Raise ValueError naming the first differing key."""
    if arch is None:
        return
    want = dict(RELEASE_ARCH)
    if hmr_cfg is not None:
        want.update(CODE_DIM=hmr_cfg.code_dim, NB_CODE=hmr_cfg.token_classes, WIDTH=hmr_cfg.vq_width,
                    DILATION_RATE=hmr_cfg.vq_dilation)
    for k, w in want.items():
        g = arch.get(k)
        same = (str(g) == str(w)) if isinstance(w, str) else (isinstance(g, (int, float)) and not isinstance(g, bool) and float(g) == float(w))
        if not same:
            raise ValueError(f"tokenizer checkpoint architecture differs from the engine's: hparams.ARCH.{k} = {g!r}, "
                             f"the HIP kernels are built for {w!r} (tokenization/configs/tokenizer_amass_moyo.yaml:41-54)")
    # num_tokens as DecodeTokens derives it (vanilla_pose_vqvae.py:279): ((21 // 10) * 10) * 2**TOKEN_SIZE_MUL / 2**DOWN_T
    n_tok = ((21 // 10) * 10) * (2 ** int(arch["TOKEN_SIZE_MUL"])) / (2 ** int(arch["DOWN_T"]))
    want_tok = hmr_cfg.token_num if hmr_cfg is not None else 160
    if n_tok != want_tok:
        raise ValueError(f"tokenizer checkpoint yields {n_tok} tokens per pose, the engine is built for {want_tok}")


def select_state(full_state, prefixes, known, strict=True, what="checkpoint"):
    """The reference's prepare_statedict (misc.py:215-240) keeps the keys of one sub-module and load_state_dict()s them.  Its
    strict failure is caught and only logged (`:228-233`), after torch has already copied every matching tensor — i.e. the
    reference is lenient about UNEXPECTED keys in practice.  Here: strict=True (default) raises KeyError on a key under
    `prefixes` that the engine has no slot for; strict=False warns and skips it, like the reference.  Missing tensors are
    reported by weights.validate_state either way (the reference would silently keep a random initialisation)."""
    out, unknown = {}, []
    for k, v in full_state.items():
        if not k.startswith(prefixes):
            continue
        if k in known:
            out[k] = v
        else:
            unknown.append(k)
    if unknown:
        msg = f"{what}: {len(unknown)} tensor(s) the inference engine has no slot for: {unknown[:6]}{' ...' if len(unknown) > 6 else ''}"
        if strict:
            raise KeyError(msg + " (pass strict=False to skip them with a warning, as the reference's prepare_statedict does)")
        warnings.warn("Mismatch in statedict! " + msg + " — skipped", RuntimeWarning, stacklevel=3)
    return out
