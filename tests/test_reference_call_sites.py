"""Drop-in conformance by static analysis (runs where /root/reference exists, i.e. in the build container; no GPU needed).

The reference's CLIs can be executed against the facade nowhere: the reference tree exists only here (no GPU), the GPU box has no
reference tree, and the product has no CPU path.  What CAN be checked here is that everything those scripts DO with the model is
served by the facade: every keyword they pass to load_tokenhmr, every attribute they touch on the returned model, every key they
read from the output dict — collected from the reference's own sources with `ast`, compared with what tokenhmr_amd.model
declares (its signature, its class, the dict literal of TokenHMR._pack).  The loop bodies themselves run on the GPU box in
tests/test_gpu_pipeline.py."""
import ast
import inspect
import os

import pytest

REF = "/root/reference/tokenhmr"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box): static call-site check runs in the build container")


def _parse(rel):
    with open(os.path.join(REF, rel)) as f:
        return ast.parse(f.read())


def _subscript_keys(tree, names):
    """string keys k of `name[k]` loads for name in names"""
    keys = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Subscript) and isinstance(n.value, ast.Name) and n.value.id in names:
            s = n.slice
            if isinstance(s, ast.Constant) and isinstance(s.value, str):
                keys.add(s.value)
    return keys


def _attrs(tree, name):
    return {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == name}


def _facade_output_keys():
    from tokenhmr_amd import model as M
    tree = ast.parse(inspect.getsource(M))
    pack = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_pack")
    keys, nested = set(), {}
    for n in ast.walk(pack):
        if isinstance(n, ast.Dict):
            for k, v in zip(n.keys, n.values):
                if isinstance(k, ast.Constant):
                    keys.add(k.value)
                    if isinstance(v, ast.Dict):
                        nested[k.value] = {kk.value for kk in v.keys if isinstance(kk, ast.Constant)}
    return keys, nested


def test_load_tokenhmr_accepts_every_keyword_the_reference_passes():
    from tokenhmr_amd.model import load_tokenhmr
    params = set(inspect.signature(load_tokenhmr).parameters)
    ref_sig = None
    for n in ast.walk(_parse("lib/models/__init__.py")):
        if isinstance(n, ast.FunctionDef) and n.name == "load_tokenhmr":
            ref_sig = [a.arg for a in n.args.args]
    assert ref_sig, "reference load_tokenhmr not found"
    assert set(ref_sig) <= params, (ref_sig, params)                       # same names (lib/models/__init__.py:3)
    assert list(inspect.signature(load_tokenhmr).parameters)[:len(ref_sig)] == ref_sig      # same positional order
    used = set()
    for script in ("eval.py", "demo.py", "track.py"):
        for n in ast.walk(_parse(script)):
            if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "load_tokenhmr":
                assert not n.args, f"{script}: positional call"          # the scripts call it by keyword
                used |= {k.arg for k in n.keywords}
    assert used and used <= params, (used, params)


def test_every_model_attribute_the_scripts_touch_exists_on_the_facade():
    from tokenhmr_amd.model import TokenHMR
    init = ast.parse(inspect.getsource(TokenHMR))
    inst_attrs = {n.attr for n in ast.walk(init) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "self"}
    have = set(dir(TokenHMR)) | inst_attrs
    touched = set()
    for script in ("eval.py", "demo.py", "track.py"):
        touched |= _attrs(_parse(script), "model")
    # track.py wraps the model in a PHALP predictor class; what it calls on it is forward(batch) through __call__
    assert touched, "no model.<attr> use found in the reference scripts"
    missing = {a for a in touched if a not in have}
    assert not missing, f"the reference scripts use model.{missing}, the facade has no such attribute"
    # model.smpl.faces (demo.py:52)
    assert any(isinstance(n, ast.Attribute) and n.attr == "faces" and isinstance(n.value, ast.Attribute) and n.value.attr == "smpl"
               for n in ast.walk(_parse("demo.py")))
    from tokenhmr_amd.model import _SmplHandle
    assert "faces" in {n.attr for n in ast.walk(ast.parse(inspect.getsource(_SmplHandle))) if isinstance(n, ast.Attribute)}


def test_every_output_key_the_reference_reads_is_produced():
    keys, nested = _facade_output_keys()
    read = set()
    read |= _subscript_keys(_parse("demo.py"), {"out"})
    read |= _subscript_keys(_parse("eval.py"), {"out"})
    read |= _subscript_keys(_parse("track.py"), {"model_out"})
    read |= _subscript_keys(_parse("lib/utils/pose_utils.py"), {"output"})          # Evaluator.__call__(output, batch)
    assert {"pred_vertices", "pred_keypoints_3d", "pred_cam", "pred_cam_t"} <= read, read
    missing = read - keys
    assert not missing, f"the reference reads output keys {missing} that TokenHMR._pack does not produce"
    # the output dict of the reference's forward_step (tokenhmr.py:156-188): every key it sets at inference is produced too
    fs = next(n for n in ast.walk(_parse("lib/models/tokenhmr.py")) if isinstance(n, ast.FunctionDef) and n.name == "forward_step")
    set_keys = set()
    for n in ast.walk(fs):
        if isinstance(n, ast.Assign):
            for t in n.targets:
                if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Name) and t.value.id == "output" and isinstance(t.slice, ast.Constant):
                    set_keys.add(t.slice.value)
    assert set_keys, "no output[...] assignment found in forward_step"
    assert set_keys <= keys, (set_keys - keys)
    assert nested.get("pred_smpl_params") == {"global_orient", "body_pose", "betas"}


def test_tokenizer_dropins_keep_the_reference_constructors_and_call_sites():
    """tokenhmr_amd.tokenizer.DecodeTokens / EncodeTokens: the reference classes' constructor parameters come first, in order, with the same
    defaults (vanilla_pose_vqvae.py:258-262, :304-307), and the one place the reference instantiates and calls DecodeTokens
    (heads/token_classifier.py: Proxy) passes nothing the drop-in does not take."""
    from tokenhmr_amd import tokenizer as T
    tok_src = os.path.join(os.path.dirname(REF), "tokenization", "models", "vanilla_pose_vqvae.py")
    with open(tok_src) as f:
        tree = ast.parse(f.read())
    for cls_name in ("DecodeTokens", "EncodeTokens"):
        cls = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls_name)
        init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
        ref_args = [a.arg for a in init.args.args][1:]
        ref_defaults = [ast.literal_eval(d) for d in init.args.defaults]
        mine = inspect.signature(getattr(T, cls_name).__init__)
        names = list(mine.parameters)[1:]
        assert names[:len(ref_args)] == ref_args, (cls_name, names, ref_args)
        assert [mine.parameters[a].default for a in ref_args] == ref_defaults, cls_name
        fwd = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
        assert len(fwd.args.args) == 2 and len(inspect.signature(getattr(T, cls_name).forward).parameters) == 2      # (self, x)
    # the reference's own call site (token_classifier.py:85): eval('VanillaDecodeTokens')(tokenizer_checkpoint_path), handed to Proxy, which
    # moves it with .to(x.device) and calls it with the token probabilities (:12-20)
    tc = _parse("lib/models/heads/token_classifier.py")
    calls = [n for n in ast.walk(tc) if isinstance(n, ast.Call) and isinstance(n.func, ast.Call) and isinstance(n.func.func, ast.Name)
             and n.func.func.id == "eval"]
    assert calls, "the reference no longer instantiates the tokenizer through eval(...)(ckpt_path) in token_classifier.py?"
    ok = set(inspect.signature(T.DecodeTokens.__init__).parameters)
    for c in calls:
        assert len(c.args) == 1 and all(k.arg in ok for k in c.keywords), ast.dump(c)
    proxy = next(n for n in ast.walk(tc) if isinstance(n, ast.ClassDef) and n.name == "Proxy")
    used = {n.attr for n in ast.walk(proxy) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Attribute) and n.value.attr == "tokenizer"}
    assert used <= {"to"} and all(hasattr(T.DecodeTokens, a) for a in used | {"__call__"}), used
