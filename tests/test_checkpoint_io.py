"""Reading the reference's checkpoints WITHOUT yacs / omegaconf / pytorch_lightning (not-gpu).

tokenizer.pth pickles a yacs CfgNode (tokenization/utils/eval_poseVQ.py:118-125 -> vanilla_pose_vqvae.py:265-278) and the
Lightning checkpoint's hyper_parameters hold config nodes (tokenhmr.py:42).  The files here are written with classes that
live under those exact module paths and are removed again before reading: a plain torch.load must fail on them with
ModuleNotFoundError (that is the failure a user of the released files hits in this image), tokenhmr_amd must not."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from _ref_files import FOREIGN, foreign_modules, write_reference_files


@pytest.fixture(scope="module")
def tiny():
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    cfg = HMRConfig(vit_depth=1, dec_depth=2)
    return cfg, W.make_synthetic_state(cfg, 3), W.make_synthetic_tokenizer(cfg, 3), make_synthetic_smpl(cfg, 3)


def test_premise_plain_torch_load_fails_without_yacs(tiny, tmp_path):
    cfg, sd, tok, smpl = tiny
    ck, _ = write_reference_files(tmp_path, cfg, sd, tok, smpl)
    assert not any(m in sys.modules for m in FOREIGN)
    for path in (ck, str(tmp_path / "tokenizer.pth")):
        with pytest.raises(ModuleNotFoundError):
            torch.load(path, map_location="cpu", weights_only=False)
        with pytest.raises(Exception):                      # the torch >= 2.6 default refuses the config classes as well
            torch.load(path, map_location="cpu")


@pytest.mark.parametrize("legacy", [False, True], ids=["zip", "legacy"])
def test_read_reference_files_without_foreign_packages(tiny, tmp_path, legacy):
    from tokenhmr_amd.model import read_reference_files
    from tokenhmr_amd import ckpt_io
    cfg, sd, tok, smpl = tiny
    ck, yml = write_reference_files(tmp_path, cfg, sd, tok, smpl, legacy_format=legacy)
    hcfg, state, tk, sm, mcfg = read_reference_files(ck, yml)
    assert not any(m in sys.modules for m in FOREIGN), "the loader imported a foreign package"
    assert hcfg.vit_depth == 1 and hcfg.dec_depth == 2 and mcfg.MODEL.BBOX_SHAPE == [192, 256]
    assert set(state) == set(sd) and all(torch.equal(state[k], sd[k]) for k in sd)
    assert set(tk) == set(tok) and all(torch.equal(tk[k], tok[k]) for k in tok)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J19_regressor"):
        assert torch.allclose(sm[k], smpl[k], atol=0, rtol=0), k
    # the config nodes are there as inert data, reachable the way the reference reaches them
    t = ckpt_io.load_checkpoint(str(tmp_path / "tokenizer.pth"))
    assert type(t["hparams"]).__mro__[1] is ckpt_io.InertNode and t["hparams"]._inert_origin == ("yacs.config", "CfgNode")
    assert t["hparams"].ARCH.NB_CODE == [2048] and t["hparams"].ARCH.ROT_TYPE == "rot6d" and t["hparams"].DATA.BATCH_SIZE == 256
    assert ckpt_io.tokenizer_arch(t) == ckpt_io.RELEASE_ARCH
    light = ckpt_io.load_checkpoint(ck)
    assert light["epoch"] == 7 and light["hyper_parameters"]["cfg"].MODEL.BACKBONE.TYPE == "vit"
    assert light["hyper_parameters"]["hydra_cfg"]._content["trainer"]["devices"] == 8
    assert float(light["np_scalar"]) == 2.5 and np.array_equal(light["np_array"], np.arange(4))
    assert torch.equal(light["optimizer_states"][0]["state"][0]["exp_avg"], torch.zeros(3))


def test_tokenizer_with_encoder_half_is_picked_up(tiny, tmp_path):
    from tokenhmr_amd.model import read_reference_files
    from tokenhmr_amd import weights as W
    cfg, sd, tok, smpl = tiny
    enc = W.make_synthetic_encoder(cfg, 3)
    full = dict(tok)
    full.update(enc)
    ck, yml = write_reference_files(tmp_path, cfg, sd, full, smpl)
    _, _, tk, _, _ = read_reference_files(ck, yml)
    assert all(k in tk and torch.equal(tk[k], enc[k]) for k in enc)


@pytest.mark.parametrize("key,val", [("NB_CODE", [1024]), ("CODE_DIM", 512), ("WIDTH", 256), ("DEPTH", 3), ("DILATION_RATE", 2),
                                      ("DOWN_T", 2), ("TOKEN_SIZE_MUL", 3), ("TOKEN_SIZE_DIV", 2), ("ROT_TYPE", "rotmat")])
def test_tokenizer_arch_mismatch_names_the_key(tiny, tmp_path, key, val):
    from tokenhmr_amd.model import read_reference_files
    cfg, sd, tok, smpl = tiny
    ck, yml = write_reference_files(tmp_path, cfg, sd, tok, smpl, arch_overrides={key: val})
    with pytest.raises(ValueError, match=f"ARCH.{key}"):
        read_reference_files(ck, yml)


def test_tokenizer_without_hparams_or_arch_key(tiny, tmp_path):
    from tokenhmr_amd import ckpt_io
    assert ckpt_io.tokenizer_arch({"net": {}}) is None                   # this repo's bare fixtures: nothing to compare
    ckpt_io.check_tokenizer_arch(None)
    with pytest.raises(KeyError, match="ARCH"):
        ckpt_io.tokenizer_arch({"net": {}, "hparams": {"DATA": {}}})
    with pytest.raises(KeyError, match="WIDTH"):
        ckpt_io.tokenizer_arch({"net": {}, "hparams": {"ARCH": {k: v for k, v in ckpt_io.RELEASE_ARCH.items() if k != "WIDTH"}}})


def test_strict_and_lenient_unexpected_keys(tiny, tmp_path):
    """misc.py:228-238: the reference logs the strict-load error and carries on with every matching tensor copied.
    strict=True (default here) is a hard error; strict=False = the reference's behaviour for UNEXPECTED keys."""
    from tokenhmr_amd.model import read_reference_files
    cfg, sd, tok, smpl = tiny
    extra = {"smpl_head.some_buffer": torch.zeros(3), "backbone.cls_token": torch.zeros(1, 1, 1280)}
    ck, yml = write_reference_files(tmp_path, cfg, sd, tok, smpl, extra_state=extra)
    with pytest.raises(KeyError, match="smpl_head.some_buffer|backbone.cls_token"):
        read_reference_files(ck, yml)
    with pytest.warns(RuntimeWarning, match="Mismatch in statedict"):
        _, state, _, _, _ = read_reference_files(ck, yml, strict=False)
    assert set(state) == set(sd)
    # a MISSING tensor is an error in both modes (the reference would keep a random initialisation and run)
    bad = {k: v for k, v in sd.items() if k != "backbone.blocks.0.attn.qkv.bias"}
    sub = tmp_path / "bad"
    sub.mkdir()
    ck2, yml2 = write_reference_files(sub, cfg, bad | {"smpl_head.init_cam": sd["smpl_head.init_cam"]}, tok, smpl)
    for strict in (True, False):
        with pytest.raises(KeyError, match="qkv.bias"):
            read_reference_files(ck2, yml2, strict=strict)


def test_missing_files_and_entries(tiny, tmp_path):
    from tokenhmr_amd.model import read_reference_files
    cfg, sd, tok, smpl = tiny
    ck, yml = write_reference_files(tmp_path, cfg, sd, tok, smpl)
    with pytest.raises(FileNotFoundError, match="Missing full pretrained model"):       # misc.py:252-254
        read_reference_files(str(tmp_path / "nope.ckpt"), yml)
    torch.save({"weights": {}}, tmp_path / "no_sd.ckpt")
    with pytest.raises(KeyError, match="state_dict"):
        read_reference_files(str(tmp_path / "no_sd.ckpt"), yml)
    torch.save({"model": {}}, tmp_path / "tokenizer.pth")
    with pytest.raises(KeyError, match="'net'"):
        read_reference_files(ck, yml)


class _Gadget:
    def __reduce__(self):
        return (os.system, ("echo pwned > /tmp/thmr_pickle_gadget",))


class _NumpyGadget:                      # a FUNCTION of an allowed module (numpy) that is not an array-reconstruction helper
    def __reduce__(self):
        return (np.load, ("/tmp/thmr_does_not_exist.npy",))


class _TorchGadget:                      # torch.hub / torch.load-style entry points are not tensor rebuild helpers either
    def __reduce__(self):
        return (torch.hub.load, ("someone/repo", "model"))


class _NestedArchiveGadget:
    """torch.storage._load_from_bytes is torch.load(io.BytesIO(b), weights_only=False) — an UNRESTRICTED nested un-pickle — in torch
    itself; a pickle that REDUCEs it with an inner archive holding an os.system gadget ran the command through round 3's loader
    (ADVICE r3)."""

    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        import io
        buf = io.BytesIO()
        torch.save({"inner": _MarkerGadget(self.marker), "t": torch.arange(3.0)}, buf)
        return (torch.storage._load_from_bytes, (buf.getvalue(),))


class _MarkerGadget:
    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        return (os.system, (f"echo pwned > {self.marker}",))


class _MemmapGadget:                     # numpy.memmap is a TYPE of an allowed module whose constructor creates / truncates files
    def __init__(self, path):
        self.path = path

    def __reduce__(self):
        return (np.memmap, (self.path, "uint8", "w+", 0, (16,)))


def test_restricted_unpickler_nested_archive_and_memmap(tmp_path):
    """Negative tests for the two bypasses ADVICE r3 demonstrated: a nested archive through `_load_from_bytes` is read by the SAME
    restricted un-pickler (its tensors load, its gadget stays data), and numpy.memmap does not resolve (no file is created)."""
    from tokenhmr_amd import ckpt_io
    marker = str(tmp_path / "nested_gadget_ran")
    victim = str(tmp_path / "memmap_victim.bin")
    torch.save({"state_dict": {"w": torch.ones(2)}, "nested": _NestedArchiveGadget(marker), "mm": _MemmapGadget(victim)},
               tmp_path / "nested.ckpt")
    c = ckpt_io.load_checkpoint(str(tmp_path / "nested.ckpt"))
    assert not os.path.exists(marker), "the nested archive's os.system gadget was executed"
    assert not os.path.exists(victim), "numpy.memmap was resolved and created a file"
    assert isinstance(c["nested"], dict) and torch.equal(c["nested"]["t"], torch.arange(3.0))
    assert isinstance(c["nested"]["inner"], ckpt_io.InertNode) and c["nested"]["inner"]._inert_origin[1] == "system"
    assert isinstance(c["mm"], ckpt_io.InertNode) and c["mm"]._inert_origin == ("numpy", "memmap")
    # a tensor pickled OUTSIDE torch.save legitimately reduces through _load_from_bytes: still loads
    import pickle as _p
    blob = _p.dumps({"t": torch.arange(5.0)})
    got = ckpt_io.RestrictedUnpickler(__import__("io").BytesIO(blob)).load()
    assert torch.equal(got["t"], torch.arange(5.0))


def test_restricted_unpickler_never_resolves_code(tmp_path):
    """The reference's torch.load un-pickles arbitrary globals; this loader cannot: anything that is not a tensor / array /
    plain container becomes inert data."""
    from tokenhmr_amd import ckpt_io
    marker = "/tmp/thmr_pickle_gadget"
    if os.path.exists(marker):
        os.remove(marker)
    torch.save({"state_dict": {"w": torch.ones(2)}, "evil": _Gadget(), "getattr": getattr, "eval": eval, "np": _NumpyGadget(),
                "hub": _TorchGadget(), "ok_np": np.float32(1.5), "ok_arr": np.eye(2), "ok_dtype": np.dtype("int16"), "ok_size": torch.Size([2, 3]),
                "ok_tdtype": torch.bfloat16}, tmp_path / "evil.ckpt")
    c = ckpt_io.load_checkpoint(str(tmp_path / "evil.ckpt"))
    assert not os.path.exists(marker)
    assert isinstance(c["np"], ckpt_io.InertNode) and c["np"]._inert_origin == ("numpy", "load")
    assert isinstance(c["hub"], ckpt_io.InertNode) and c["hub"]._inert_origin[0] == "torch.hub"
    assert float(c["ok_np"]) == 1.5 and np.array_equal(c["ok_arr"], np.eye(2)) and c["ok_dtype"] == np.dtype("int16")
    assert c["ok_size"] == torch.Size([2, 3]) and c["ok_tdtype"] is torch.bfloat16
    assert isinstance(c["evil"], ckpt_io.InertNode) and c["evil"]._inert_args[0] == ("echo pwned > /tmp/thmr_pickle_gadget",)
    assert issubclass(c["getattr"], ckpt_io.InertNode) and issubclass(c["eval"], ckpt_io.InertNode)
    assert torch.equal(c["state_dict"]["w"], torch.ones(2))
    with pytest.raises(pickle.PicklingError):
        pickle.dumps(c["evil"])
