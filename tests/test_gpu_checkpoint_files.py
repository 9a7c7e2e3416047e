"""load_tokenhmr() against files in the reference's on-disk formats (-m gpu): a Lightning-style checkpoint with unrelated
keys and config nodes under 'hyper_parameters', the tokenizer checkpoint with a CfgNode pickled under the module path
`yacs.config` (not importable when the file is read), the SMPL model pickle with a chumpy-typed array and a scipy sparse
J_regressor, SMPL_to_J19.pkl, smpl_mean_params.npz and model_config.yaml — read exactly where the reference reads them
(lib/models/__init__.py:3-26, lib/utils/misc.py:242-256, vanilla_pose_vqvae.py:265-278,299-301, smpl_wrapper.py:11-25).
The model loaded from disk must equal the model built from the same tensors in memory, bit for bit.  The file-format half of
this (no GPU) is tests/test_checkpoint_io.py."""
import sys

import pytest
import torch

from _ref_files import FOREIGN, write_reference_files

pytestmark = pytest.mark.gpu


def test_load_tokenhmr_from_reference_style_files(built_lib, cuda_dev, tmp_path):
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.model import TokenHMR, load_tokenhmr

    cfg = HMRConfig(vit_depth=1, dec_depth=2)
    sd, tok, smpl = W.make_synthetic_state(cfg, 3), W.make_synthetic_tokenizer(cfg, 3), make_synthetic_smpl(cfg, 3)
    tok_full = dict(tok)
    tok_full.update(W.make_synthetic_encoder(cfg, 3))                    # a released tokenizer.pth carries the encoder half too
    ck, yml = write_reference_files(tmp_path, cfg, sd, tok_full, smpl)
    assert not any(m in sys.modules for m in FOREIGN)

    model, mcfg = load_tokenhmr(ck, yml, max_batch=2, device=cuda_dev)
    assert not any(m in sys.modules for m in FOREIGN)
    assert mcfg.MODEL.BBOX_SHAPE == [192, 256] and model.engine.cfg.vit_depth == 1 and model.engine.cfg.dec_depth == 2
    ref = TokenHMR.from_state(cfg, sd, tok_full, smpl, max_batch=2, device=cuda_dev)
    img = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(cuda_dev)
    a, b = model({"img": img}), ref({"img": img})
    for k in ("pred_vertices", "pred_keypoints_3d", "pred_keypoints_2d", "pred_cam", "cls_logits_softmax"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["pred_smpl_params"]["body_pose"], b["pred_smpl_params"]["body_pose"])
    # the encoder half travelled: EncodeTokens (vanilla_pose_vqvae.py:334-342) works on the engine loaded from disk
    pose = torch.randn(2, 21, 6, generator=torch.Generator().manual_seed(2)).to(cuda_dev)
    assert torch.equal(model.engine.encode_tokens(pose), ref.engine.encode_tokens(pose))

    # the drop-in's DEFAULT is the mode the benchmark times: load_tokenhmr() without a mode argument (and without $THMR_VIT_GEMM) delivers an
    # engine in "split3", whose 4-crop call runs the bf16-pipe kernels — equal to the in-memory model's, different from the exact-fp32 opt-out's
    import os
    assert "THMR_VIT_GEMM" not in os.environ
    model4, _ = load_tokenhmr(ck, yml, max_batch=4, device=cuda_dev)
    assert model4.engine.vit_gemm() == "split3"
    ref4 = TokenHMR.from_state(cfg, sd, tok_full, smpl, max_batch=4, device=cuda_dev)
    img4 = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(5)).to(cuda_dev)
    a4, b4 = model4({"img": img4}), ref4({"img": img4})
    optout, _ = load_tokenhmr(ck, yml, max_batch=4, device=cuda_dev, vit_gemm="f32")
    assert optout.engine.vit_gemm() == "f32"
    c4 = optout({"img": img4})
    for k in ("pred_vertices", "pred_keypoints_3d", "cls_logits_softmax"):
        assert torch.equal(a4[k], b4[k]), k
    assert not torch.equal(a4["pred_vertices"], c4["pred_vertices"]) and (a4["pred_vertices"] - c4["pred_vertices"]).abs().max() < 1e-4
    assert int((a4["token_idx"] != c4["token_idx"]).sum()) <= 2          # 640 tokens: only a near-tie may flip between the two arithmetics
    del model4, ref4, optout

    # strict-load failures are loud (load_state_dict(strict=True) semantics, misc.py:246-250)
    bad = dict(sd)
    bad.pop("backbone.blocks.0.attn.qkv.bias")
    sub = tmp_path / "bad"
    sub.mkdir()
    ck2, yml2 = write_reference_files(sub, cfg, bad, tok, smpl)
    with pytest.raises((KeyError, ValueError, RuntimeError)):
        load_tokenhmr(ck2, yml2, max_batch=2, device=cuda_dev)
    # an unexpected smpl_head tensor: error by default, warning + identical model with strict=False (misc.py:228-238)
    sub = tmp_path / "extra"
    sub.mkdir()
    ck3, yml3 = write_reference_files(sub, cfg, sd, tok, smpl, extra_state={"smpl_head.extra_buffer": torch.zeros(5)})
    with pytest.raises(KeyError, match="extra_buffer"):
        load_tokenhmr(ck3, yml3, max_batch=2, device=cuda_dev)
    with pytest.warns(RuntimeWarning, match="Mismatch in statedict"):
        lenient, _ = load_tokenhmr(ck3, yml3, max_batch=2, device=cuda_dev, strict=False)
    c = lenient({"img": img})
    assert torch.equal(c["pred_vertices"], b["pred_vertices"])
