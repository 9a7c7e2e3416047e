"""Tokenizer encode / decode paths (rows a10, a11, a15 as stand-alone entry points).
CPU: the oracle against golden indices produced by the reference's own PoseSPEncoderV1 + QuantizeEMAReset.
GPU: thmr_encode_tokens / thmr_vq_decode through the C ABI against oracle and golden."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from tokenhmr_amd.config import HMRConfig, RELEASE
from tokenhmr_amd import weights as W
from oracle import tokenhmr_oracle as O
from oracle.gen_golden_encode import make_pose


def test_encode_oracle_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN_DIR, "encode_small.npz"))
    enc, tok = W.make_synthetic_encoder(RELEASE, 0), W.make_synthetic_tokenizer(RELEASE, 0)
    assert abs(W.checksum(enc) - g["checksum"][0]) < 1e-9 * abs(g["checksum"][0])
    with torch.no_grad():
        idx, lat, _ = O.vq_encode(make_pose(), enc, tok["quantizer.codebook"])
    assert np.abs(lat[::7].numpy() - g["latent_sample"]).max() < 1e-5
    assert np.array_equal(idx.numpy().astype(np.int32), g["idx"])


def test_encoder_upsample_indices_match_torch():
    """nn.Upsample(size=40) and nn.Upsample(scale_factor=2) of the encoder (vanilla_pose_vqvae.py:70,75)."""
    x = torch.arange(21, dtype=torch.float32).view(1, 1, 21)
    assert torch.equal(torch.nn.Upsample(40)(x).view(-1).long(), O.nearest_index(21, 40))
    for t in (40, 80, 160):
        x = torch.arange(t, dtype=torch.float32).view(1, 1, t)
        assert torch.equal(torch.nn.Upsample(scale_factor=2, mode="nearest")(x).view(-1).long(), torch.arange(2 * t) // 2)


def test_encoder_spec_names():
    names = [n for n, *_ in W.tokenizer_encoder_spec(RELEASE)]
    assert len(names) == 22 and names[0] == "encoder.encoder.0.weight" and "encoder.encoder.14.0.weight" in names


@pytest.mark.gpu
def test_gpu_encode_and_decode(built_lib, cuda_dev):
    from tokenhmr_amd import weights as Wt
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine
    from tokenhmr_amd._cabi import EngineError
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    sd, tok, smpl = Wt.make_synthetic_state(cfg, 0), Wt.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
    enc = Wt.make_synthetic_encoder(cfg, 0)
    eng = Engine(cfg, max_batch=8, device=cuda_dev)
    eng.load_state(sd, tok)
    eng.load_smpl(smpl)
    eng.finalize()
    with pytest.raises(EngineError):                      # encoder weights absent -> loud error, not a fallback
        eng.encode_tokens(make_pose().to(cuda_dev))
    eng.load_state({}, enc)
    eng.finalize()
    g = np.load(os.path.join(GOLDEN_DIR, "encode_small.npz"))
    pose = make_pose()
    idx, lat = eng.encode_tokens(pose, want_latent=True)
    with torch.no_grad():
        oidx, olat, dist = O.vq_encode(pose, enc, tok["quantizer.codebook"])
    assert (lat.cpu().reshape(-1, 256) - olat).abs().max() < 1e-4
    two = dist.topk(2, dim=-1, largest=False).values
    safe = (two[:, 1] - two[:, 0]) > 1e-2                   # expanded-form distances are O(1e2-1e3) in fp32
    got = idx.cpu().reshape(-1).long()
    assert torch.equal(got[safe], oidx[safe]) and safe.float().mean() > 0.95
    assert np.array_equal(got.numpy().astype(np.int32)[safe.numpy()], g["idx"][safe.numpy()])   # reference's own indices
    # ragged batch (B=5 of max 8) agrees with per-item calls
    p5 = make_pose(5, seed=3)
    i5 = eng.encode_tokens(p5)
    i1 = eng.encode_tokens(p5[3:4])
    assert torch.equal(i5[3], i1[0])
    # decode: hard (one-hot) and soft probabilities vs the oracle's DecodeTokens restatement; tolerance 1e-4
    onehot = torch.zeros(3, 160, 2048)
    onehot.scatter_(2, idx.cpu().long().unsqueeze(-1), 1.0)
    dec = eng.vq_decode(onehot)
    with torch.no_grad():
        ref = O.vq_decode(onehot, tok, cfg)
    assert (dec.cpu() - ref).abs().max() < 1e-4
    soft = torch.softmax(torch.randn(2, 160, 2048, generator=torch.Generator().manual_seed(4)), -1)
    with torch.no_grad():
        ref = O.vq_decode(soft, tok, cfg)
    assert (eng.vq_decode(soft).cpu() - ref).abs().max() < 1e-4
    # an engine that RECEIVED the arena (broadcast, or a shared arena) and only finalised it can encode too: "the encoder tensors
    # are present" travels with the arena as a flag word (round 1 kept it host-side, so broadcast receivers could not encode)
    e2 = Engine(cfg, max_batch=8, device=cuda_dev, weight_arena=eng.weight_arena)
    e2.finalize(assume_all_loaded=True)
    assert torch.equal(e2.encode_tokens(pose), idx)
    e2.close()
    eng.close()


def _write_tokenizer_file(tmp_path, cfg, with_encoder=True, arch_overrides=None):
    """tokenizer.pth exactly as the reference writes it ({'net', 'hparams': yacs CfgNode}, eval_poseVQ.py:118-125), yacs absent at read time"""
    from _ref_files import write_reference_files
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    sd, tok, smpl = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
    net = dict(tok)
    if with_encoder:
        net.update(W.make_synthetic_encoder(cfg, 0))
    write_reference_files(tmp_path, cfg, sd, net, smpl, arch_overrides=arch_overrides)
    return str(tmp_path / "tokenizer.pth"), tok


def test_tokenizer_dropins_reject_bad_files_before_touching_a_gpu(tmp_path):
    """DecodeTokens / EncodeTokens (tokenhmr_amd/tokenizer.py) read hparams.ARCH where the reference does (vanilla_pose_vqvae.py:265-278) and
    refuse a tokenizer of another architecture, a file without 'net', and — EncodeTokens — a file without the encoder half."""
    from tokenhmr_amd.tokenizer import DecodeTokens, EncodeTokens
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir(); (tmp_path / "c").mkdir()
    path, _ = _write_tokenizer_file(tmp_path / "a", cfg, arch_overrides={"NB_CODE": 1024})
    with pytest.raises(ValueError, match="NB_CODE"):
        DecodeTokens(path)
    path, _ = _write_tokenizer_file(tmp_path / "b", cfg, with_encoder=False)
    with pytest.raises(KeyError, match="encoder"):
        EncodeTokens(path)
    torch.save({"state_dict": {}}, tmp_path / "c" / "x.pth")
    with pytest.raises(KeyError, match="net"):
        DecodeTokens(str(tmp_path / "c" / "x.pth"))


@pytest.mark.gpu
def test_gpu_tokenizer_dropins(built_lib, cuda_dev, tmp_path):
    """The reference's stand-alone tokenizer classes as drop-ins: same constructor, same call, same result layout — against the oracle
    (pinned to the reference's own PoseSPEncoderV1 / PoseSPDecoderV1 / QuantizeEMAReset) on a file in the reference's format."""
    from tokenhmr_amd.tokenizer import DecodeTokens, EncodeTokens
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    path, tok = _write_tokenizer_file(tmp_path, cfg)
    dec = DecodeTokens(path, device=cuda_dev, max_batch=4).eval().to(cuda_dev)
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(4.0 * torch.randn(9, 160, 2048, generator=g), -1)           # 9 > max_batch: chunked
    out = dec(probs.to(cuda_dev))
    assert out.shape == (9, 21, 6) and out.dtype == torch.float32 and out.is_cuda
    with torch.no_grad():
        ref = O.vq_decode(probs, tok, cfg)
    assert (out.cpu() - ref).abs().max() < 1e-4
    enc = EncodeTokens(path, device=cuda_dev, max_batch=4)
    pose = make_pose()
    idx = enc(pose.to(cuda_dev))
    with torch.no_grad():
        ridx, _, _ = O.vq_encode(pose, W.make_synthetic_encoder(cfg, 0), tok["quantizer.codebook"])
    assert idx.dtype == torch.int64 and idx.shape == (pose.shape[0] * 160,)
    assert torch.equal(idx.cpu().view(pose.shape[0], -1), ridx.view(pose.shape[0], -1).long())
    # sharing a loaded model's engine instead of building one
    shared = DecodeTokens(engine=dec.engine)
    assert torch.equal(shared(probs[:3].to(cuda_dev)), out[:3])
    with pytest.raises(ValueError):
        dec(torch.zeros(2, 160, 100, device=cuda_dev))
