"""-m gpu: the data-parallel collectives through the C ABI (thmr_pack_records / thmr_bcast_weights / thmr_allgather_records,
include/tokenhmr_hip.h) on a real RCCL communicator created WITHOUT torch.distributed — what a non-torch host would do
(ncclGetUniqueId + ncclCommInitRank through ctypes on the process's librccl; world size 1 on the one GPU of the box)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def _rccl():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path if os.path.exists(path) else "librccl.so.1")
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


def test_pack_kernel_equals_concatenation(built_lib, cuda_dev):
    from tokenhmr_amd import dist as D
    g = torch.Generator().manual_seed(0)
    B = 5
    o = {"pred_vertices": torch.randn(B, 6890, 3, generator=g), "pred_keypoints_3d": torch.randn(B, 44, 3, generator=g),
         "pred_keypoints_2d": torch.randn(B, 44, 2, generator=g), "rotmat": torch.randn(B, 24, 3, 3, generator=g),
         "betas": torch.randn(B, 10, generator=g), "pred_cam": torch.randn(B, 3, generator=g), "pred_cam_t": torch.randn(B, 3, generator=g),
         "token_idx": torch.randint(0, 2048, (B, 160), generator=g, dtype=torch.int32)}
    ref = D.pack_records(o)                                        # CPU: torch.cat
    got = D.pack_records({k: v.to(cuda_dev) for k, v in o.items()})   # GPU: thmr_pack_records
    assert got.shape == (B, D.RECORD_WORDS) and torch.equal(got.cpu().view(torch.int32), ref.view(torch.int32))
    back = D.unpack_records(got)
    assert all(torch.equal(back[k].cpu(), o[k]) for k in o)


def test_cabi_collectives_on_a_raw_rccl_communicator(built_lib, cuda_dev):
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W, dist as D
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(cuda_dev)
    rccl = _rccl()
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        cfg = HMRConfig(vit_depth=1, dec_depth=1)
        eng = Engine(cfg, max_batch=4, device=cuda_dev)
        eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
        eng.load_smpl(make_synthetic_smpl(cfg, 0))
        eng.finalize()
        lib = built_lib
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        before = eng.weight_arena.clone()
        assert lib.thmr_bcast_weights(eng.h, comm, 0, st) == 0, lib.thmr_collective_last_error()
        torch.cuda.synchronize()
        assert torch.equal(before, eng.weight_arena)              # world size 1: the arena is its own broadcast
        img = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(2)).to(cuda_dev)
        o = eng.forward(img)
        rec = D.pack_records(o)
        recv = torch.empty_like(rec)
        assert lib.thmr_allgather_records(comm, C.c_void_p(rec.data_ptr()), 4, C.c_void_p(recv.data_ptr()), st) == 0, lib.thmr_collective_last_error()
        torch.cuda.synchronize()
        out = D.unpack_records(recv)
        for k in out:
            assert torch.equal(out[k], o[k]), k
        assert lib.thmr_allgather_records(None, C.c_void_p(rec.data_ptr()), 4, C.c_void_p(recv.data_ptr()), st) < 0   # null communicator
    finally:
        rccl.ncclCommDestroy(comm)


def test_receiver_of_a_broadcast_made_before_the_root_finalized(built_lib, cuda_dev):
    """The round-2 regression: bench.py and the INTEGRATION.md recipe broadcast the arena BEFORE rank 0
    finalizes, and the resample tables / padding row / flag words were written by finalize(assume_all_loaded=0) only — ranks
    != 0 ran on uninitialised tables.  One GPU per box, so the receiver is a second engine whose arena starts as garbage and is
    filled by exactly the bytes a broadcast would deliver at that moment (a device copy of the root's arena after load, before
    its finalize).  Its outputs must equal the root's bit for bit and the CPU oracle's within the usual tolerance."""
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W, _cabi
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine
    from oracle import tokenhmr_oracle as O
    cfg = HMRConfig(vit_depth=1, dec_depth=2)
    sd, tok, smpl = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0), make_synthetic_smpl(cfg, 0)
    tok_full = dict(tok)
    tok_full.update(W.make_synthetic_encoder(cfg, 0))
    root = Engine(cfg, max_batch=3, device=cuda_dev)
    recv = Engine(cfg, max_batch=3, device=cuda_dev)
    recv.weight_arena.view(torch.int32).fill_(0x7FC0DEAD)          # NaN patterns / wild indices wherever nothing arrives
    torch.cuda.synchronize()
    # a receiver whose arena never got a loaded model must refuse to finalize (magic word), not run on garbage
    with pytest.raises(_cabi.EngineError, match="does not carry a loaded model"):
        recv.finalize(assume_all_loaded=True)
    root.load_state(sd, tok_full)
    root.load_smpl(smpl)
    torch.cuda.synchronize()
    recv.weight_arena.copy_(root.weight_arena)                       # "the broadcast", before the root finalizes
    torch.cuda.synchronize()
    root.finalize()
    recv.finalize(assume_all_loaded=True)
    img = torch.randn(3, 3, 256, 256, generator=torch.Generator().manual_seed(11))
    a, b = root.forward(img.to(cuda_dev), taps=True), recv.forward(img.to(cuda_dev), taps=True)
    torch.cuda.synchronize()
    root.status(), recv.status()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ref = O.forward(img, sd, tok, smpl, cfg)
    assert (b["pred_vertices"].cpu() - ref["pred_vertices"]).abs().max() < 1e-4
    assert (b["pred_keypoints_2d"].cpu() - ref["pred_keypoints_2d"]).abs().max() < 1e-3
    # the "encoder present" flag travelled with the arena as well
    pose = torch.randn(2, 21, 6, generator=torch.Generator().manual_seed(12)).to(cuda_dev)
    assert torch.equal(root.encode_tokens(pose), recv.encode_tokens(pose))
