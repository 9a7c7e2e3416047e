"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the C++ and Python
copies of the checkpoint contract agree, error paths are loud, and host-side logic (sharding,
record packing, validation) is correct.  No compute kernels are launched (there is no GPU here)."""
import ctypes as C
import math
import os

import pytest
import torch

from tokenhmr_amd import _cabi, weights as W, dist as D
from tokenhmr_amd.config import HMRConfig, RELEASE


def test_library_exports_every_declared_symbol(built_lib):
    syms = _cabi.declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(built_lib, s), s
    assert built_lib.thmr_abi_version() == _cabi.ABI_VERSION
    assert b"gfx950" in built_lib.thmr_build_info()


def test_loaded_library_was_built_from_the_current_sources(built_lib):
    """Stale-binary guard: the library reports the content hash of csrc/ + include/ + compiler flags it was compiled from
    (thmr_build_info), and that must be the hash of the sources in the tree right now.  (The .so is git-ignored but travels
    to the GPU box; without this a forgotten rebuild tests yesterday's kernels against today's sources.)"""
    import __graft_entry__
    info = built_lib.thmr_build_info().decode()
    assert f"src:{__graft_entry__.source_hash()}" in info, info


def test_library_has_gfx950_code_object(built_lib):
    with open(_cabi.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"gfx950" in blob and b"gemm_f32_kernel" in blob and b"vit_attention_kernel" in blob
    # round 2: persistent attention, tiny-M GEMM, the decoder kernel that also runs the mixer stack
    for sym in (b"vit_attention_persistent_kernel", b"gemm_tiny_kernel", b"gemm_ring_kernel", b"decoder_persistent_kernel", b"mixer_stack_kernel"):
        assert sym in blob, sym


def test_one_hip_runtime_per_process_whatever_the_import_order(built_lib):
    """The driver's smoke entry is `__graft_entry__.smoke()`, which calls build() — i.e. loads libtokenhmr_hip.so — BEFORE it
    imports torch.  The library needs libamdhip64.so.7, PyTorch's libraries need "libamdhip64.so" and bring their own copy: loaded in
    that order the process ends up with two HIP runtimes and thmr_create fails with "hipSetDevice failed" on a GPU box (seen in run Z
    of round 2; the CPU-side symptom is two libamdhip64 mappings).  _cabi.load() therefore imports torch first.  Checked in a fresh
    interpreter for both orders."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = "print(len(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))"
    for code in (f"import sys; sys.path.insert(0, {root!r}); import __graft_entry__ as g; g.build(); import torch; {probe}",
                 f"import sys; sys.path.insert(0, {root!r}); import torch; from tokenhmr_amd import _cabi; _cabi.load(); {probe}"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == "1", (code, r.stdout, r.stderr[-500:])


@pytest.mark.parametrize("vd,dd", [(2, 2), (32, 6)])
def test_checkpoint_contract_cpp_equals_python(built_lib, vd, dd):
    cfg = HMRConfig(vit_depth=vd, dec_depth=dd)
    cc = _cabi.Config(abi_version=_cabi.ABI_VERSION, vit_depth=vd, dec_depth=dd, max_batch=2, device=0)
    n = built_lib.thmr_spec(C.byref(cc), -1, None, None)
    cpp = {}
    for i in range(n):
        nm, ne = C.c_char_p(), C.c_int64()
        built_lib.thmr_spec(C.byref(cc), i, C.byref(nm), C.byref(ne))
        cpp[nm.value.decode()] = ne.value
    py = {name: math.prod(shape) for name, shape, *_ in W.spec(cfg) + W.tokenizer_spec(cfg)}
    assert cpp == py


def test_arena_sizes(built_lib):
    cc = _cabi.Config(abi_version=_cabi.ABI_VERSION, vit_depth=32, dec_depth=6, max_batch=64, device=0)
    wb, sb = C.c_size_t(0), C.c_size_t(0)
    assert built_lib.thmr_arena_bytes(C.byref(cc), C.byref(wb), C.byref(sb)) == 0
    n_params = sum(math.prod(s) for _, s, *_ in W.spec(RELEASE) + W.tokenizer_spec(RELEASE))
    # + SMPL constants (20 MB), conv repacks (26 MB), optional tokenizer encoder raw + repacked (53 MB)
    assert wb.value >= 4 * n_params and wb.value < 4 * n_params + 128 * 2 ** 20
    assert sb.value >= 4 * 64 * 192 * (1280 * 2 + 6144)


def test_bad_config_is_rejected(built_lib):
    wb = C.c_size_t(0)
    for kw in (dict(abi_version=99), dict(vit_depth=0), dict(dec_depth=7), dict(max_batch=0)):
        base = dict(abi_version=_cabi.ABI_VERSION, vit_depth=2, dec_depth=2, max_batch=2, device=0)
        base.update(kw)
        cc = _cabi.Config(**base)
        assert built_lib.thmr_arena_bytes(C.byref(cc), C.byref(wb), None) == -1
        assert built_lib.thmr_last_error(None)


def test_mode_bytes_and_creation_flags(built_lib):
    """ABI 5: what the engine allocates OUTSIDE the caller's arenas is a query (thmr_mode_bytes), and the mode / the
    co-residency-free kernels are creation-time choices (thmr_config.flags)."""
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    cc = _cabi.Config(abi_version=_cabi.ABI_VERSION, vit_depth=32, dec_depth=6, max_batch=64, device=0)
    assert built_lib.thmr_mode_bytes(C.byref(cc), 1, C.byref(a), C.byref(b), C.byref(c)) == 0
    per_block = 1280 * 3840 + 1280 * 1280 + 2 * 1280 * 5120
    assert a.value == (per_block * 32 + 6 * 1024 * 1280 + 1280 * 768) * 6            # 3.8 GB: ViT weights + stacked to_kv + patch embed, 6 bytes each
    assert b.value == 64 * 192 * (1280 + 5120) * 6 + 2 * 64 * 192 * 1280 * 4
    assert c.value > 256 * 128 * 256 * 4                                               # 256 slabs of one raw accumulator tile + flags
    assert built_lib.thmr_mode_bytes(C.byref(cc), 0, C.byref(a), C.byref(b), C.byref(c)) == 0 and (a.value, b.value, c.value) == (0, 0, 0)
    cc.flags = _cabi.CFG_NO_PERSISTENT
    assert built_lib.thmr_mode_bytes(C.byref(cc), 1, C.byref(a), C.byref(b), C.byref(c)) == 0 and a.value > 0 and c.value == 0
    cc.flags, cc.max_batch = 0, 2                                                      # one and two crops never reach the mode
    assert built_lib.thmr_mode_bytes(C.byref(cc), 1, C.byref(a), C.byref(b), C.byref(c)) == 0 and (a.value, b.value, c.value) == (0, 0, 0)
    assert built_lib.thmr_mode_bytes(C.byref(cc), 2, C.byref(a), None, None) == -1
    for bad in (dict(flags=4), dict(flags=-1), dict(reserved=(C.c_int32 * 2)(1, 0))):
        kw = dict(abi_version=_cabi.ABI_VERSION, vit_depth=2, dec_depth=2, max_batch=4, device=0)
        kw.update(bad)
        assert built_lib.thmr_arena_bytes(C.byref(_cabi.Config(**kw)), C.byref(a), None) == -1


def test_create_without_gpu_fails_loudly(built_lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cc = _cabi.Config(abi_version=_cabi.ABI_VERSION, vit_depth=1, dec_depth=1, max_batch=1, device=0)
    h = C.c_void_p(0)
    rc = built_lib.thmr_create(C.byref(cc), None, None, C.byref(h))
    assert rc != 0 and not h.value
    from tokenhmr_amd.engine import Engine
    with pytest.raises(Exception):
        Engine(HMRConfig(vit_depth=1, dec_depth=1), max_batch=1, device="cpu")     # no CPU fallback


def test_missing_library_message(monkeypatch):
    monkeypatch.setattr(_cabi, "_libs", {})
    monkeypatch.setattr(_cabi, "LIB_PATH", "/nonexistent/libtokenhmr_hip.so")
    monkeypatch.setattr(_cabi, "LIB_PATH_EXP", "/nonexistent/libtokenhmr_hip_exp.so")
    monkeypatch.delenv("THMR_LIB", raising=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _cabi.load()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _cabi.load(exp=True)


def test_validate_state_strict():
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    sd, tok = W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0)
    W.validate_state(sd, cfg, tok)
    bad = dict(sd)
    bad.pop("backbone.pos_embed")
    with pytest.raises(KeyError):
        W.validate_state(bad, cfg)
    bad = dict(sd)
    bad["backbone.pos_embed"] = torch.zeros(1, 10, 1280)
    with pytest.raises(ValueError):
        W.validate_state(bad, cfg)


def test_synthetic_weights_are_deterministic():
    cfg = HMRConfig(vit_depth=1, dec_depth=1)
    a, b = W.make_synthetic_state(cfg, 3), W.make_synthetic_state(cfg, 3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = W.make_synthetic_state(cfg, 4)
    assert not torch.equal(a["backbone.pos_embed"], c["backbone.pos_embed"])
    assert a["smpl_head.init_body_pose"].reshape(24, 6)[5].tolist() == [1, 0, 0, 0, 1, 0]


def test_shard_ranges_cover_batch():
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 4, 8):
            rs = [D.shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in rs]
            assert max(sizes) - min(sizes) <= 1 and sizes == D.shard_sizes(total, world)


def test_record_pack_roundtrip_is_bit_exact():
    g = torch.Generator().manual_seed(0)
    B = 3
    o = {"pred_vertices": torch.randn(B, 6890, 3, generator=g), "pred_keypoints_3d": torch.randn(B, 44, 3, generator=g),
         "pred_keypoints_2d": torch.randn(B, 44, 2, generator=g), "rotmat": torch.randn(B, 24, 3, 3, generator=g),
         "betas": torch.randn(B, 10, generator=g), "pred_cam": torch.randn(B, 3, generator=g),
         "pred_cam_t": torch.randn(B, 3, generator=g),
         "token_idx": torch.randint(0, 2048, (B, 160), generator=g, dtype=torch.int32)}
    rec = D.pack_records(o)
    assert rec.shape == (B, D.RECORD_WORDS) and D.RECORD_WORDS * 4 == 85128   # ~85.1 KB/crop
    back = D.unpack_records(rec)
    for k in o:
        assert torch.equal(back[k], o[k]), k


def test_yaml_cfg_reader(tmp_path):
    from tokenhmr_amd.model import _read_yaml_cfg
    p = tmp_path / "model_config.yaml"
    p.write_text("MODEL:\n  IMAGE_SIZE: 256\n  BACKBONE:\n    TYPE: vit\n  SMPL_HEAD:\n    TYPE: token\n"
                 "    TRANSFORMER_DECODER:\n      depth: 6\nSMPL:\n  GENDER: neutral\n")
    cfg, _ = _read_yaml_cfg(str(p))
    assert cfg.MODEL.IMAGE_SIZE == 256 and cfg.MODEL.BACKBONE.TYPE == "vit"
    assert cfg.MODEL.SMPL_HEAD.TRANSFORMER_DECODER.depth == 6
    assert cfg.MODEL.get("BBOX_SHAPE", None) is None
    # what the reference's callers do with the yacs node (lib/models/__init__.py:9-23, eval.py:119, demo.py:85): membership, mapping
    # access, assignment, defrost / freeze, dict(), and the defaults get_config merges the file into (lib/configs/__init__.py:15-62)
    assert "BBOX_SHAPE" not in cfg.MODEL and cfg["MODEL"]["IMAGE_SIZE"] == 256 and cfg.EXTRA.FOCAL_LENGTH == 5000
    assert cfg.DATASETS.CONFIG.SCALE_FACTOR == 0.3 and cfg.GENERAL.NUM_WORKERS == 4 and cfg.get("ckpt_path", None) is None
    cfg.defrost()
    cfg.ckpt_path = "x.ckpt"
    cfg.MODEL.BBOX_SHAPE = [192, 256]
    cfg.freeze()
    assert cfg.get("ckpt_path") == "x.ckpt" and cfg.MODEL.BBOX_SHAPE == [192, 256]
    assert {k.lower(): v for k, v in dict(cfg.SMPL).items()} == {"gender": "neutral"}
    with pytest.raises(AttributeError):
        cfg.MODEL.NOPE
    c2 = cfg.clone()
    c2.MODEL.IMAGE_SIZE = 1
    assert cfg.MODEL.IMAGE_SIZE == 256


def test_load_tokenhmr_missing_checkpoint(tmp_path):
    from tokenhmr_amd.model import load_tokenhmr
    p = tmp_path / "model_config.yaml"
    p.write_text("MODEL:\n  IMAGE_SIZE: 256\n  BACKBONE:\n    TYPE: vit\n  SMPL_HEAD:\n    TYPE: token\n"
                 "    TRANSFORMER_DECODER:\n      depth: 6\nSMPL:\n  GENDER: neutral\nDATASETS:\n  DATASET_DIR: x\n")
    with pytest.raises(FileNotFoundError):
        load_tokenhmr(str(tmp_path / "nope.ckpt"), str(p))
    with pytest.raises(NotImplementedError):
        load_tokenhmr("x", str(p), is_train_state=True)


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tokenhmr_amd/ may import it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tokenhmr_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_stateless_entry_points_reject_bad_arguments_without_a_gpu(built_lib):
    """Argument validation happens before any HIP call, so it is testable here: null buffers, bad epilogue ids, missing
    bias / residual, null handles.  Every failure is a negative status plus a message, never a crash."""
    L = built_lib
    null = C.c_void_p(0)
    one = C.c_void_p(16)             # any non-null address: validation fails before it is dereferenced
    assert L.thmr_op_gemm(null, 32, one, null, null, one, 32, 1, 1, 32, 0, 1.0, 0, -1, null) < 0          # null A
    assert L.thmr_op_gemm(one, 32, one, null, null, one, 32, 1, 1, 32, 99, 1.0, 0, -1, null) < 0          # bad epilogue id
    assert L.thmr_op_gemm(one, 32, one, null, null, one, 32, 1, 1, 32, 1, 1.0, 0, -1, null) < 0           # bias epilogue, no bias
    assert L.thmr_op_gemm(one, 32, one, one, null, one, 32, 1, 1, 32, 4, 1.0, 0, -1, null) < 0            # residual epilogue, no residual
    assert b"resid" in L.thmr_last_error(None)
    assert L.thmr_op_layernorm(null, one, one, one, 1, 64, 1e-5, 0, null) < 0
    assert L.thmr_op_rot6d(null, one, 1, null) < 0 and L.thmr_op_aa_to_rotmat(one, null, 1, null) < 0
    assert L.thmr_op_aa_to_rotmat(one, one, 0, null) < 0
    assert L.thmr_forward(null, one, 1, None, null) < 0 and L.thmr_vit_forward(null, one, 1, one, null) < 0
    assert L.thmr_encode_tokens(null, one, 1, one, null, null) < 0 and L.thmr_vq_decode(null, one, 1, one, null) < 0
    assert L.thmr_cropper_run(null, one, 8, 8, 24, None, 1, 256, 1, None, None, one, null) < 0
    assert b"null cropper" in L.thmr_cropper_last_error(None)
    assert L.thmr_smpl_create(None, 1, 0, None) < 0
    h = C.c_void_p(0)
    assert L.thmr_cropper_create(-1, C.byref(h)) < 0 and not h.value


def test_gemm_k_loops_carry_no_valu_instruction(built_lib, tmp_path):
    """ISA-level regression guard (CPU): on gfx950 every VALU instruction issued between fp32 MFMAs costs matrix-pipe time
    (profiles/r1_mfma_valu_microbench.log), so the K loops of the product GEMM (128x160 tile, fc1 epilogue) and of the
    small-batch ring kernel must consist of MFMA / LDS / LDS-DMA / SALU instructions only, with saddr-form copies."""
    import glob
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    texts = {}
    for name in ("gemm_f32.o", "gemm_split16.o"):
        obj = os.path.join(os.path.dirname(_cabi.LIB_PATH), name)
        if not (os.path.exists(objdump) and os.path.exists(obj)):
            pytest.skip("llvm-objdump or the GEMM object file is not available")
        work = str(tmp_path / name)
        shutil.copy(obj, work)
        subprocess.run([objdump, "-d", "--offloading", work], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path), check=False)
        cos = glob.glob(work + "*gfx950*")
        assert cos, "no gfx950 code object inside " + name
        texts[name] = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True, check=True).stdout.split("\n")

    def k_loop(name, symbol_re, min_mfma):
        text = texts[name]
        start = [i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <.*" + symbol_re, l)]
        assert start, symbol_re
        # up to the next symbol (a kernel may hold more than one s_endpgm: the tail kernel's idle waves leave early)
        end = next((i for i in range(start[0] + 1, len(text)) if re.match(r"^[0-9a-f]+ <", text[i])), len(text)) - 1
        ins = []
        for l in text[start[0]:end + 1]:
            m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        loops = []
        for addr, op, args in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"(\d+)\s*$", args.split("<")[0].strip())
                if not m:
                    continue
                off = int(m.group(1))
                off -= 65536 if off >= 32768 else 0
                tgt = addr + 4 + off * 4
                if tgt < addr:
                    seg = [x for x in ins if tgt <= x[0] <= addr]
                    if sum(1 for x in seg if x[1].startswith("v_mfma")) >= min_mfma:
                        loops.append(seg)
        assert loops, "no MFMA loop found in " + symbol_re
        return min(loops, key=len)                      # the innermost loop with that many MFMAs

    # ... and the split3 GEMM (gemm_split16.hip: 16x16x32 bf16 MFMAs, 96 per K tile and wave, two K tiles per loop trip): the per-tile
    # instantiations of the 128 x 256 tile with the GELU and the residual epilogue, and the 128 x 128 tile (the persistent instantiations
    # run the same K tile body; their loop also holds the cold branch that looks up the next tile of the stream)
    for name, sym, min_mfma in (("gemm_f32.o", r"gemm_f32_kernelILi4ELi1ELi1ELi5ELb1ELi2", 160), ("gemm_f32.o", r"gemm_ring_kernelILi4ELi5ELb0", 64),
                                # template <WN, NI, EPI, PERSIST, ABLK, NST>
                                ("gemm_split16.o", r"gemm_split16_kernelILi4ELi4ELi2ELb0ELb0ELi2E", 192),
                                ("gemm_split16.o", r"gemm_split16_kernelILi4ELi4ELi4ELb0ELb0ELi2E", 192),
                                ("gemm_split16.o", r"gemm_split16_kernelILi2ELi4ELi0ELb0ELb0ELi2E", 192),
                                # round 6: the 128 x 128 tile with a three-stage K ring (six K tiles per loop trip), GELU / raw partial sums
                                ("gemm_split16.o", r"gemm_split16_kernelILi2ELi4ELi2ELb0ELb0ELi3E", 576),
                                ("gemm_split16.o", r"gemm_split16_kernelILi2ELi4ELi0ELb0ELb0ELi3E", 576),
                                # round 5: the mixed grid (8-wave body on wide tiles, 4-wave body on the half tiles of the last round): fc1's instantiation
                                ("gemm_split16.o", r"gemm_split16_tail_kernelILi2ELb0ELb0E", 192)):
        seg = k_loop(name, sym, min_mfma)
        valu = [op for _, op, _ in seg if op.startswith("v_") and not op.startswith("v_mfma")]
        dma = [(op, args) for _, op, args in seg if op.startswith("global_load_lds")]
        assert not valu, (sym, valu[:8])
        assert dma and all(re.search(r"s\[\d+:\d+\]", args) for _, args in dma), (sym, dma[:2])
        # The M0 assumption of dma16_saddr (gemm_f32.hip): the inline asm writes M0 without being able to declare it, so the
        # LDS destination of every copy is only right if NOTHING sits between its `s_mov_b32 m0` and the load.  Guard it at the
        # ISA level: each global_load_lds in the K loop is immediately preceded by `s_mov_b32 m0, sN` + `s_nop` (same basic
        # block by construction: no branch, label or other M0 writer can be in between), and M0 has no other writer in the loop.
        for i, (_, op, args) in enumerate(seg):
            if op.startswith("global_load_lds"):
                assert i >= 2 and seg[i - 1][1] == "s_nop" and seg[i - 2][1] == "s_mov_b32" and seg[i - 2][2].startswith("m0,"), \
                    (sym, seg[max(0, i - 3):i + 1])
        m0_writers = [x for x in seg if re.match(r"m0\b", x[2]) and x[1].startswith("s_")]
        assert len(m0_writers) == len(dma), (sym, len(m0_writers), len(dma))


def test_shipped_library_reads_no_environment_and_carries_no_experiment(built_lib):
    """the A/B knobs (22 getenv("THMR_*") in round 3), the debug hooks and the kernels that lost their A/B live in the
    EXPERIMENTS build only (-DTHMR_EXPERIMENTS -> lib/libtokenhmr_hip_exp.so).  The shipped library imports no getenv, holds no knob
    name, none of those kernels' symbols, and no source file calls getenv outside the one guarded helper in common.h."""
    import glob
    import re
    shipped = open(_cabi.LIB_PATH, "rb").read()
    exp = open(_cabi.LIB_PATH_EXP, "rb").read()
    assert b"getenv" not in shipped and b"getenv" in exp
    knobs = [b"THMR_LEGACY_HEAD", b"THMR_DEC_FORCE_TIMEOUT", b"THMR_DEC_BARRIER", b"THMR_SPLIT3_SMALL", b"THMR_ALONE_PENALTY", b"THMR_SPLIT3_TILE",
             b"THMR_ATTN_VARIANT", b"THMR_MID_SPLIT", b"THMR_SPLIT3_PERSIST"]
    for k in knobs:
        assert k not in shipped and k in exp, k
    for sym in (b"gemm_split3_wide_kernel", b"gemm_split3_ring_kernel", b"gemm_split3_persist_kernel", b"18gemm_split3_kernel"):
        assert sym not in shipped and sym in exp, sym          # incl. the 32x32x16 split3 kernels the 16x16x32 ones replaced
    assert b"gemm_split16_kernel" in shipped
    csrc = os.path.join(os.path.dirname(os.path.dirname(_cabi.LIB_PATH)), "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        n = len(re.findall(r"\bgetenv\s*\(", open(f).read()))
        assert n == (1 if f.endswith("common.h") else 0), (f, n)
    assert b"experiments" in _cabi.load(exp=True).thmr_build_info() and b"experiments" not in _cabi.load(exp=False).thmr_build_info()


def test_attention_lds_dma_copies_keep_their_m0(built_lib, tmp_path):
    """The attention kernel issues its K / V copies through the same inline-assembly saddr LDS-DMA as the GEMM (attention.hip
    dma16_saddr): M0 is written inside the asm and cannot be declared, so — on every toolchain bump — check at the ISA level that each
    global_load_lds of every attention instantiation is immediately preceded by its own `s_mov_b32 m0` + `s_nop`, that the copies are
    the saddr form (uniform 64-bit base in SGPRs), and that nothing else in the kernel writes M0."""
    import glob
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(os.path.dirname(_cabi.LIB_PATH), "attention.o")
    if not (os.path.exists(objdump) and os.path.exists(obj)):
        pytest.skip("llvm-objdump or the attention object file is not available")
    work = str(tmp_path / "attention.o")
    shutil.copy(obj, work)
    subprocess.run([objdump, "-d", "--offloading", work], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path), check=False)
    cos = glob.glob(work + "*gfx950*")
    assert cos, "no gfx950 code object inside attention.o"
    text = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True, check=True).stdout.split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <.*vit_attention_(persistent_)?kernel", l)]
    assert len(starts) >= 3 and any("persistent" in text[i] for i in starts)
    assert sum(1 for i in starts if "persistent" in text[i]) == 2          # the fp32-output and the split3-output instantiation
    for st in starts:
        end = next(i for i in range(st, len(text)) if "s_endpgm" in text[i])
        ins = []
        for l in text[st:end + 1]:
            m = re.match(r"\s+(\S+)\s+(.*?)\s*//", l)
            if m:
                ins.append((m.group(1), m.group(2)))
        dma = [i for i, (op, _) in enumerate(ins) if op.startswith("global_load_lds")]
        assert len(dma) >= 8, (text[st], len(dma))
        for i in dma:
            assert ins[i - 1][0] == "s_nop" and ins[i - 2][0] == "s_mov_b32" and ins[i - 2][1].startswith("m0,"), ins[i - 3:i + 1]
            assert re.search(r"s\[\d+:\d+\]", ins[i][1]), ins[i]
        assert sum(1 for op, a in ins if op.startswith("s_") and re.match(r"m0\b", a)) == len(dma)
        if "persistent" in text[st]:
            # The persistent kernel counts VMEM completion by hand across compiler-visible operations: before the next item starts it
            # waits with vmcnt(30) for the K' copies, i.e. it RELIES on exactly 15 output stores + 15 Q' loads being the only younger
            # operations of the wave.  Fewer (a toolchain that merges stores, or spill traffic moved elsewhere) would make the wait too
            # weak: check the instruction stream between the last copy and that wait, and that the kernel has no scratch traffic.
            # The SPLIT instantiation (output as a split3 operand: after the permlane16 swaps 7 tile pairs x three 16-byte stores + one
            # unpaired tile x three 8-byte stores) waits with vmcnt(39) for 24 stores + 15 loads.
            split = "Lb1E" in text[st]
            n_wait, store_ops = (39, {"global_store_dwordx4": 21, "global_store_dwordx2": 3}) if split else (30, {"global_store_dwordx4": 15})
            w30 = [i for i, (op, a) in enumerate(ins) if op == "s_waitcnt" and a.replace(" ", "") == f"vmcnt({n_wait})"]
            assert len(w30) == 1, [a for op, a in ins if op == "s_waitcnt" and "vmcnt" in a]
            last_dma = max(i for i in dma if i < w30[0])
            between = [op for op, _ in ins[last_dma + 1:w30[0]]]
            assert all(between.count(op) == n for op, n in store_ops.items()) and between.count("global_load_dwordx4") == 15, between
            assert not any(op.startswith(("global_", "buffer_", "scratch_", "flat_")) and op not in (*store_ops, "global_load_dwordx4")
                           for op in between), between
            if split:
                assert sum(1 for op, _ in ins if op.startswith("v_permlane16_swap")) >= 28
            assert not any(op.startswith("scratch_") for op, _ in ins), "register spills in the persistent attention kernel"


# gfx950 serves a wave's ds_read_b128 in four groups of 16 lanes, one LDS cycle each when the 16 lanes' 16-byte reads fall into 16 different
# bank slots (64 banks x 4 bytes = 16 slots of 16 bytes); the groups are NOT consecutive lanes (MI355X_MICROARCH.md, LDS section)
_B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
_B128_GROUPS += [[l + 32 for l in g] for g in _B128_GROUPS]


def _b128_conflict_cycles(byte_addr_of_lane):
    """extra LDS cycles of one ds_read_b128 wave instruction (0 = conflict-free) for a lane -> byte address map"""
    extra = 0
    for grp in _B128_GROUPS:
        slots = {}
        for lane in grp:
            a = byte_addr_of_lane(lane)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)          # identical addresses broadcast
        extra += max(len(v) for v in slots.values()) - 1
    return extra


def test_split16_stage_image_reads_are_conflict_free_under_gfx950_lane_groups():
    """csrc/gemm_split16.hip: lane (l15, g) reads chunk 3 g + pc of row l15 of a 16-row tile from a stage image with rows of 192 bytes whose 12
    chunks are rotated by rot16(row).  The rotation in the source must make every fragment read conflict-free under the hardware's real lane
    groups (PMC on the GPU: SQ_LDS_BANK_CONFLICT 60.3 M -> 0 per fc1 launch, profiles/r4ai_pmc_lds_after_swizzle_fix.json); the first
    version's rotation (row >> 2) & 3 — right for 16 consecutive lanes — costs one extra cycle in every group (the model reproduces the
    measured 8 instead of 4 cycles per read)."""
    import re
    src = open(os.path.join(os.path.dirname(_cabi.__file__), "csrc", "gemm_split16.hip")).read()
    m = re.search(r"constexpr int rot16\(int row\) \{ return (.+?); \}", src)
    assert m, "rot16 not found in gemm_split16.hip"
    assert m.group(1).replace(" ", "") == "((row>>2)&1)*2", f"rot16 changed ({m.group(1)}): restate it below and re-run the model"

    def rot_src(row):                       # the source's rot16, restated (checked textually above)
        return ((row >> 2) & 1) * 2

    def cycles(rot, pc, tile_row0):
        def addr(lane):
            l15, g = lane & 15, lane >> 4
            row = tile_row0 + l15
            return row * 192 + ((3 * g + pc + rot(row)) % 12) * 16
        return _b128_conflict_cycles(addr)

    for pc in range(3):
        for tile_row0 in (0, 16, 48, 112):
            assert cycles(rot_src, pc, tile_row0) == 0, (pc, tile_row0)
            assert cycles(lambda row: (row >> 2) & 3, pc, tile_row0) == 4, (pc, tile_row0)      # one extra cycle in each of the four groups
    # the row-blocked A image ([rows / 32][k-group x piece][32 rows][16 bytes]) needs no rotation
    for pc in range(3):
        assert _b128_conflict_cycles(lambda lane: (3 * (lane >> 4) + pc) * 512 + (lane & 15) * 16) == 0


def _b64_write_conflict_cycles(byte_addr_of_lane):
    """extra LDS-array cycles of one ds_write_b64 wave instruction: four groups of 16 CONSECUTIVE lanes, bank = (addr / 4) mod 32, two banks
    per lane (MI355X_MICROARCH.md, LDS)"""
    extra = 0
    for g in range(4):
        banks = {}
        for lane in range(16 * g, 16 * g + 16):
            a = byte_addr_of_lane(lane)
            assert a % 8 == 0
            for b in ((a // 4) % 32, (a // 4 + 1) % 32):
                banks.setdefault(b, set()).add(a)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def test_attention_b16_images_are_conflict_free_under_gfx950_lane_groups():
    """csrc/attention_b16.hip stages K as [piece][key][96 d] and V^T as [piece][d][64 key slots] and reads both with ds_read_b128 (K: row = key
    l15, chunk 4 s + g; V^T: row = d l15, chunk 4 st + g), writes both with ds_write_b64.  Round 4's strides (208 / 144 bytes) cost every
    fragment read one extra cycle per lane group (PMC: 5.8 M conflict cycles of 12.0 M, profiles/r4ah_pmc_lds.json).  Round 5: K rows of 224
    bytes, V^T rows of 160 bytes with chunk ^= (row >> 2) & 1 and the staging threads re-mapped; this restates the kernel's address maps (the
    constants are read from the source) and asserts that EVERY read and write instruction is conflict-free in the lane-group model — and that
    the round-4 layout reproduces its measured conflicts, so the model is not vacuous."""
    import re
    src = open(os.path.join(os.path.dirname(_cabi.__file__), "csrc", "attention_b16.hip")).read()
    krs = int(re.search(r"constexpr int KRS = (\d+);", src).group(1))
    vrs = int(re.search(r"constexpr int VRS = (\d+);", src).group(1))
    assert (krs, vrs) == (224, 160), "strides changed: restate the maps below"
    # the source's maps, textually (restated below)
    assert "vdg = (tid & 7) + 8 * ((tid >> 4) & 1), vkq = ((tid >> 5) & 3) | (((tid >> 3) & 1) << 2) | ((tid >> 7) << 3)" in src
    assert "vimg + vdg * VRS + (((4 * (vkq >> 3) + (vkq & 3)) ^ ((vdg >> 2) & 1)) * 16) + 8 * ((vkq >> 2) & 1)" in src
    assert "vimg + l15 * VRS + ((g ^ ((l15 >> 2) & 1)) * 16)" in src
    assert "kimg + kkey * KRS + kpart * 8" in src and "kimg + l15 * KRS + g * 16" in src

    def k_read(stride, kt, s):
        return _b128_conflict_cycles(lambda lane: (16 * kt + (lane & 15)) * stride + (4 * s + (lane >> 4)) * 16)

    def v_read(stride, dt, st, swz):
        return _b128_conflict_cycles(lambda lane: (16 * dt + (lane & 15)) * stride + (((4 * st + (lane >> 4)) ^ ((((lane & 15) >> 2) & 1) if swz else 0)) * 16))

    def k_write(stride, wave, m):
        def addr(lane):
            tid = wave * 64 + lane
            return (tid >> 2) * stride + (tid & 3) * 8 + m * 32
        return _b64_write_conflict_cycles(addr)

    def v_write(stride, wave, m, new):
        def addr(lane):
            tid = wave * 64 + lane
            if new:
                dg, kq = (tid & 7) + 8 * ((tid >> 4) & 1), ((tid >> 5) & 3) | (((tid >> 3) & 1) << 2) | ((tid >> 7) << 3)
                return (dg + 16 * m) * stride + (((4 * (kq >> 3) + (kq & 3)) ^ ((dg >> 2) & 1)) * 16) + 8 * ((kq >> 2) & 1)
            kq, dg = tid >> 4, tid & 15
            return (dg + 16 * m) * stride + 2 * (32 * (kq >> 3) + 8 * (kq & 3) + 4 * ((kq >> 2) & 1))
        return _b64_write_conflict_cycles(addr)

    # the new thread map is a bijection onto (16 key quads) x (16 d) and every (key slot, d) element has one home
    pairs = {((t & 7) + 8 * ((t >> 4) & 1), ((t >> 5) & 3) | (((t >> 3) & 1) << 2) | ((t >> 7) << 3)) for t in range(256)}
    assert len(pairs) == 256
    homes = set()
    for dg, kq in pairs:
        for m in range(5):
            homes.add((dg + 16 * m) * 160 + (((4 * (kq >> 3) + (kq & 3)) ^ ((dg >> 2) & 1)) * 16) + 8 * ((kq >> 2) & 1))
    assert len(homes) == 256 * 5 and max(homes) < 80 * 160
    new = dict(k_read=sum(k_read(224, kt, s) for kt in range(4) for s in range(3)), v_read=sum(v_read(160, dt, st, True) for dt in range(5) for st in range(2)),
               k_write=sum(k_write(224, w, m) for w in range(4) for m in range(5)), v_write=sum(v_write(160, w, m, True) for w in range(4) for m in range(5)))
    assert all(v == 0 for v in new.values()), new
    old = dict(k_read=sum(k_read(208, kt, s) for kt in range(4) for s in range(3)), v_read=sum(v_read(144, dt, st, False) for dt in range(5) for st in range(2)),
               k_write=sum(k_write(208, w, m) for w in range(4) for m in range(5)), v_write=sum(v_write(144, w, m, False) for w in range(4) for m in range(5)))
    assert old["k_read"] == 48 and old["v_read"] == 40 and old["k_write"] > 0 and old["v_write"] > 0, old      # one extra cycle in each group of each read
    # the padded V^T stride alone (HISTORY.md 11.13) fixes the reads but would have made round 4's writes 4-way: hence the swizzle + thread map
    assert sum(v_read(160, dt, st, False) for dt in range(5) for st in range(2)) == 0
    assert sum(v_write(160, w, m, False) for w in range(4) for m in range(5)) > old["v_write"]
    assert 2 * (3 * 64 * 224 + 3 * 80 * 160) <= 160 * 1024                      # two workgroups per CU


def test_batch_metrics_is_a_faithful_read_only_mapping():
    """the per-batch result of `Evaluator.__call__`.  Round 4 returned a dict subclass with placeholder values (`dict(res)`,
    `{**res}`, `==` bypassed its `__getitem__`); round 5's Mapping read the EVALUATOR's arrays at first access (so a read after
    `merge_evaluator` returned the merged arrays' slice) and kept the evaluator alive.  Now: a Mapping over the batch's OWN (3, B) result,
    every access a fresh copy (pose_utils.py:246), `to_dict()` = the reference's plain dict, `Evaluator(eager_results=True)` returns it."""
    import numpy as np
    from tokenhmr_amd.evaluator import _BatchMetrics
    own = torch.tensor([[2.0, 3.0, 4.0], [4.0, 6.0, 8.0], [0.0, 0.0, 0.0]])
    res = _BatchMetrics(own.clone(), ["mode_mpjpe", "mode_re"])
    other = _BatchMetrics(own.clone() + 3.0, ["mode_mpjpe", "mode_re"])
    assert list(res) == ["mode_mpjpe", "mode_re"] and len(res) == 2 and "mode_re" in res and "mode_pve" not in res
    np.testing.assert_array_equal(res["mode_mpjpe"], [2.0, 3.0, 4.0])
    assert res["mode_mpjpe"].dtype == np.float64
    d = dict(res)
    assert set(d) == {"mode_mpjpe", "mode_re"} and d["mode_re"] is not None
    np.testing.assert_array_equal({**res}["mode_re"], [4.0, 6.0, 8.0])
    np.testing.assert_array_equal(res.get("mode_re"), [4.0, 6.0, 8.0])
    assert res.get("mode_pve") is None
    assert not np.array_equal(dict(res)["mode_mpjpe"], dict(other)["mode_mpjpe"])      # two batches are not "equal"
    res["mode_mpjpe"][0] = -1.0                                # a caller's write does not reach the stored values
    assert res["mode_mpjpe"][0] == 2.0
    assert res._dev3 is None and not hasattr(res, "_ev")       # materialised once; nothing of the evaluator is kept alive
    plain = res.to_dict()
    assert type(plain) is dict and set(plain) == {"mode_mpjpe", "mode_re"}
    plain["mode_mpjpe"] = np.zeros(3)                          # the plain dict is the caller's
    assert res["mode_mpjpe"][0] == 2.0
    with pytest.raises(KeyError):
        res["nope"]
    with pytest.raises(TypeError):
        res["mode_re"] = np.zeros(3)                           # the Mapping is read-only


def test_tile_stream_decomposition_model():
    """The persistent split3 GEMMs' decomposition (csrc/gemm_split16.hip split16_body, PERSIST: 8 XCDs x 32 lanes; lane `ln` owns the units
    j * 32 + ln, its list of T units x nk K tiles is cut into 8 contiguous ranges, one per XCD; a range boundary inside a unit = the first
    part is stored to the slab by XCD x's workgroup (its FIRST segment) and continued by XCD x + 1's (its LAST segment)) restated in Python
    and checked over the shapes the engine streams in round 6 — 128 x 256 tiles, 128 x 128 tiles, (tile, K slice) units of split-K launches,
    ragged M: every (unit, K tile) is multiplied exactly once, a unit is shared by at most two workgroups, each workgroup has at most one
    producer and one consumer segment, the consumer's range continues its producer's, and no range is shorter than one unit (the
    launcher's >= 256 units), so a producer's part precedes its consumer's need."""
    def ranges(units, nk):
        out = {}
        for ln in range(32):
            T = (units - ln + 31) // 32
            for x in range(8):
                S0, S1 = x * T * nk // 8, (x + 1) * T * nk // 8
                j0, k0 = divmod(S0, nk)
                j1 = (S1 - 1) // nk
                k1 = S1 - j1 * nk
                has_pre, has_post = int(k1 < nk), int(k0 > 0)
                jf0 = j0 + has_post
                nfull = max(j1 - has_pre - jf0 + 1, 0)
                segs = []
                if has_pre:
                    segs.append((j1, 0, k1, 1))
                segs += [(jf0 + m, 0, nk, 0) for m in range(nfull)]
                if has_post:
                    segs.append((j0, k0, nk, 2))
                out[(x, ln)] = segs
        return out

    cases = []
    for crops, N, K, ks, bn in ((64, 1280, 5120, 1, 256), (35, 1280, 5120, 1, 256), (24, 3840, 1280, 1, 256), (14, 3840, 1280, 1, 256), (18, 5120, 1280, 1, 256),
                                (6, 3840, 1280, 1, 128), (12, 3840, 1280, 1, 128), (10, 5120, 1280, 1, 128), (5, 5120, 1280, 1, 128), (13, 3840, 1280, 1, 128),
                                (9, 1280, 5120, 2, 128), (18, 1280, 5120, 2, 128), (20, 1280, 5120, 2, 128), (10, 1280, 1280, 2, 128), (6, 1280, 5120, 4, 128)):
        rows = (192 * crops + 127) // 128
        cases.append((rows * (N // bn) * ks, K // 32 // ks))
    cases += [(256, 11), (257, 3), (544, 3), (1000, 80)]
    for units, nk in cases:
        assert units >= 256
        segs = ranges(units, nk)
        seen = {}
        for (x, ln), ss in segs.items():
            assert sum(1 for s_ in ss if s_[3] == 1) <= 1 and sum(1 for s_ in ss if s_[3] == 2) <= 1
            assert not ss or ss[0][3] != 2 or len(ss) == 1                       # a consumer segment is the workgroup's last
            assert all(s_[3] != 1 for s_ in ss[1:])                              # a producer segment is its first
            assert sum(ke - kb for _, kb, ke, _ in ss) >= nk, (units, nk, x, ln)  # no range shorter than one unit
            for j, kb, ke, kind in ss:
                u = j * 32 + ln
                assert 0 <= u < units and 0 <= kb < ke <= nk
                for k in range(kb, ke):
                    assert (u, k) not in seen, (units, nk, u, k)
                    seen[(u, k)] = (x, kind)
                if kind == 2:                                                    # continues what XCD x - 1's workgroup of this lane stored
                    prod = [s_ for s_ in segs[(x - 1, ln)] if s_[3] == 1]
                    assert prod and prod[0][0] == j and prod[0][2] == kb and prod[0][1] == 0
        assert len(seen) == units * nk, (units, nk, len(seen))
        per_unit = {}
        for (u, k), (x, kind) in seen.items():
            per_unit.setdefault(u, set()).add(x)
        assert max(len(v) for v in per_unit.values()) <= 2


def test_split3_handover_epoch_protocol_model():
    """A model of the persistent split3 GEMM's slab hand-over (csrc/gemm_split16.hip: flag[xcd][lane] holds the EPOCH of the launch that
    published the slab; every workgroup reads the workspace's epoch word at its start; a consumer accepts a slab iff its flag equals this
    launch's epoch; the last of the launch's arrivals advances the epoch word) against the failure ADVICE r4 described for round 4's 0 / 1
    flags: a producer that publishes AFTER its consumer's bounded wait ran out must not be able to feed a LATER launch a stale slab.
    Launches on a stream are serialised by the kernel boundary and the (producer, consumer) pairs of a launch are independent, so a launch
    is, per pair, either publish -> consume (on time) or timed-out consume -> late publish."""
    import random

    def run(protocol, seed):
        rng = random.Random(seed)
        lanes = 8
        flag = [0] * lanes
        slab = [None] * lanes               # (launch, lane) that wrote it
        epoch_word = 0
        stale_accepts = 0
        for launch in range(1, 40):
            ep = epoch_word + 1             # read once by every workgroup at its start
            want = ep if protocol == "epoch" else 1

            def publish(l):
                slab[l] = (launch, l)
                flag[l] = want

            def consume(l, timed_out):
                nonlocal stale_accepts
                if flag[l] == want:         # the spin loop's exit condition
                    if slab[l] != (launch, l):
                        stale_accepts += 1
                    if protocol == "flags01":
                        flag[l] = 0         # round 4: the consumer re-arms the flag
                else:
                    assert timed_out        # an on-time pair always finds its flag
                    if protocol == "flags01":
                        flag[l] = 0         # round 4's time-out path stored 0 — and the late producer's 1 then survives the launch

            for l in range(lanes):
                if rng.random() < 0.2:      # this producer misses its consumer's bounded wait
                    consume(l, True)
                    publish(l)
                else:
                    publish(l)
                    consume(l, False)
            epoch_word = ep                 # the last arrival closes the epoch
        return stale_accepts

    assert sum(run("flags01", s) for s in range(20)) > 0          # the model reproduces round 4's hazard ...
    assert all(run("epoch", s) == 0 for s in range(200))          # ... and an epoch flag cannot admit a slab of another launch
