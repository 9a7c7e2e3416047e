"""ctypes binding of include/tokenhmr_hip.h (libtokenhmr_hip.so).

There is NO CPU fallback: if the shared library is missing or a symbol is absent this module
raises, so a GPU box can never silently run anything but the HIP path.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtokenhmr_hip.so")
# the EXPERIMENTS build of the same sources (-DTHMR_EXPERIMENTS: THMR_* environment knobs, debug hooks, kernels that lost their A/B —
# csrc/common.h).  Never the product path: tests and scripts ask for it with load(exp=True) / Engine(..., experiments=True), or a whole
# process with THMR_LIB=exp.
LIB_PATH_EXP = os.path.join(_HERE, "lib", "libtokenhmr_hip_exp.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tokenhmr_hip.h")

ABI_VERSION = 5
# thmr_config.flags (header: THMR_CFG_*)
CFG_VIT_GEMM_F32, CFG_NO_PERSISTENT = 1, 2
PROF_NAMES = ["gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "attention", "layernorm", "patch_embed",
              "dec_kv", "head", "lbs"]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("vit_depth", C.c_int32), ("dec_depth", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 2)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64),
                ("on_device", C.c_int32), ("reserved", C.c_int32)]


class SmplDesc(C.Structure):
    _fields_ = [("v_template", C.c_void_p), ("shapedirs", C.c_void_p), ("posedirs", C.c_void_p),
                ("J_regressor", C.c_void_p), ("lbs_weights", C.c_void_p), ("J19_regressor", C.c_void_p),
                ("parents", C.c_void_p), ("extra_verts", C.c_void_p), ("joint_map", C.c_void_p),
                ("on_device", C.c_int32), ("update_hips", C.c_int32)]


OUTPUT_FIELDS = ["pred_cam", "rotmat", "betas", "cls_logits_softmax", "pred_cam_t", "focal_length",
                 "pred_keypoints_3d", "pred_vertices", "pred_keypoints_2d", "token_idx",
                 "vit_features", "token_out", "cls_logits", "pose6d"]


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUTPUT_FIELDS]


class CropDesc(C.Structure):
    _fields_ = [("M", C.c_double * 6), ("sigma", C.c_double), ("truncate", C.c_double)]


class ProfEntry(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("launches", C.c_int64)]


def declared_symbols():
    """Every function the header declares (used by the symbol-export test)."""
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(thmr_[a-z0-9_]+)\s*\(", text)))


_libs = {}


def load(exp=None):
    """The shipped library, or (exp=True, or exp=None with THMR_LIB=exp in the environment) the experiments build.
    exp = a PATH (str): that very file — another BUILD of this library, e.g. the previous round's, loaded beside the current one by
    scripts/ab_same_box.py so that two builds are timed interleaved in one process on one box (A/B tooling only; ABI 3 and 4 builds
    accepted: no struct layout or signature changed between 3 and 5 — 4 changed the creation default of the ViT GEMM mode, 5 gave
    thmr_config.reserved[0] a meaning and added thmr_mode_bytes, which such a build simply lacks)."""
    if exp is None:
        exp = os.environ.get("THMR_LIB", "") == "exp"
    if isinstance(exp, str):
        path = os.path.abspath(exp)
        exp = path
    else:
        exp = bool(exp)
        path = LIB_PATH_EXP if exp else LIB_PATH
    if exp in _libs:
        return _libs[exp]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # ONE HIP runtime per process.  libtokenhmr_hip.so needs "libamdhip64.so.7" (resolved to /opt/rocm when nothing of that soname
    # is loaded yet); PyTorch's libraries need "libamdhip64.so" and find the copy bundled under torch/lib, which the loader does NOT
    # recognise as the same library when /opt/rocm's is already mapped under the other name — two runtimes, and the second one to
    # touch the device fails (seen as "hipSetDevice failed" in thmr_create when this module was loaded before `import torch`).
    # With torch imported first its runtime carries the soname libamdhip64.so.7 and is reused here.  Hosts without PyTorch
    # (the plain C ABI) are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    older = isinstance(exp, str) and lib.thmr_abi_version() in (3, 4)          # a previous round's build, loaded by path (A/B tooling)
    missing = [s for s in declared_symbols() if not hasattr(lib, s) and not (older and s in ("thmr_mode_bytes",))]
    if missing:
        raise RuntimeError(f"libtokenhmr_hip.so lacks symbols declared in tokenhmr_hip.h: {missing}")
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    lib.thmr_abi_version.restype = C.c_int
    lib.thmr_build_info.restype = C.c_char_p
    lib.thmr_last_error.restype = C.c_char_p
    lib.thmr_last_error.argtypes = [vp]
    lib.thmr_arena_bytes.argtypes = [C.POINTER(Config), C.POINTER(sz), C.POINTER(sz)]
    lib.thmr_spec.argtypes = [C.POINTER(Config), i32, C.POINTER(C.c_char_p), C.POINTER(i64)]
    if hasattr(lib, "thmr_mode_bytes"):
        lib.thmr_mode_bytes.argtypes = [C.POINTER(Config), i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    lib.thmr_create.argtypes = [C.POINTER(Config), vp, vp, C.POINTER(vp)]
    lib.thmr_destroy.argtypes = [vp]
    lib.thmr_destroy.restype = None
    lib.thmr_load_weights.argtypes = [vp, C.POINTER(TensorDesc), sz, vp]
    lib.thmr_load_smpl.argtypes = [vp, C.POINTER(SmplDesc), vp]
    lib.thmr_finalize_weights.argtypes = [vp, i32, vp]
    lib.thmr_weight_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    lib.thmr_forward.argtypes = [vp, vp, i32, C.POINTER(Outputs), vp]
    lib.thmr_engine_status.argtypes = [vp, vp]
    lib.thmr_debug_decoder_timeline.argtypes = [vp, C.POINTER(C.c_uint64), i32, vp]
    lib.thmr_vit_forward.argtypes = [vp, vp, i32, vp, vp]
    lib.thmr_head_forward.argtypes = [vp, vp, i32, C.POINTER(Outputs), vp]
    lib.thmr_lbs_forward.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.thmr_vq_argmin.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.thmr_encode_tokens.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.thmr_vq_decode.argtypes = [vp, vp, i32, vp, vp]
    lib.thmr_op_gemm.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp]
    lib.thmr_op_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, vp]
    lib.thmr_op_vit_attention.argtypes = [vp, vp, i32, vp]
    lib.thmr_op_vit_attention_variant.argtypes = [vp, vp, i32, i32, vp]
    lib.thmr_op_vit_attention_split3.argtypes = [vp, vp, i32, vp]
    lib.thmr_op_vit_attention_b16.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.thmr_op_split3.argtypes = [vp, i64, vp, i64, i64, i32, vp]
    lib.thmr_op_gemm_split3.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp]
    lib.thmr_op_gemm_split3_out_split3.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp]
    lib.thmr_op_rot6d.argtypes = [vp, vp, i32, vp]
    lib.thmr_op_aa_to_rotmat.argtypes = [vp, vp, i32, vp]
    lib.thmr_smpl_create.argtypes = [C.POINTER(SmplDesc), i32, i32, C.POINTER(vp)]
    lib.thmr_smpl_destroy.argtypes = [vp]
    lib.thmr_smpl_destroy.restype = None
    lib.thmr_smpl_forward.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    lib.thmr_eval_pose.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.thmr_regress_joints.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.thmr_cropper_create.argtypes = [i32, C.POINTER(vp)]
    lib.thmr_cropper_destroy.argtypes = [vp]
    lib.thmr_cropper_destroy.restype = None
    lib.thmr_cropper_last_error.argtypes = [vp]
    lib.thmr_cropper_last_error.restype = C.c_char_p
    lib.thmr_cropper_run.argtypes = [vp, vp, i32, i32, i64, C.POINTER(CropDesc), i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp]
    lib.thmr_pack_records.argtypes = [C.POINTER(Outputs), i32, vp, vp]
    lib.thmr_bcast_weights.argtypes = [vp, vp, i32, vp]
    lib.thmr_allgather_records.argtypes = [vp, vp, i32, vp, vp]
    lib.thmr_collective_last_error.restype = C.c_char_p
    lib.thmr_prof_enable.argtypes = [vp, i32]
    lib.thmr_set_vit_gemm.argtypes = [vp, i32, vp]
    lib.thmr_get_vit_gemm.argtypes = [vp]
    lib.thmr_prof_collect.argtypes = [vp, C.POINTER(ProfEntry), i32]
    for name in declared_symbols():
        if not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        if name not in ("thmr_build_info", "thmr_last_error", "thmr_destroy", "thmr_smpl_destroy", "thmr_cropper_destroy",
                        "thmr_cropper_last_error", "thmr_collective_last_error"):
            fn.restype = C.c_int
    if lib.thmr_abi_version() != ABI_VERSION and not older:
        raise RuntimeError("libtokenhmr_hip.so ABI version mismatch")
    _libs[exp] = lib
    return lib


class EngineError(RuntimeError):
    pass


def check(rc, engine=None, lib=None):
    if rc != 0:
        lib = lib if lib is not None else load()
        msg = lib.thmr_last_error(engine)
        raise EngineError(f"tokenhmr_hip error {rc}: {msg.decode() if msg else '?'}")
