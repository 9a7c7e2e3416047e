"""Interleaved A/B of compile-time variants of gemm_f32.hip on the four ViT GEMM shapes at B = 64.
`python scripts/gemm_flag_ab.py build` (here, hipcc only) makes build_ab/libv_<name>.so for every entry of VARIANTS; on the
GPU box `python scripts/gemm_flag_ab.py [rounds]` times them round-robin through thmr_op_gemm (auto tile selection = what the
engine uses) and checks that every variant is bit-identical to `base`."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "build_ab")
sys.path.insert(0, ROOT)

# name -> extra hipcc flags (an entry "src=<path>" compiles that file instead of tokenhmr_amd/csrc/gemm_f32.hip, e.g. a copy of
# an older revision: git show <rev>:tokenhmr_amd/csrc/gemm_f32.hip > build_ab/gemm_f32_old.hip).
# Round-1 experiments: profiles/r1_gemm_setprio_ntstore_experiment.log (-DTHMR_SETPRIO=1/2, -DTHMR_NT_STORE: no effect, macros
# removed again) and profiles/r1_gemm_zero_valu_loop.log (old = builtin DMA with VALU address updates, new = saddr DMA).
VARIANTS = {
    "base": [],
    # "gelu2": ["-DTHMR_GELU_IMPL=2"],      # the two-piece erf_fast epilogue (previous default): profiles/r1_gelu_single_piece_ab.log
    # "vgprform": ["-mllvm", "-amdgpu-mfma-vgpr-form"],   # accumulators in VGPRs: no effect (profiles/r1_gemm_vgpr_form_experiment.log)
}


def build():
    import __graft_entry__ as G
    G.build()
    os.makedirs(AB, exist_ok=True)
    others = [os.path.join(G.LIBDIR, s.replace(".hip", ".o")) for s in G.SOURCES if s != "gemm_f32.hip"]
    procs = []
    for name, flags in VARIANTS.items():
        o = os.path.join(AB, f"gemm_f32_{name}.o")
        src = os.path.join(G.CSRC, "gemm_f32.hip")
        for f in flags:
            if f.startswith("src="):
                src = os.path.join(ROOT, f[4:])
        if not os.path.exists(src):
            print("skipping", name, "(no", src, ")")
            continue
        flags = [f for f in flags if not f.startswith("src=")]
        procs.append((name, o, subprocess.Popen([G._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                                 "-I", G.CSRC] + flags + ["-c", src, "-o", o])))
    for name, o, p in procs:
        assert p.wait() == 0
        subprocess.check_call([G._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(AB, f"libv_{name}.so"), o] + others)
    print("built", sorted(os.listdir(AB)))


def run(rounds):
    import torch
    dev = torch.device("cuda:0")
    libs = {}
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
    for name in VARIANTS:
        if not os.path.exists(os.path.join(AB, f"libv_{name}.so")):
            continue
        lib = C.CDLL(os.path.join(AB, f"libv_{name}.so"))
        lib.thmr_op_gemm.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp]
        libs[name] = lib
    M = 64 * 192
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, 5120, generator=g).to(dev)
    w = (torch.randn(5120, 5120, generator=g) / 70).to(dev)
    b = torch.randn(5120, generator=g).to(dev)
    r = torch.randn(M, 1280, generator=g).to(dev)
    out = {k: torch.empty(M, 5120, device=dev) for k in libs}
    s = torch.cuda.current_stream().cuda_stream
    # name: (N, K, epilogue)   1 = bias, 2 = bias + GELU, 3 = bias + residual
    shapes = {"qkv": (3840, 1280, 1), "proj": (1280, 1280, 3), "fc1": (5120, 1280, 2), "fc2": (1280, 5120, 3)}

    def call(k, nm):
        n, kk, epi = shapes[nm]
        rc = libs[k].thmr_op_gemm(a.data_ptr(), 5120, w.data_ptr(), b.data_ptr(), r.data_ptr() if epi == 3 else None,
                                  out[k].data_ptr(), n, M, n, kk, epi, 1.0, 0, 8, s)
        assert rc == 0

    for nm, (n, kk, _) in shapes.items():
        for k in libs:
            out[k].zero_()
            call(k, nm)
        torch.cuda.synchronize()
        print(nm, {k: bool(torch.equal(out[k][:, :n], out["base"][:, :n])) for k in libs}, "(bitwise == base)")
    times = {(nm, k): [] for nm in shapes for k in libs}
    order = list(libs)
    for rd in range(rounds):
        for nm in shapes:
            for k in order[rd % len(order):] + order[:rd % len(order)]:      # rotate who goes first
                call(k, nm)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    call(k, nm)
                e1.record()
                torch.cuda.synchronize()
                times[(nm, k)].append(e0.elapsed_time(e1) / 6)
    for nm, (n, kk, _) in shapes.items():
        for k in libs:
            t = sorted(times[(nm, k)])
            med = t[len(t) // 2]
            print(f"{nm:5s} {k:10s}: median {med * 1e3:8.1f} us  {2.0 * M * n * kk / (med * 1e-3) / 1e12:6.1f} TF   (min {t[0] * 1e3:8.1f} us)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
