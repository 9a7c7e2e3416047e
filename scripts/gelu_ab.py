"""A/B of the GELU epilogue implementations (THMR_GELU_IMPL 0 = ocml erff, 1 = branch-free scalar, 2 = packed pairs).
`python scripts/gelu_ab.py build` (here, hipcc only) makes build_ab/libgelu{0,1,2}.so; on the GPU box
`python scripts/gelu_ab.py [rounds]` times fc1 (12288x5120x1280, bias+GELU) interleaved across the three libraries and
fc2 as the no-GELU yardstick, all through thmr_op_gemm."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "build_ab")
sys.path.insert(0, ROOT)


def build():
    import __graft_entry__ as G
    G.build()
    os.makedirs(AB, exist_ok=True)
    others = [os.path.join(G.LIBDIR, s.replace(".hip", ".o")) for s in G.SOURCES if s != "gemm_f32.hip"]
    procs = []
    for k in (0, 1, 2, 3):        # 3 = packed GELU but the old bounds-checked epilogue everywhere
        o = os.path.join(AB, f"gemm_f32_gelu{k}.o")
        flags = [f"-DTHMR_GELU_IMPL={min(k, 2)}"] + (["-DTHMR_NO_FAST_EPILOGUE"] if k == 3 else [])
        procs.append((k, o, subprocess.Popen([G._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
                                             + flags + ["-c", os.path.join(G.CSRC, "gemm_f32.hip"), "-o", o])))
    for k, o, p in procs:
        assert p.wait() == 0
        subprocess.check_call([G._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(AB, f"libgelu{k}.so"), o] + others)
    print("built", os.listdir(AB))


def run(rounds):
    import torch
    dev = torch.device("cuda:0")
    libs = {}
    for k in (0, 1, 2, 3):
        lib = C.CDLL(os.path.join(AB, f"libgelu{k}.so"))
        vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
        lib.thmr_op_gemm.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp]
        libs[k] = lib
    M, N, K = 64 * 192, 5120, 1280
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = {k: torch.empty(M, N, device=dev) for k in libs}
    s = torch.cuda.current_stream().cuda_stream

    def call(k, epi):
        rc = libs[k].thmr_op_gemm(a.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, out[k].data_ptr(), N, M, N, K, epi, 1.0, 0, 8, s)
        assert rc == 0

    for k in libs:
        call(k, 2)
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double())
    for k in libs:
        print(f"impl {k}: max|err| vs fp64 = {(out[k].double() - ref).abs().max().item():.3e}")
    print("impl1 == impl2 bitwise:", torch.equal(out[1], out[2]))
    print("fast epilogue == checked epilogue bitwise:", torch.equal(out[2], out[3]))
    # proj / fc2 shapes (bias + residual) for the epilogue A/B
    shapes = {"fc1_gelu": (N, K, 2), "fc1_bias": (N, K, 1), "proj_resid": (1280, 1280, 3), "fc2_resid": (1280, 5120, 3)}
    a2 = torch.randn(M, 5120, generator=g).to(dev)
    w2 = (torch.randn(5120, 5120, generator=g) / 70).to(dev)
    r2 = torch.randn(M, 1280, generator=g).to(dev)
    o2 = torch.empty(M, 5120, device=dev)

    def call2(k, n, kk, epi):
        rc = libs[k].thmr_op_gemm(a2.data_ptr(), 5120, w2.data_ptr(), b.data_ptr(), r2.data_ptr() if epi == 3 else None, o2.data_ptr(), n,
                                  M, n, kk, epi, 1.0, 0, 8, s)
        assert rc == 0

    times = {(nm, k): [] for nm in shapes for k in libs}
    for _ in range(rounds):
        for nm, (n, kk, epi) in shapes.items():
            for k in libs:
                call2(k, n, kk, epi)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    call2(k, n, kk, epi)
                e1.record()
                torch.cuda.synchronize()
                times[(nm, k)].append(e0.elapsed_time(e1) / 4)
    for (nm, k), t in sorted(times.items()):
        n, kk, _ = shapes[nm]
        med = sorted(t)[len(t) // 2]
        print(f"{nm:11s} lib {k}: {med * 1e3:8.1f} us  {2.0 * M * n * kk / (med * 1e-3) / 1e12:6.1f} TF")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
