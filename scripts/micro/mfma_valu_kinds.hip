// Micro-benchmark 2: which VALU instruction kinds steal time from back-to-back fp32 MFMAs on gfx950?
// (mfma_valu_overlap.hip showed that v_pk_fma_f32 does not overlap with v_mfma_f32_32x32x2_f32 at all.)
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/mfma_valu_kinds scripts/micro/mfma_valu_kinds.hip
// Per wave: ITER x 5 independent MFMAs, each followed by NV filler instructions of kind OP on independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { PK_FMA = 0, FMA32, ADD_U32, LSHL_ADD_U64, MOV, SALU_ADD, FMA64, DS_READ, NKIND };
static const char* kNames[NKIND] = {"v_pk_fma_f32", "v_fma_f32", "v_add_u32", "v_lshl_add_u64", "v_mov_b32", "s_add_u32", "v_fma_f64", "ds_read_b128"};

template <int OP, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x2 v[8];
    uint64_t w[8];
    uint32_t u[8];
    double d[8];
    uint32_t sacc = 0;
    __shared__ float lds[256 * 4 * 8];
    for (int i = threadIdx.x; i < 256 * 4 * 8; i += 256) lds[i] = (float)i;
    __syncthreads();
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 q[8];
    const uint32_t laddr = (uint32_t)(uintptr_t)lds + threadIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = f32x2{(float)threadIdx.x, (float)i};
        w[i] = threadIdx.x + i;
        u[i] = threadIdx.x * 3 + i;
        d[i] = threadIdx.x + i;
    }
    const float a = a0 + threadIdx.x * 1e-9f, b = b0;
    const f32x2 va = f32x2{a, a * 0.5f}, vb = f32x2{b * 1e-3f, b * 2e-3f};
    const double da = a, db = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = (m * NV + j) % 8;
                if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(va), "v"(vb));
                if constexpr (OP == FMA32) { float t = v[r].x; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(a), "v"(b)); v[r].x = t; }
                if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) % 8]));
                if constexpr (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[r]) : "v"(w[(r + 1) % 8]));
                if constexpr (OP == MOV) { float t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(a)); v[r].y = t; }
                if constexpr (OP == SALU_ADD) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if constexpr (OP == DS_READ) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[r]) : "v"(laddr), "n"(4096 * (r % 8)));
                if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[r]) : "v"(da), "v"(db));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = (float)sacc;
    if constexpr (OP == DS_READ) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][3];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + (float)w[i] + (float)d[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
static double g_base_ms;

template <int OP, int NV>
void run(int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    if (NV == 0) g_base_ms = ms;
    const double waves = (double)blocks * 4, mf = waves * iters * 5.0 * 4096.0;
    // extra time per filler instruction, in MFMA-pipe cycles of one SIMD: 64 cycles per MFMA <-> g_base_ms per (iters*5*waves/SIMD)
    const double per = NV ? (ms - g_base_ms) / g_base_ms * 64.0 / NV : 0.0;
    printf("%-15s x%2d per MFMA: %8.3f ms   MFMA %6.1f TF   (+%5.1f %%, ~%4.1f MFMA-pipe cycles per filler op)\n", kNames[OP], NV, ms, mf / ms / 1e9,
           (ms / g_base_ms - 1) * 100, per);
}

template <int OP>
void kind(int blocks, int iters) {
    run<OP, 0>(blocks, iters);
    run<OP, 1>(blocks, iters);
    run<OP, 2>(blocks, iters);
    run<OP, 4>(blocks, iters);
    run<OP, 8>(blocks, iters);
}

int main() {
    (void)hipMalloc(&g_out, 4096 * 256 * sizeof(float));
    const int iters = 8000, blocks = 512;     // 2 waves per SIMD, as in the product GEMM
    kind<PK_FMA>(blocks, iters);
    kind<FMA32>(blocks, iters);
    kind<ADD_U32>(blocks, iters);
    kind<LSHL_ADD_U64>(blocks, iters);
    kind<MOV>(blocks, iters);
    kind<SALU_ADD>(blocks, iters);
    kind<FMA64>(blocks, iters);
    kind<DS_READ>(blocks, iters);
    return 0;
}
