// Skinny fp32 GEMM for the 1-token decoder path:  C[M,N] = epilogue(A[M,K] . W[N,K]^T),  M = #crops (<= 64 per
// block row), i.e. weight-streaming "GEMV-class" work.
//
// Replaces the M = B Linears of pose_transformer.py:40-52,67,69-73,103,105-109 (to_qkv's v slice, to_out,
// to_q, FF), token_head.py:40-43,99-105 (read-outs) and token_classifier.py:71-73 (mixer_trans Linear).
//
// gfx950 design: every block owns 16 output columns for 64 rows; its 4 waves split K four ways (so
// N/16 blocks x 4 waves stream the weight matrix exactly once, 16-byte loads straight from global into
// MFMA fragments — no LDS round trip for a stream that is read once); v_mfma_f32_16x16x4_f32 with the
// k-permutation trick; partial sums are combined through LDS in a FIXED order (deterministic), then the
// fused epilogue is applied.  K % 64 == 0.
#include "common.h"

namespace {

template <int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][4][64][4];   // [wave][mtile][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int kper = a.K >> 2;
    const int kbeg = wave * kper;

    const float* wp = a.W + (int64_t)min(n0 + l15, a.N - 1) * a.ldw + kbeg + g * 4;
    const float* ap[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ap[mt] = a.A + (int64_t)min(m0 + mt * 16 + l15, a.M - 1) * a.lda + kbeg + g * 4;

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The stream is read once and nothing else hides its latency, so the loads run THREE K steps (of 16) ahead of the MFMAs
    // that consume them: a 4-slot register ring (the first version loaded and multiplied step by step: one exposed memory
    // round trip per 16 k).
    constexpr int PF = 3;
    f32x4 wq[PF + 1], xq[PF + 1][4];
    auto fetch = [&](int k, int slot) {
        wq[slot] = *reinterpret_cast<const f32x4*>(wp + k);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xq[slot][mt] = *reinterpret_cast<const f32x4*>(ap[mt] + k);
    };
    const int nstep = kper >> 4;
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < nstep) fetch(i * 16, i);
    for (int i0 = 0; i0 < nstep; i0 += PF + 1) {
#pragma unroll
        for (int u = 0; u < PF + 1; ++u) {          // ring slots are compile-time constants (registers, not scratch)
            const int i = i0 + u;
            if (i < nstep) {
                if (i + PF < nstep) fetch((i + PF) * 16, (u + PF) % (PF + 1));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[u][mt][t], wq[u][t], acc[mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(&red[wave][mt][lane][0]) = acc[mt];
    __syncthreads();

    // wave w finalises M-tile w: fixed summation order over the 4 K-slices
    const int mt = wave;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][mt][lane][0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[w][mt][lane][0]);
        v += p;
    }
    const int n = n0 + l15;
    if (n < a.N) {
        float bias = 0.f;
        if constexpr (EPI != EPI_NONE) bias = a.bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + mt * 16 + g * 4 + r;   // D layout 16x16: row = 4*(lane>>4) + reg, col = lane&15
            if (m < a.M) a.C[(int64_t)m * a.ldc + n] = gemm_epilogue<EPI>(a, v[r], bias, m, n);
        }
    }
}

}  // namespace

int launch_gemm_skinny(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 64) != 0) return -1;
    if ((a.lda % 4) != 0 || (a.ldw % 4) != 0) return -1;
    dim3 grid((a.N + 15) / 16, (a.M + 63) / 64), block(256);
#define THMR_SK_CASE(E) \
    case E: hipLaunchKernelGGL((gemm_skinny_kernel<E>), grid, block, 0, s, a); break;
    switch (epi) {
        THMR_SK_CASE(EPI_NONE)
        THMR_SK_CASE(EPI_BIAS)
        THMR_SK_CASE(EPI_BIAS_GELU)
        THMR_SK_CASE(EPI_BIAS_RELU)
        THMR_SK_CASE(EPI_BIAS_RESID)
        default: return -1;
    }
#undef THMR_SK_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
