// Device helpers shared by the split3 GEMM kernels (gemm_split.hip: one workgroup per tile; gemm_split_persist.hip: persistent workgroups
// over a tile stream): the operand-format constants, the transposing split3 epilogue and the order of the six piece products.
#pragma once
#include "common.h"
#include "gemm_device.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;                     // K tile, fp32 elements
constexpr int SLOTS = SBK / 8 * 3;          // 16-byte chunks per row and K tile
constexpr int ROWB = SLOTS * 16;            // 192 bytes

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_b(const char* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// The epilogue's result as the NEXT GEMM's split3 operand (GemmArgs::c_split).  The MFMA leaves a lane with one column and 16 rows; a
// split3 chunk is 8 consecutive columns of one row, so the wave tile is transposed through LDS (T: wave-private, TM*32 rows of
// WT = 32 TN + 4 floats, in stage buffers that are dead by now): lane (row r, column group q) reads 8 consecutive columns, applies bias +
// activation, splits and writes 48 contiguous bytes; the 4 TN lanes of a row write 192 TN contiguous bytes.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile_split3(const GemmArgs& a, f32x16 (&acc)[TM][TN], float* T, int m0, int n0, int lane) {
    constexpr int WT = TN * 32 + 4;
    constexpr int GQ = TN * 4;                                     // column groups of 8 per wave-tile row
    constexpr int RP = 64 / GQ;                                    // rows per pass
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                T[(mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf) * WT + ni * 32 + lrow] = acc[mi][ni][e];
    const int q = lane % GQ, r0 = lane / GQ;
    const int n = n0 + q * 8;
    float bias[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bias[u] = (EPI != EPI_NONE && n + u < a.N) ? a.bias[n + u] : 0.f;
#pragma unroll
    for (int ps = 0; ps < TM * 32 / RP; ++ps) {
        const int r = r0 + ps * RP, m = m0 + r;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + r * WT + q * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + r * WT + q * 8 + 4);
        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const f32x2 gl = gelu_erf2(f32x2{v[u] + bias[u], v[u + 1] + bias[u + 1]});
                v[u] = gl.x;
                v[u + 1] = gl.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gemm_epilogue<EPI>(a, v[u], bias[u], min(m, a.M - 1), min(n + u, a.N - 1));
        }
        uint32_t H[4], M[4], L[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) split3_pair(v[2 * u], v[2 * u + 1], H[u], M[u], L[u]);
        if (m < a.M && n < a.N) {
            char* cb = reinterpret_cast<char*>(a.c_split);
            const bool blk = a.cs_blk != 0;
            *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 0, blk)) = u32x4{H[0], H[1], H[2], H[3]};
            *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 1, blk)) = u32x4{M[0], M[1], M[2], M[3]};
            *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 2, blk)) = u32x4{L[0], L[1], L[2], L[3]};
        }
    }
}

// the six piece pairs (A piece, W piece) kept, smallest terms first: lh hl mm mh hm hh
constexpr int NPROD = 6;
constexpr int piece_a(int p) { return p == 0 ? 2 : (p == 2 || p == 3) ? 1 : 0; }
constexpr int piece_w(int p) { return p == 1 ? 2 : (p == 2 || p == 4) ? 1 : 0; }

}  // namespace
