// Tiled fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every large nn.Linear of the reference's hot path (ATen/rocBLAS sgemm there):
//   ViT qkv/proj/fc1/fc2      tokenhmr/lib/models/backbones/vit.py:104,112,123 ; :82-87
//   patch-embed (after im2col) vit.py:168,172        decoder to_kv   pose_transformer.py:102,113
//   classifier / VQ-decoder convs as GEMMs           token_classifier.py:93-101, vanilla_pose_vqvae.py:135-154
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = 157 TF chip peak.
//   * Both operands are K-contiguous (torch Linear layout), so A and W tiles are staged identically:
//     global float4 -> registers -> LDS rows of 32 floats whose eight 16-B slots are XOR-swizzled with
//     (row>>1)&7 (conflict-free for the four 16-lane groups of ds_read_b128 and for the 8-lane groups of
//     ds_write_b128, no padding).
//   * pipeline: ONE barrier per 32-deep K tile.  Inside iteration t the register-staged tile (global-loaded
//     one iteration earlier) is written to LDS right after the first 16 MFMAs and the next global loads are
//     issued after the second 16, so all staging traffic sits in the shadow of the 64-cycle MFMAs.
//       STAGES = 2: two LDS buffers, 2 blocks/CU (the co-resident block covers the barrier bubble);
//       STAGES = 3: three LDS buffers, the first fragments of tile t+1 are read BEFORE the barrier
//                   (they were written two barriers ago), so a lone block per CU has no LDS-latency bubble.
//   * k-permutation trick: one ds_read_b128 gives a lane 4 consecutive k of its row; lanes 0-31 take
//     k0..k0+3 and lanes 32-63 take k0+4..k0+7, so MFMA step t multiplies k0+t (lower half) and
//     k0+4+t (upper half).  A and W use the same permutation, hence the sum over k is unchanged.
//   * wave tile = TM x TN blocks of 32x32 (16 accumulator VGPRs each); block = WM x WN waves.
//   * XCD-aware tile order: block b runs on XCD b%8, so consecutive logical tiles (which share A
//     row panels / W column panels) are given to the same XCD's L2.
//   * fused epilogues: bias, exact-erf GELU, ReLU, residual add, q-scale, pos-embed add.  The epilogue
//     loads a whole 16-row fragment of residual/pos values BEFORE combining (no load->wait->store chains).
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 32;   // LDS row (floats) = 8 slots of 16 B; slot' = slot ^ ((row >> 1) & 7)

template <int WM, int WN, int TM, int TN, int STAGES, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int A_F4 = BM * 8 / NT;   // float4 loads per thread per K tile
    constexpr int B_F4 = BN * 8 / NT;
    constexpr int NJ = BK / 8;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");
    static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");

    __shared__ __attribute__((aligned(16))) float smem[STAGES * (BM + BN) * LDK];
    float* As = smem;                         // [STAGES][BM][LDK]
    float* Bs = smem + STAGES * BM * LDK;     // [STAGES][BN][LDK]

    // ---- XCD-aware logical tile id (bijective for any grid size) ----
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    // grouped order: GM tile-rows per group, tile_n slow, tile_m fast inside the group
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = logical / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_group = logical - group * per_group;
    const int tile_m = first_m + in_group % gsz;
    const int tile_n = in_group / gsz;
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * TM * 32;
    const int wn0 = (wave % WN) * TN * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // ---- global -> register staging ----
    // Rows past M / N are clamped to a valid row and NOT zeroed: an output element depends only on its own
    // A row and W row, and rows/cols past the edge are never stored, so their (duplicate) data is harmless.
    // Keeping the loaded registers untouched lets the loads stay in flight across the barrier.
    f32x4 ra[A_F4], rb[B_F4];
    const float* Ag[A_F4];
    const float* Wg[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * NT, row = f >> 3, c4 = f & 7;
        Ag[i] = a.A + (int64_t)min(bm0 + row, a.M - 1) * a.lda + c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int f = tid + i * NT, row = f >> 3, c4 = f & 7;
        Wg[i] = a.W + (int64_t)min(bn0 + row, a.N - 1) * a.ldw + c4 * 4;
    }
    auto load_global = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ag[i] + k0);
#pragma unroll
        for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wg[i] + k0);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * NT, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * LDK + ((c4 ^ ((row >> 1) & 7)) << 2)]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * NT, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + row) * LDK + ((c4 ^ ((row >> 1) & 7)) << 2)]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // swizzled float offset of logical slot (2j + lhalf) for this lane's rows (wm0, mi*32 are multiples of 16)
    int koff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) koff[j] = (((2 * j + lhalf) ^ ((lrow >> 1) & 7)) << 2);

    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int buf, int j, int slot) {
        const float* Ab = As + (buf * BM + wm0 + lrow) * LDK + koff[j];
        const float* Bb = Bs + (buf * BN + wn0 + lrow) * LDK + koff[j];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[slot][mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * LDK);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf[slot][ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * LDK);
    };

    const int nk = a.K / BK;
    // ---- prologue: STAGES-1 tiles in LDS, one more in registers ----
    load_global(0);
    store_lds(0);
    if constexpr (STAGES == 3) {
        load_global(min(1, nk - 1));
        store_lds(1);
        load_global(min(2, nk - 1));
    } else {
        load_global(min(1, nk - 1));
    }
    __syncthreads();
    if constexpr (STAGES == 3) read_frags(0, 0, 0);

    int buf = 0;                       // LDS buffer of tile kt
    for (int kt = 0; kt < nk; ++kt) {
        int wbuf = buf + (STAGES - 1);             // buffer receiving tile kt + STAGES - 1
        if (wbuf >= STAGES) wbuf -= STAGES;
        int nbuf = buf + 1;
        if (nbuf >= STAGES) nbuf -= STAGES;
        if constexpr (STAGES == 2) read_frags(buf, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + 1 < NJ) read_frags(buf, j + 1, (j + 1) & 1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][mi][t], bf[j & 1][ni][t],
                                                                           acc[mi][ni], 0, 0, 0);
            // staging in the MFMA shadow: registers -> LDS after the first k-group, next global loads after the second
            // (unconditional: past the last tile the writes land in a buffer nobody reads again and the loads
            //  re-read the last tile, which keeps the loop body branch-free)
            if (j == 0) store_lds(wbuf);
            if (j == 1) load_global(min(kt + STAGES, nk - 1));
        }
        if constexpr (STAGES == 3) {
            // tile kt+1 was written during iteration kt-1, i.e. two barriers ago: safe to read before this barrier
            if (kt + 1 < nk) read_frags(nbuf, 0, 0);
        }
        __syncthreads();
        buf = nbuf;
    }

    // ---- epilogue: C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = bn0 + wn0 + ni * 32 + lrow;
            const int nc = min(n, a.N - 1);
            const int mbase = bm0 + wm0 + mi * 32 + 4 * lhalf;
            float bias = 0.f;
            if constexpr (EPI != EPI_NONE) bias = a.bias[nc];
            float extra[16];
            if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = min(mbase + (e & 3) + 8 * (e >> 2), a.M - 1);
                    extra[e] = a.resid[(int64_t)m * a.ldr + nc];
                }
            }
            float pos0 = 0.f;
            if constexpr (EPI == EPI_BIAS_POS) {
                pos0 = a.resid[nc];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mbase + (e & 3) + 8 * (e >> 2);
                    extra[e] = a.resid[(int64_t)(1 + m % 192) * a.N + nc];
                }
            }
            float outv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mi][ni][e];
                if constexpr (EPI != EPI_NONE) v = v + bias;
                if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
                if constexpr (EPI == EPI_BIAS_RESID) v = extra[e] + v;
                if constexpr (EPI == EPI_BIAS_QSCALE) v = (n < a.qcols) ? v * a.qscale : v;
                if constexpr (EPI == EPI_BIAS_POS) v = (v + extra[e]) + pos0;
                outv[e] = v;
            }
            if (n < a.N) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mbase + (e & 3) + 8 * (e >> 2);
                    if (m < a.M) a.C[(int64_t)m * a.ldc + n] = outv[e];
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int STAGES>
int launch_cfg(const GemmArgs& a, int epi, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
#define THMR_GEMM_CASE(E)                                                                                         \
    case E:                                                                                                       \
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, STAGES, E>), grid, block, 0, s, a, tiles_m, tiles_n); \
        break;
    switch (epi) {
        THMR_GEMM_CASE(EPI_NONE)
        THMR_GEMM_CASE(EPI_BIAS)
        THMR_GEMM_CASE(EPI_BIAS_GELU)
        THMR_GEMM_CASE(EPI_BIAS_RELU)
        THMR_GEMM_CASE(EPI_BIAS_RESID)
        THMR_GEMM_CASE(EPI_BIAS_QSCALE)
        THMR_GEMM_CASE(EPI_BIAS_POS)
        default: return -1;
    }
#undef THMR_GEMM_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

inline double tile_efficiency(int M, int N, int BM, int BN) {
    // useful fraction of the MFMA work issued, including the partial last wave over 256 CUs
    const long tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    const double useful = (double)M * N;
    const long tiles = tm * tn;
    const long rounds = (tiles + 255) / 256;          // per-CU tile count of the busiest CU
    return useful / ((double)rounds * 256 * BM * BN);
}

}  // namespace

// variant: 0 = 128x128 2-stage (2x2 waves of 64x64)      1 = 128x160 2-stage (4x1 waves of 32x160)
//          3 = 128x128 3-stage                            4 = 128x160 3-stage
//          5 = 256x128 3-stage (2x2 waves of 128x64)      6 = 256x128 2-stage
//         -1 = pick by tile quantisation over 256 CUs     (2 is the skinny kernel, see thmr_op_gemm)
int launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % BK) != 0) return -1;
    if ((a.lda % 4) != 0 || (a.ldw % 4) != 0) return -1;
    if (variant < 0) {
        const double e0 = tile_efficiency(a.M, a.N, 128, 128);
        const double e1 = tile_efficiency(a.M, a.N, 128, 160);
        variant = (e1 > e0 * 1.02) ? 1 : 0;
    }
    switch (variant) {
        case 1: return launch_cfg<4, 1, 1, 5, 2>(a, epi, s);
        case 3: return launch_cfg<2, 2, 2, 2, 3>(a, epi, s);
        case 4: return launch_cfg<4, 1, 1, 5, 3>(a, epi, s);
        case 5: return launch_cfg<2, 2, 4, 2, 3>(a, epi, s);
        case 6: return launch_cfg<2, 2, 4, 2, 2>(a, epi, s);
        default: return launch_cfg<2, 2, 2, 2, 2>(a, epi, s);
    }
}
