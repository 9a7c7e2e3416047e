// Device helpers shared by the ViT attention kernels (attention.hip: fp32 MFMA; attention_b16.hip: split3 pieces on the bf16 MFMA).
#pragma once
#include "common.h"

namespace {

constexpr int NTOK = 192, HD = 80, NH = 16, DIM = 1280, QKV_LD = 3840;

// SPLIT output of a wave's O tiles.  The MFMA leaves lane (l15, g) with 4 consecutive d of query l15 per (qt, dt) tile: columns
// h 80 + 16 dt + 4 g ... + 3 — the 8-byte LOWER half of a split3 chunk group for even g, the upper half for odd g.  Written as such that is
// three 8-byte stores per tile (45 per item; measured 137 vs 102 us per launch at 64 crops, profiles/r3ah_split3_kernel_stats.csv).  Two
// tiles X, Y at a time, v_permlane16_swap exchanges X's odd 16-lane rows with Y's even rows: afterwards an even-g lane holds all 8 columns
// of tile X's group (its own half + its neighbour's) and the odd-g lane next to it all 8 of tile Y's — three 16-byte stores of 48
// contiguous bytes each, 21 + 3 stores per item instead of 45.  The values are the same fp32 numbers, so the pieces are too.
struct SplitPair { int qa, da, qb, db; };
template <int QT>
__device__ __forceinline__ void store_o_split3(char* out, int64_t tok0, int col0, int l15, int g, f32x4 (&o)[QT][5], const float (&inv)[QT]) {
    constexpr int NPAIR = QT == 3 ? 7 : 2 * QT;
    constexpr SplitPair P3[7] = {{0, 0, 0, 1}, {0, 2, 0, 3}, {1, 0, 1, 1}, {1, 2, 1, 3}, {2, 0, 2, 1}, {2, 2, 2, 3}, {0, 4, 1, 4}};
    const bool odd = (g & 1) != 0;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        const SplitPair pr = QT == 3 ? P3[p] : SplitPair{p / 2, 2 * (p % 2), p / 2, 2 * (p % 2) + 1};
        f32x4 x = o[pr.qa][pr.da] * inv[pr.qa], y = o[pr.qb][pr.db] * inv[pr.qb];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[j]), __float_as_uint(y[j]), false, false);
            x[j] = __uint_as_float(sw.x);
            y[j] = __uint_as_float(sw.y);
        }
        const int q = odd ? pr.qb : pr.qa, dt = odd ? pr.db : pr.da;
        store_split3_oct(out + (tok0 + q * 16 + l15) * (DIM * 6), col0 + dt * 16 + (g & 2) * 4, x, y);
    }
    // the tile without a partner (QT = 3: (2, 4); otherwise every (qt, 4)): the 8-byte halves
#pragma unroll
    for (int qt = (QT == 3 ? 2 : 0); qt < QT; ++qt)
        store_split3_quad(out + (tok0 + qt * 16 + l15) * (DIM * 6), col0 + 4 * 16 + g * 4, o[qt][4] * inv[qt]);
}
constexpr int kSplitStores3 = 7 * 3 + 3;       // VMEM stores per item of store_o_split3<3> (the persistent kernel counts them: vmcnt)

}  // namespace
