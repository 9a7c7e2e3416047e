// Device helpers shared by the GEMM kernels (gemm_f32.hip: exact-fp32 MFMA tiles; gemm_split.hip: fp32 operands split into three
// bf16 pieces): compile-time loops, the saddr-form LDS-DMA copy, the XCD-aware tile order and the fused-epilogue tile store.
#pragma once
#include "common.h"

namespace {

template <int V>
struct IntC {
    constexpr operator int() const { return V; }
};
// f(IntC<0>{}), f(IntC<1>{}), ..., f(IntC<N-1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<N, I + 1>(f);
    }
}

// One LDS-DMA wave instruction in the saddr form: 64 x 16 B from {uniform 64-bit base} + {per-lane 32-bit byte offset} to the
// 1 KiB of LDS at `lds_base` (wave-uniform, goes through M0).  Written as inline assembly because hipcc only selects the
// saddr form when the zero-extension of the offset sits in the same basic block as the load; with the offsets hoisted out of
// the K loop it falls back to a 64-bit VALU add per copy.  The compiler does not count these copies in vmcnt: every consumer
// must wait with dma_wait_barrier() below.  M0 is a reserved register that cannot be named as a clobber; the kernels using
// this helper have no other M0 user (no LDS-DMA builtin, no movrel / sendmsg), and hipcc re-materialises M0 before its own uses.
__device__ __forceinline__ void dma16_saddr(const char* base, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_base));
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
// all of this wave's LDS-DMA copies and LDS reads done, then the workgroup barrier
__device__ __forceinline__ void dma_wait_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// XCD-aware logical tile id (bijective on [0, nwg) for any launch size nwg, offset by the launch's first tile) -> (tile_m, tile_n),
// grouped GM tile-rows at a time
__device__ __forceinline__ int logical_block(int nwg, int base) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int logical, int& tile_m, int& tile_n) {
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = logical / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_group = logical - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
}

// C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile(const GemmArgs& a, f32x16 (&acc)[TM][TN], int m0, int n0, int lrow, int lhalf) {
#if !defined(THMR_NO_FAST_EPILOGUE)
    // Interior wave tiles (every ViT GEMM at batch sizes that are multiples of 2 has only these): no bounds checks, and the
    // address of element (mi, ni, e) is a wave-uniform pointer (SALU) plus ONE per-lane 32-bit offset computed once, so each
    // store / residual load is a single saddr-form global instruction instead of ~12 VALU of 64-bit address arithmetic.
    if constexpr (EPI != EPI_BIAS_POS) {
        if (m0 + TM * 32 <= a.M && n0 + TN * 32 <= a.N && a.ldc < (1 << 24) && a.ldr < (1 << 24)) {   // wave-uniform
            const uint32_t coff = (uint32_t)(4 * lhalf) * (uint32_t)a.ldc + (uint32_t)lrow;
            const uint32_t roff = (uint32_t)(4 * lhalf) * (uint32_t)a.ldr + (uint32_t)lrow;
            float* Cw = a.C + (int64_t)m0 * a.ldc + n0;
            const float* Rw = nullptr;
            if constexpr (EPI == EPI_BIAS_RESID) Rw = a.resid + (int64_t)m0 * a.ldr + n0;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    float bias = 0.f;
                    if constexpr (EPI != EPI_NONE) bias = (a.bias + n0 + ni * 32)[lrow];
                    float extra[16];
                    if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            extra[e] = (Rw + (int64_t)(mi * 32 + (e & 3) + 8 * (e >> 2)) * a.ldr + ni * 32)[roff];
                    }
                    float outv[16];
                    if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) {
                            const f32x2 g = gelu_erf2(f32x2{acc[mi][ni][e] + bias, acc[mi][ni][e + 1] + bias});
                            outv[e] = g.x;
                            outv[e + 1] = g.y;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float v = acc[mi][ni][e];
                            if constexpr (EPI != EPI_NONE) v = v + bias;
                            if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
                            if constexpr (EPI == EPI_BIAS_RESID) v = extra[e] + v;
                            if constexpr (EPI == EPI_BIAS_QSCALE) v = (n0 + ni * 32 + lrow < a.qcols) ? v * a.qscale : v;
                            outv[e] = v;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        (Cw + (int64_t)(mi * 32 + (e & 3) + 8 * (e >> 2)) * a.ldc + ni * 32)[coff] = outv[e];
                }
            }
            return;
        }
    }
#endif
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + ni * 32 + lrow;
            const int nc = min(n, a.N - 1);
            const int mbase = m0 + mi * 32 + 4 * lhalf;
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
            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 g = gelu_erf2(f32x2{acc[mi][ni][e] + bias, acc[mi][ni][e + 1] + bias});
                    outv[e] = g.x;
                    outv[e + 1] = g.y;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr (EPI == EPI_BIAS_GELU) break;
                float v = acc[mi][ni][e];
                if constexpr (EPI != EPI_NONE) v = v + bias;
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

}  // namespace
