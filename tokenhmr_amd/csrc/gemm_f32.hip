// Tiled fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every large nn.Linear of the reference's hot path (ATen/rocBLAS sgemm there):
//   ViT qkv/proj/fc1/fc2      tokenhmr/lib/models/backbones/vit.py:104,112,123 ; :82-87
//   patch-embed (after im2col) vit.py:168,172        decoder to_kv   pose_transformer.py:102,113
//   classifier / VQ-decoder convs as GEMMs           token_classifier.py:93-101, vanilla_pose_vqvae.py:135-154
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = 157 TF chip peak.
//   * Both operands are K-contiguous (torch Linear layout), so A and W tiles are staged identically into
//     LDS rows of 32 floats whose eight 16-B slots are XOR-swizzled with (row>>1)&7: conflict-free for the
//     four 16-lane groups of ds_read_b128, no padding (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * two staging paths, same LDS image:
//       DMA  (product path): global_load_lds_dwordx4 — HBM/L2 -> LDS directly, no VGPR round trip and no
//            ds_write.  The LDS destination of a wave instruction is lane-linear (base + 16*lane = 8 rows x
//            8 slots), so the swizzle is applied to the per-lane SOURCE address: lane (r, p) fetches the
//            logical slot p ^ ((r>>1)&7) of row r.  Every 8 lanes still read one whole 128-B line.
//       REG  (kept for A/B): float4 global loads -> registers -> ds_write_b128 in the MFMA shadow.
//     Ablation on MI355X (profiles/r1_gemm_ablation_raw.log, r1_gemm_dma_ablation_raw.log): the register staging traffic cost 10-13 % of the
//     MFMA rate, the barrier 0-4 %, a pure-MFMA loop of this shape runs at 145-155 TF.
//   * pipeline: two LDS buffers, ONE barrier per 32-deep K tile, 2 blocks per CU (the co-resident block
//     covers the barrier / first-fragment bubble; 3-stage single-block variants measured 15 % slower).
//   * NO VALU instruction in the K loop: on gfx950 a VALU instruction issued between fp32 MFMAs costs 4-14 cycles of
//     matrix-pipe time even with a second wave resident (profiles/r1_mfma_valu_microbench.log).  The copies use the saddr
//     form (uniform base bumped on the SALU + constant 32-bit lane offsets, inline asm) and the K loop is unrolled by two
//     so that every LDS address is a loop-invariant register + an immediate: +5-7 % (profiles/r1_gemm_zero_valu_loop.log).
//   * k-permutation trick: one ds_read_b128 gives a lane 4 consecutive k of its row; lanes 0-31 take
//     k0..k0+3 and lanes 32-63 take k0+4..k0+7, so MFMA step t multiplies k0+t (lower half) and
//     k0+4+t (upper half).  A and W use the same permutation, hence the sum over k is unchanged.
//   * wave tile = TM x TN blocks of 32x32 (16 accumulator VGPRs each); block = WM x WN waves.
//   * XCD-aware tile order: block b runs on XCD b%8, so consecutive logical tiles (which share A
//     row panels / W column panels) are given to the same XCD's L2.
//   * fused epilogues: bias, exact-erf GELU, ReLU, residual add, q-scale, pos-embed add.  The epilogue
//     loads a whole 16-row fragment of residual/pos values BEFORE combining (no load->wait->store chains).
#include <cstdlib>

#include "common.h"
#include "gemm_device.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 32;   // LDS row (floats) = 8 slots of 16 B; slot' = slot ^ ((row >> 1) & 7)
constexpr int NJ = BK / 8;



// Epilogue that also scatters the tile into the NEXT Conv1d's im2col operand (GemmArgs::cs_*).  Element-wise and bounds-checked:
// it only runs on the VQ decoder's small GEMMs (M = 21 ... 160 rows per crop), where it replaces one gather launch per conv.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile_scatter(const GemmArgs& a, f32x16 (&acc)[TM][TN], int m0, int n0, int lrow, int lhalf) {
    const int64_t ld3 = (int64_t)3 * a.N;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mi * 32 + 4 * lhalf + (e & 3) + 8 * (e >> 2);
            if (m >= a.M) continue;
            const int b = m / a.cs_tin, ts = m - b * a.cs_tin;
            const int tp = a.cs_inv ? a.cs_inv[ts] : ts;
            float* g = a.cs_out + (int64_t)b * a.cs_tout * ld3;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + ni * 32 + lrow;
                if (n >= a.N) continue;
                float bias = 0.f;
                if constexpr (EPI != EPI_NONE) bias = a.bias[n];
                const float v = gemm_epilogue<EPI>(a, acc[mi][ni][e], bias, m, n);
                if (a.C) a.C[(int64_t)m * a.ldc + n] = v;
                if (tp < 0) continue;
                const float f = a.cs_relu ? fmaxf(v, 0.f) : v;
                float* gn = g + n;
                if (tp + a.cs_dil < a.cs_tout) gn[(int64_t)(tp + a.cs_dil) * ld3] = f;                 // tap 0 of row tp + dil
                else gn[(int64_t)tp * ld3 + 2 * a.N] = 0.f;                                             // own tap 2 is padding
                gn[(int64_t)tp * ld3 + a.N] = f;                                                        // tap 1 of row tp
                if (tp - a.cs_dil >= 0) gn[(int64_t)(tp - a.cs_dil) * ld3 + 2 * a.N] = f;               // tap 2 of row tp - dil
                else gn[(int64_t)tp * ld3] = 0.f;                                                        // own tap 0 is padding
            }
        }
    }
}

// DMA = true: global_load_lds staging; DMA = false: register staging
// ABL (timing-only experiments, results are garbage): bit0 no DMA copies in the loop, bit1 no per-tile barrier,
// bit2 no LDS fragment reads.  ABL = 0 is the product kernel.
// DS: one DMA piece every DS-th MFMA (0 = auto).  BARPOS (DMA path): per-tile barrier pinned BARPOS (0 = 32) MFMAs before the
// end of the tile, never before the tile's last staging instruction (see GBAR below).
#ifndef THMR_GEMM_BARPOS
#define THMR_GEMM_BARPOS 0
#endif
// DPH (experiment, scripts/micro/gemm_dephase.hip): the block on the odd threadgroup slot of its CU starts g_gemm_dephase_ticks
// (10 ns each) late when it belongs to the first 512 blocks of the grid, so that the two co-resident blocks of a CU do not run their
// prologues / epilogues at the same time.
__device__ int g_gemm_dephase_ticks;
template <int WM, int WN, int TM, int TN, bool DMA, int EPI, int ABL = 0, int DSP = 0, int BARPOS = THMR_GEMM_BARPOS, int DPH = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmArgs a, int tiles_m, int tiles_n, int nwg, int tile_base) {
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int A_F4 = BM * 8 / NT;   // 16-byte pieces per thread per K tile (== wave instructions per wave)
    constexpr int B_F4 = BN * 8 / NT;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
    float* As = smem;                    // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;     // [2][BN][LDK]

    int tile_m, tile_n;
    int logical = logical_block(nwg, tile_base);
    int nk = a.K / BK;
    if (a.ksplit > 1) {      // split-K: this block belongs to copy sp of the tile grid and reduces K slice sp into part[sp] (wave-uniform)
        const int tiles = tiles_m * tiles_n, sp = logical / tiles;
        logical -= sp * tiles;
        nk /= a.ksplit;
        a.A += (int64_t)sp * nk * BK;
        a.W += (int64_t)sp * nk * BK;
        a.C += (int64_t)sp * a.M * a.ldc;
    }
    tile_coords(tiles_m, tiles_n, logical, tile_m, tile_n);
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;
    if constexpr (DPH != 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, TG_ID = bits 16..19
        if (blockIdx.x < 512 && tile_base == 0 && ((hw >> 16) & 1u) != 0) {
            const unsigned long long t0 = wall_clock64();
            const int dt = g_gemm_dephase_ticks;
            while ((long long)(wall_clock64() - t0) < (long long)dt) __builtin_amdgcn_s_sleep(32);
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM * 32;
    const int wn0 = (wave % WN) * TN * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // ---- staging addresses.  Piece i of this thread covers tile row `row`, physical slot `ps`.
    // Rows past M / N are clamped to a valid row and NOT zeroed: an output element depends only on its own
    // A row and W row, and rows/cols past the edge are never stored, so the duplicate data is harmless.
    //   REG: thread f = tid + i*NT -> row f>>3, logical slot f&7, stored at slot (f&7) ^ swz(row)
    //   DMA: wave instruction q = wave + i*NW covers rows 8q..8q+7; lane -> row 8q + (lane>>3), physical slot
    //        lane&7, so it FETCHES logical slot (lane&7) ^ swz(row).
    const float* Ag[A_F4];
    const float* Wg[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        int row, cs;
        if constexpr (DMA) { row = (wave + i * NW) * 8 + (lane >> 3); cs = (lane & 7) ^ ((row >> 1) & 7); }
        else { const int f = tid + i * NT; row = f >> 3; cs = f & 7; }
        Ag[i] = a.A + (int64_t)min(bm0 + row, a.M - 1) * a.lda + cs * 4;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        int row, cs;
        if constexpr (DMA) { row = (wave + i * NW) * 8 + (lane >> 3); cs = (lane & 7) ^ ((row >> 1) & 7); }
        else { const int f = tid + i * NT; row = f >> 3; cs = f & 7; }
        Wg[i] = a.W + (int64_t)min(bn0 + row, a.N - 1) * a.ldw + cs * 4;
    }

    // DMA path: the same addresses as {uniform 64-bit base of the tile's first row, advanced by one K tile per iteration on
    // the SALU} + {per-lane 32-bit byte offset, constant over K}, i.e. the saddr form of global_load_lds.  The K loop then
    // has NO per-piece VALU address arithmetic: on gfx950 every VALU instruction issued between MFMAs takes 4-14 cycles away
    // from the matrix pipe even with a second wave resident (profiles/r1_mfma_valu_microbench.log).
    uint32_t Aoff[DMA ? A_F4 : 1], Woff[DMA ? B_F4 : 1];
    const char* Abase = reinterpret_cast<const char*>(a.A + (int64_t)bm0 * a.lda);
    const char* Wbase = reinterpret_cast<const char*>(a.W + (int64_t)bn0 * a.ldw);
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = (wave + i * NW) * 8 + (lane >> 3), cs = (lane & 7) ^ ((row >> 1) & 7);
            Aoff[i] = ((uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)a.lda + (uint32_t)cs * 4u) * 4u;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int row = (wave + i * NW) * 8 + (lane >> 3), cs = (lane & 7) ^ ((row >> 1) & 7);
            Woff[i] = ((uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)a.ldw + (uint32_t)cs * 4u) * 4u;
        }
    }

    // register staging path
    f32x4 ra[DMA ? 1 : A_F4], rb[DMA ? 1 : B_F4];
    auto load_global = [&](int kt) {
        if constexpr (!DMA) {
            const int k0 = kt * BK;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ag[i] + k0);
#pragma unroll
            for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wg[i] + k0);
        }
    };
    auto store_lds = [&](int buf) {
        if constexpr (!DMA) {
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
        }
    };
    // LDS-DMA path: one wave instruction moves 8 rows x 128 B = 1 KiB; LDS base is wave-uniform
    constexpr int NP = A_F4 + B_F4;     // DMA pieces per wave per K tile
    auto dma_piece = [&](int kt, int buf, int p) {      // p is a compile-time constant after unrolling
        if constexpr (DMA) {
            const int64_t k0b = (int64_t)kt * (BK * 4);      // wave-uniform: added to the base on the SALU
            if (p < A_F4)
                dma16_saddr(Abase + k0b, Aoff[p], lds_addr(As + (buf * BM + (wave + p * NW) * 8) * LDK));
            else
                dma16_saddr(Wbase + k0b, Woff[p - A_F4], lds_addr(Bs + (buf * BN + (wave + (p - A_F4) * NW) * 8) * LDK));
        }
    };
    auto dma_tile = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(kt, buf, p);
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

    // ---- prologue: tile 0 in LDS (REG: tile 1 already in flight to registers) ----
    if constexpr (DMA) {
        dma_tile(0, 0);
    } else {
        load_global(0);
        store_lds(0);
        load_global(min(1, nk - 1));
    }
    if constexpr (DMA) dma_wait_barrier();
    else __syncthreads();

    // one fragment (f < TM: A rows, else W rows) of k-group j into register slot `slot`
    auto read_one = [&](int buf, int j, int slot, int f) {
        if (f < TM)
            af[slot][f] = *reinterpret_cast<const f32x4*>(As + (buf * BM + wm0 + lrow + f * 32) * LDK + koff[j]);
        else
            bf[slot][f - TM] = *reinterpret_cast<const f32x4*>(Bs + (buf * BN + wn0 + lrow + (f - TM) * 32) * LDK + koff[j]);
    };
    constexpr int G = 4 * TM * TN;                                   // MFMAs per k-group
    constexpr int DS_AUTO = (NJ * G / 2) / NP >= 4 ? 4 : ((NJ * G / 2) / NP >= 1 ? (NJ * G / 2) / NP : 1);
    constexpr int DS = DSP > 0 ? DSP : DS_AUTO;                     // measured: spreading the copies 4 MFMAs apart +2-5 %
    constexpr int OFF0 = (NP < G - (TM + TN)) ? NP : G - (TM + TN);  // where the fragment prefetch starts in k-group 0
    static_assert(TM + TN <= G && NP * DS <= NJ * G, "tile too small for the staging interleave");
    // The per-tile barrier (this wave's copies of tile kt+1 landed + its LDS reads of tile kt returned) goes 32 MFMAs before
    // the end of the tile, but never before the tile's last staging instruction: the MFMAs after it only touch registers
    // and hide the barrier skew.  (A/B on the ViT shapes, profiles/r1_gemm_zero_valu_loop.log: 32 before the end is 0.3-1 %
    // faster than 8 or 15 before the end.)
    constexpr int LAST_READ = NJ >= 2 ? (NJ - 2) * G + (NJ == 2 ? OFF0 : 0) + TM + TN - 1 : 0;
    constexpr int LAST_DMA = (NP - 1) * DS;
    constexpr int LAST_STAGE = LAST_READ > LAST_DMA ? LAST_READ : LAST_DMA;
    constexpr int GBAR_WANT = NJ * G - 1 - (BARPOS > 0 ? BARPOS : 32);
    constexpr int GBAR = GBAR_WANT > LAST_STAGE ? GBAR_WANT : LAST_STAGE;
    static_assert(GBAR < NJ * G, "barrier position");

    // One K tile.  `bufc` is an integral constant on the DMA path (the loop below is unrolled by two), so every LDS address
    // of the tile is {per-lane base computed once before the loop} + {immediate offset}: no VALU address updates per tile.
    auto ktile = [&](int kt, auto bufc) {
        const int buf = bufc;
        if constexpr (DMA) {
            // Issue order pinned with sched_barrier fences (hipcc otherwise clusters the DMA copies up front and sinks
            // the fragment prefetch below the MFMAs it should hide under):
            //   frags(0) | (MFMA, DMA piece of tile kt+1) x NP | (MFMA, ds_read of frags(j+1)) x (TM+TN) | MFMAs ...
            // Tile kt+1 goes straight into the other buffer (last read in iteration kt-1, barrier passed); past the
            // last tile the copy re-reads the last tile into a buffer nobody reads (keeps the body branch-free).
            const int ktn = min(kt + 1, nk - 1);
            if constexpr (!(ABL & 4)) read_frags(buf, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) {
                            const int idx = (t * TM + mi) * TN + ni;          // compile-time after unrolling
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][mi][t], bf[j & 1][ni][t],
                                                                               acc[mi][ni], 0, 0, 0);
                            const int g = j * G + idx;                        // MFMA index within the K tile
                            const int ridx = idx - (j == 0 ? OFF0 : 0);       // slot in the fragment-prefetch run
                            const bool do_dma = (g % DS == 0) && (g / DS < NP);     // one DMA piece every DS-th MFMA
                            const bool do_read = j + 1 < NJ && ridx >= 0 && ridx < TM + TN;
                            if constexpr (!(ABL & 1)) { if (do_dma) dma_piece(ktn, buf ^ 1, g / DS); }
                            if constexpr (!(ABL & 4)) { if (do_read) read_one(buf, j + 1, (j + 1) & 1, ridx); }
                            if (do_dma || do_read) __builtin_amdgcn_sched_barrier(0);
                            if constexpr (!(ABL & 2)) {
                                if (g == GBAR) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    dma_wait_barrier();
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
            }
        } else {
            read_frags(buf, 0, 0);
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
                // staging in the MFMA shadow: registers -> LDS after the first k-group, next loads after the second
                if (j == 0) store_lds(buf ^ 1);
                if (j == 1) load_global(min(kt + 2, nk - 1));
            }
        }
        if constexpr (!(ABL & 2) && !DMA) __syncthreads();
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) asm volatile("" : "+v"(af[0][mi]), "+v"(af[1][mi]));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) asm volatile("" : "+v"(bf[0][ni]), "+v"(bf[1][ni]));
        }
    };
    if constexpr (DMA) {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(kt, IntC<0>{});
            ktile(kt + 1, IntC<1>{});
        }
        if (kt < nk) ktile(kt, IntC<0>{});
    } else {
        for (int kt = 0; kt < nk; ++kt) ktile(kt, kt & 1);
    }

    if constexpr (EPI != EPI_BIAS_POS && EPI != EPI_BIAS_QSCALE) {
        if (a.cs_out) {                     // wave-uniform
            store_tile_scatter<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
            return;
        }
    }
    store_tile<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
}

// ---------------------------------------------------------------------------------------------------------------------
// Small-M GEMM (few crops: M = 192*B with B <~ 8).  The big-tile kernel above leaves most of the 256 CUs idle there and
// its 2-buffer pipeline pays one full memory round trip per 32-deep K tile when only one block sits on a CU
// (measured 1.1 us per K tile at B = 1: 45 us for K = 1280, 157 us for K = 5120 — profiles/r1_small_batch_classes.log).
//   * 64x64 tiles (2x2 waves of 32x32), ST-deep LDS ring fed by global_load_lds: the copy of K tile kt+ST-1 is issued
//     while tile kt is multiplied, completion is tracked with s_waitcnt vmcnt(N) (LDS-DMA returns in order), and the
//     ONE barrier per K tile sits in the middle of the tile's MFMAs so it is covered by queued matrix work.
//   * split-K: `ksplit` blocks share an output tile, each reducing K/ksplit, so the N = 1280 GEMMs (60 tiles at B = 1)
//     still occupy every CU.  PARTIAL blocks write raw fp32 partial tiles to part[ksplit][M][N]; they are summed in
//     a FIXED order (deterministic) by splitk_* kernels (rowops.hip), which also apply the epilogue and, in the engine,
//     the LayerNorm that follows proj / fc2 anyway — so split-K adds no launch.
//   * block -> work map: (column tile, K slice) pairs are dealt round-robin to the 8 XCDs and the tiles_m row tiles that
//     share that W slice run on the same XCD, so W is fetched from HBM once and re-read from that XCD's L2.
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ABL (timing-only experiments of scripts/micro/ring_ablation.hip, garbage results): bit0 no copies inside the K loop, bit1 no
// per-tile wait + barrier, bit2 no LDS fragment reads.  ABL = 0 is the product kernel.
template <int ST, int EPI, bool PARTIAL, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs a, int tiles_m, int groups, int ksplit, float* part) {
    constexpr int BM = 64, BN = 64, NP = 4;       // NP = DMA wave-instructions per wave per K tile (2 for A, 2 for W)
    static_assert((ST & (ST - 1)) == 0 && ST >= 4, "ring depth must be a power of two >= 4");
    __shared__ __attribute__((aligned(16))) float smem[ST * (BM + BN) * LDK];
    float* As = smem;                       // [ST][BM][LDK]
    float* Bs = smem + ST * BM * LDK;       // [ST][BN][LDK]

    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int gl = within / tiles_m, tile_m = within - gl * tiles_m;
    const int g = gl * 8 + xcd;
    if (g >= groups) return;
    const int tile_n = g / ksplit, sp = g - tile_n * ksplit;
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;
    const int kper = a.K / ksplit, kbeg = sp * kper, nk = kper / BK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // wave instruction q = wave + 4p covers tile rows 8q..8q+7 (see the DMA notes of gemm_f32_kernel).  As there, the copies use
    // the saddr form (uniform base + constant 32-bit lane offset) and the ring slot of every tile is a compile-time constant
    // (the K loop is unrolled by ST), so the K loop contains no VALU instruction at all: with 16 MFMAs per tile the 17 address
    // VALU instructions of the first version cost >10 % of the matrix pipe (profiles/r1_mfma_valu_microbench.log).
    uint32_t Aoff[2], Woff[2];
    const char* Abase = reinterpret_cast<const char*>(a.A + (int64_t)bm0 * a.lda + kbeg);
    const char* Wbase = reinterpret_cast<const char*>(a.W + (int64_t)bn0 * a.ldw + kbeg);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave + 4 * p) * 8 + (lane >> 3), cs = (lane & 7) ^ ((row >> 1) & 7);
        Aoff[p] = ((uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)a.lda + (uint32_t)cs * 4u) * 4u;
        Woff[p] = ((uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)a.ldw + (uint32_t)cs * 4u) * 4u;
    }
    // ABL bit3 (experiment): column tile tile_n starts its K sweep at K tile (tile_n mod nk) and wraps around, so that concurrently
    // running blocks do not all fetch the same 128-byte column of A / W rows (row stride K * 4 B) at the same time
    const int krot = (ABL & 8) ? tile_n % nk : 0;
    auto dma_tile = [&](int kt, auto stc) {      // K tile kt -> ring slot stc (= kt % ST, an integral constant)
        const int st = stc;
        int kk = kt + krot;
        if (kk >= nk) kk -= nk;
        const int64_t k0b = (int64_t)kk * (BK * 4);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dma16_saddr(Abase + k0b, Aoff[p], lds_addr(As + (st * BM + (wave + 4 * p) * 8) * LDK));
            dma16_saddr(Wbase + k0b, Woff[p], lds_addr(Bs + (st * BN + (wave + 4 * p) * 8) * LDK));
        }
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    int koff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) koff[j] = (((2 * j + lhalf) ^ ((lrow >> 1) & 7)) << 2);
    f32x4 af[2], bf[2];
    auto read_frags = [&](auto stc, int j, int slot) {
        const int st = stc;
        if constexpr ((ABL & 4) != 0) {
            asm volatile("" : "+v"(af[slot]), "+v"(bf[slot]));
            return;
        }
        af[slot] = *reinterpret_cast<const f32x4*>(As + (st * BM + wm0 + lrow) * LDK + koff[j]);
        bf[slot] = *reinterpret_cast<const f32x4*>(Bs + (st * BN + wn0 + lrow) * LDK + koff[j]);
    };
    // one k-group: MFMA 0, then the fragment reads of the NEXT group (nxt() may be empty), then MFMAs 1-3.  The order is
    // pinned with sched_barrier fences: the reads must not precede MFMA 0 (hipcc waits lgkmcnt(0) before it, which would then
    // also wait for the reads just issued) and must not sink below the MFMAs that hide their latency.
    auto group = [&](int slot, auto nxt) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][0], bf[slot][0], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        nxt();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][t], bf[slot][t], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: K tiles 0..ST-2 in flight; tile 0 must have landed (for every wave) before the first fragment read
    static_for<ST - 1>([&](auto t) {
        if (t < nk) dma_tile(t, t);
    });
    if (nk >= ST - 1) wait_vm_barrier<(ST - 2) * NP>(); else wait_vm_barrier<0>();
    read_frags(IntC<0>{}, 0, 0);

    auto tile = [&](int kt, auto stc, bool last) {      // stc = kt % ST; `last` is a literal at every call site
        constexpr int S = decltype(stc){};
        group(0, [&] { read_frags(stc, 1, 1); });
        group(1, [&] { read_frags(stc, 2, 0); });
        if (!last) {
            // tile kt+1 landed (in-order completion: at most the ST-3 younger tiles may still be in flight); after the
            // barrier every wave is past tile kt-1, whose ring slot receives tile kt+ST-1
            if constexpr ((ABL & 2) == 0) {
                if (kt + ST - 2 < nk) wait_vm_barrier<(ST - 3) * NP>(); else wait_vm_barrier<0>();
            }
            if constexpr ((ABL & 1) == 0) {
                if (kt + ST - 1 < nk) dma_tile(kt + ST - 1, IntC<(S + ST - 1) % ST>{});
            }
        }
        group(0, [&] { read_frags(stc, 3, 1); });
        group(1, [&] { if (!last) read_frags(IntC<(S + 1) % ST>{}, 0, 0); });
    };
    int kt = 0;
    for (; kt + ST <= nk - 1; kt += ST) static_for<ST>([&](auto sc) { tile(kt + sc, sc, false); });
    static_for<ST>([&](auto sc) {            // < ST tiles left before the last one; kt is a multiple of ST
        if (kt + sc < nk - 1) tile(kt + sc, sc, false);
    });
    static_for<ST>([&](auto sc) {
        if (((nk - 1) & (ST - 1)) == sc) tile(nk - 1, sc, true);
    });

    if constexpr (PARTIAL) {
        GemmArgs pa = a;
        pa.C = part + (int64_t)sp * a.M * a.N;
        pa.ldc = a.N;
        store_tile<1, 1, EPI_NONE>(pa, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
    } else {
        if constexpr (EPI != EPI_BIAS_POS && EPI != EPI_BIAS_QSCALE) {
            if (a.cs_out) {
                store_tile_scatter<1, 1, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
                return;
            }
        }
        store_tile<1, 1, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
    }
}

template <int ST>
int launch_ring(const GemmArgs& a, int epi, int ksplit, float* part, hipStream_t s) {
    const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 63) / 64;
    const int groups = tiles_n * ksplit;
    dim3 grid(8 * tiles_m * ((groups + 7) / 8)), block(256);
    if (ksplit > 1) {
        hipLaunchKernelGGL((gemm_ring_kernel<ST, EPI_NONE, true>), grid, block, 0, s, a, tiles_m, groups, ksplit, part);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
#define THMR_RING_CASE(E)                                                                                           \
    case E:                                                                                                         \
        hipLaunchKernelGGL((gemm_ring_kernel<ST, E, false>), grid, block, 0, s, a, tiles_m, groups, 1, nullptr);     \
        break;
    switch (epi) {
        THMR_RING_CASE(EPI_NONE)
        THMR_RING_CASE(EPI_BIAS)
        THMR_RING_CASE(EPI_BIAS_GELU)
        THMR_RING_CASE(EPI_BIAS_RELU)
        THMR_RING_CASE(EPI_BIAS_RESID)
        THMR_RING_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_RING_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------------------
// Small-M GEMM on 16x16x4 MFMA tiles, for the QKV projection of one and two crops (round 3).
// The ring kernel above gives the (192, 3840) x 1280 product of one crop 3 x 60 = 180 workgroups = 720 waves for 1024 SIMDs, and a
// wave's 32x32 tile is ONE dependent chain of 640 MFMAs of 64 cycles = 17.1 us of the launch's 25 us.  With v_mfma_f32_16x16x4_f32
// (the same 64 flop per cycle) the tile granularity is 16: a workgroup tile of 64 x 48 (4 waves stacked in M, each 16 rows x three
// 16x16 column tiles) makes 3 x 80 = 240 workgroups = 960 waves, each with 40 K tiles x 24 MFMAs x 32 cycles = 12.8 us.
// Same staging as the ring kernel (ST-deep LDS ring fed by global_load_lds in the saddr form, identical swizzled LDS image,
// vmcnt-tracked completion, one barrier per K tile in the middle of its MFMAs); the 14 copies of a K tile (8 for A, 6 for W) are
// dealt 4 / 4 / 3 / 3 to the waves, so the vmcnt bookkeeping is instantiated for both counts behind a wave-uniform branch.
// K is summed in a different order than by the 32x32x2 kernels (per 16-k group: MFMA t adds k0 + t, k0 + 4 + t, k0 + 8 + t,
// k0 + 12 + t), so it serves a layer at EVERY batch size of a regime or not at all: the engine uses it for qkv in its
// one-and-two-crop regime (kKeysplitMaxB).
template <int ST, int EPI>
__global__ __launch_bounds__(256) void gemm_ring16_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int BM = 64, BN = 48;
    static_assert((ST & (ST - 1)) == 0 && ST >= 4, "ring depth must be a power of two >= 4");
    __shared__ __attribute__((aligned(16))) float smem[ST * (BM + BN) * LDK];
    float* As = smem;                       // [ST][BM][LDK]
    float* Bs = smem + ST * BM * LDK;       // [ST][BN][LDK]
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int gl = within / tiles_m, tile_m = within - gl * tiles_m;
    const int tile_n = gl * 8 + xcd;
    if (tile_n >= tiles_n) return;
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;
    const int nk = a.K / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const bool two_w = wave < 2;            // waves 0, 1 also copy W rows 32 .. 47

    uint32_t Aoff[2], Woff[2];
    const char* Abase = reinterpret_cast<const char*>(a.A + (int64_t)bm0 * a.lda);
    const char* Wbase = reinterpret_cast<const char*>(a.W + (int64_t)bn0 * a.ldw);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave + 4 * p) * 8 + (lane >> 3), cs = (lane & 7) ^ ((row >> 1) & 7);
        Aoff[p] = ((uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)a.lda + (uint32_t)cs * 4u) * 4u;
        Woff[p] = ((uint32_t)(min(bn0 + min(row, BN - 1), a.N - 1) - bn0) * (uint32_t)a.ldw + (uint32_t)cs * 4u) * 4u;
    }
    auto dma_tile = [&](int kt, auto stc) {      // K tile kt -> ring slot stc (an integral constant)
        const int st = stc;
        const int64_t k0b = (int64_t)kt * (BK * 4);
        dma16_saddr(Abase + k0b, Aoff[0], lds_addr(As + (st * BM + wave * 8) * LDK));
        dma16_saddr(Abase + k0b, Aoff[1], lds_addr(As + (st * BM + (wave + 4) * 8) * LDK));
        dma16_saddr(Wbase + k0b, Woff[0], lds_addr(Bs + (st * BN + wave * 8) * LDK));
        if (two_w) dma16_saddr(Wbase + k0b, Woff[1], lds_addr(Bs + (st * BN + (wave + 4) * 8) * LDK));
    };

    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment of k-group j2 (16 k): lane (row l15, k slot g) reads the 16 bytes at logical slot 4 j2 + g of its row; the swizzle term
    // (row >> 1) & 7 only depends on l15 because the row tiles start at multiples of 16
    int koff[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) koff[j2] = (((4 * j2 + g) ^ ((l15 >> 1) & 7)) << 2);
    f32x4 af[2], bf[2][3];
    auto read_frags = [&](auto stc, int j2, int slot) {
        const int st = stc;
        af[slot] = *reinterpret_cast<const f32x4*>(As + (st * BM + 16 * wave + l15) * LDK + koff[j2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) bf[slot][c] = *reinterpret_cast<const f32x4*>(Bs + (st * BN + 16 * c + l15) * LDK + koff[j2]);
    };
    // one k-group = 12 MFMAs: the first, then the fragment reads of the NEXT group, then the other eleven (order pinned as in the ring kernel)
    auto group = [&](int slot, auto nxt) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[slot][0], bf[slot][0][0], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        nxt();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (t != 0 || c != 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[slot][t], bf[slot][c][t], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto npc) {              // NPW = copies per K tile of THIS wave (4 or 3): the vmcnt immediates depend on it
        constexpr int NPW = decltype(npc){};
        static_for<ST - 1>([&](auto t) {
            if (t < nk) dma_tile(t, t);
        });
        if (nk >= ST - 1) wait_vm_barrier<(ST - 2) * NPW>(); else wait_vm_barrier<0>();
        read_frags(IntC<0>{}, 0, 0);
        auto tile = [&](int kt, auto stc, bool last) {      // stc = kt % ST; `last` is a literal at every call site
            constexpr int S = decltype(stc){};
            group(0, [&] { read_frags(stc, 1, 1); });
            if (!last) {
                // tile kt+1 landed (in-order completion); after the barrier every wave is past tile kt-1, whose slot receives tile kt+ST-1
                if (kt + ST - 2 < nk) wait_vm_barrier<(ST - 3) * NPW>(); else wait_vm_barrier<0>();
                if (kt + ST - 1 < nk) dma_tile(kt + ST - 1, IntC<(S + ST - 1) % ST>{});
            }
            group(1, [&] { if (!last) read_frags(IntC<(S + 1) % ST>{}, 0, 0); });
        };
        int kt = 0;
        for (; kt + ST <= nk - 1; kt += ST) static_for<ST>([&](auto sc) { tile(kt + sc, sc, false); });
        static_for<ST>([&](auto sc) {
            if (kt + sc < nk - 1) tile(kt + sc, sc, false);
        });
        static_for<ST>([&](auto sc) {
            if (((nk - 1) & (ST - 1)) == sc) tile(nk - 1, sc, true);
        });
    };
    if (two_w) run(IntC<4>{}); else run(IntC<3>{});

    // C/D layout of 16x16: col = lane & 15, row = 4 * (lane >> 4) + i
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int n = bn0 + 16 * c + l15;
        if (n >= a.N) continue;
        float bias = 0.f;
        if constexpr (EPI != EPI_NONE) bias = a.bias[n];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = bm0 + 16 * wave + 4 * g + i;
            if (m >= a.M) continue;
            float v = acc[c][i] + bias;
            if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
            if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            if constexpr (EPI == EPI_BIAS_QSCALE) v = (n < a.qcols) ? v * a.qscale : v;
            a.C[(int64_t)m * a.ldc + n] = v;
        }
    }
}

template <int ST>
int launch_ring16_st(const GemmArgs& a, int epi, hipStream_t s) {
    const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 47) / 48;
    dim3 grid(8 * tiles_m * ((tiles_n + 7) / 8)), block(256);
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemm_ring16_kernel<ST, EPI_NONE>), grid, block, 0, s, a, tiles_m, tiles_n); break;
        case EPI_BIAS: hipLaunchKernelGGL((gemm_ring16_kernel<ST, EPI_BIAS>), grid, block, 0, s, a, tiles_m, tiles_n); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_ring16_kernel<ST, EPI_BIAS_GELU>), grid, block, 0, s, a, tiles_m, tiles_n); break;
        case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_ring16_kernel<ST, EPI_BIAS_RELU>), grid, block, 0, s, a, tiles_m, tiles_n); break;
        case EPI_BIAS_QSCALE: hipLaunchKernelGGL((gemm_ring16_kernel<ST, EPI_BIAS_QSCALE>), grid, block, 0, s, a, tiles_m, tiles_n); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ring depth: a K tile is only 24 MFMAs x 32 cycles = 0.32 us of matrix work per wave, so the 3 tiles in flight of ST = 4 cover less
// than one HBM round trip; ST = 8 (112 KB of LDS) covers 2.2 us but allows ONE workgroup per CU: measured -0.3 % per call at one crop
// (240 workgroups) and +4 % at two (480 workgroups want two per CU), profiles/r3m_ring16_depth_ab.log — ST = 4 it is
// (THMR_RING16_DEPTH=8 for A/B; same arithmetic, bit-identical).
int launch_ring16(const GemmArgs& a, int epi, hipStream_t s) {
    static const int depth = [] { const char* e = thmr_knob("THMR_RING16_DEPTH"); return e ? atoi(e) : 4; }();
    return depth == 8 ? launch_ring16_st<8>(a, epi, s) : launch_ring16_st<4>(a, epi, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiny-M GEMM for the head's dependent chain at up to six crops (soft codebook lookup + the VQ decoder's Conv1d GEMMs: M = 21 ... 160
// rows per crop, N = 256 / 512, K = 512 ... 2048).  The ring kernel above gives such a product 8-24 workgroups that each walk the
// WHOLE K: 24-64 K tiles at ~0.5 us = 12-32 us per launch, 13 launches in a row = a third of the head at one crop
// (profiles/r2s_head_b1_kernel_stats.csv).  Here a workgroup owns a 32x32 tile and its EIGHT waves split K eight ways:
//   * no LDS staging: wave w's K slice is used by nobody else, so its A / W fragments go global -> VGPR (one 16-byte load per lane =
//     4 consecutive k of one row: exactly the k-permuted MFMA operand), four k-groups ahead;
//   * the eight partial tiles are added through LDS in wave order (fixed association: deterministic, independent of M);
//   * wave 0 applies the same epilogues as the big kernels, including the im2col scatter for the next Conv1d.
// A K slice is K / 8 (a multiple of 32).  The association of the K sum differs from the other kernels', so the engine uses this
// kernel for a given layer at EVERY batch size of the small-batch regime (B <= 6) or not at all.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_tiny_kernel(GemmArgs a, int tiles_n) {
    constexpr int NWV = 8, PF = 4;
    __shared__ __attribute__((aligned(16))) float red[NWV][16][64];     // [wave][accumulator register][lane]
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int kper = a.K / NWV, ngrp = kper / 8;                        // 8-deep k-groups of this wave's slice; ngrp % PF == 0
    const float* ap = a.A + (int64_t)min(m0 + lrow, a.M - 1) * a.lda + wave * kper + 4 * lhalf;
    const float* wp = a.W + (int64_t)min(n0 + lrow, a.N - 1) * a.ldw + wave * kper + 4 * lhalf;
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    f32x4 af[PF], bf[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        af[j] = *reinterpret_cast<const f32x4*>(ap + j * 8);
        bf[j] = *reinterpret_cast<const f32x4*>(wp + j * 8);
    }
#pragma unroll 1
    for (int g0 = 0; g0 < ngrp; g0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t], bf[j][t], acc[0][0], 0, 0, 0);
            const int gn = min(g0 + j + PF, ngrp - 1);                  // past the end: re-fetch the last group (never used)
            af[j] = *reinterpret_cast<const f32x4*>(ap + gn * 8);
            bf[j] = *reinterpret_cast<const f32x4*>(wp + gn * 8);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[0][0][e];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float v = red[0][e][lane];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += red[w][e][lane];
        acc[0][0][e] = v;
    }
    if constexpr (EPI != EPI_BIAS_POS && EPI != EPI_BIAS_QSCALE) {
        if (a.cs_out) {
            store_tile_scatter<1, 1, EPI>(a, acc, m0, n0, lrow, lhalf);
            return;
        }
    }
    store_tile<1, 1, EPI>(a, acc, m0, n0, lrow, lhalf);
}

int launch_tiny(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.K % 256 != 0) return -1;
    const int tiles_m = (a.M + 31) / 32, tiles_n = (a.N + 31) / 32;
    dim3 grid(tiles_m * tiles_n), block(512);
#define THMR_TINY_CASE(E)                                                              \
    case E:                                                                            \
        hipLaunchKernelGGL((gemm_tiny_kernel<E>), grid, block, 0, s, a, tiles_n);      \
        break;
    switch (epi) {
        THMR_TINY_CASE(EPI_NONE)
        THMR_TINY_CASE(EPI_BIAS)
        THMR_TINY_CASE(EPI_BIAS_GELU)
        THMR_TINY_CASE(EPI_BIAS_RELU)
        THMR_TINY_CASE(EPI_BIAS_RESID)
        THMR_TINY_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_TINY_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The trailing part of a grid that does not fill the 512 resident slots (2 blocks per CU) a whole number of times.  Up to 256
// trailing tiles are best run ONE per CU (a block that is alone on its CU runs almost twice as fast: 0.5 round), but inside one
// launch the dispatcher hands them to whichever slots free up first, and the two co-resident blocks of a CU retire together: that
// CU takes two trailing tiles while another idles (scripts/micro/dispatch_map.hip).  Whether it happens is chaotic — the 768-tile
// proj grid at 64 crops ran 287 us in every build of round 1 and 375 us (2 full rounds) after an unrelated change to the epilogue
// code of round 2 (profiles/r2k_proj_tail_dispatch.log), the bimodal B = 16 fc1 of round 1 is the same thing.  So such a tail is
// launched SEPARATELY with 16 KB of (unused) dynamic LDS on top of the kernel's 72 KB: two of those blocks cannot share a CU, the
// dispatcher must spread them one per CU.  Same tiles, same arithmetic, results unchanged.
constexpr int kSlots = 512, kTailLds = 16 * 1024;

template <int WM, int WN, int TM, int TN, bool DMA>
int launch_cfg(const GemmArgs& a, int epi, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n * (a.ksplit > 1 ? a.ksplit : 1);     // split-K: ksplit copies of the tile grid in one launch
    // separate tail launch only for the 128-row LDS-DMA tiles whose static LDS (> 64 KB) + 16 KB exceeds half a CU
    constexpr bool kCanIsolate = DMA && 2 * (BM + BN) * LDK * 4 + kTailLds > 80 * 1024;
    const int tail = tiles % kSlots;
    const bool isolate = kCanIsolate && tiles > kSlots && tail > 0 && tail <= kSlots / 2;
    const int main_tiles = isolate ? tiles - tail : tiles;
    dim3 block(WM * WN * 64);
#define THMR_GEMM_CASE(E)                                                                                                               \
    case E:                                                                                                                             \
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, DMA, E>), dim3(main_tiles), block, 0, s, a, tiles_m, tiles_n, main_tiles, 0); \
        if (isolate)                                                                                                                    \
            hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, DMA, E>), dim3(tail), block, kTailLds, s, a, tiles_m, tiles_n, tail,      \
                               main_tiles);                                                                                             \
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

#ifdef THMR_GEMM_ABLATION   // timing-only experiment kernels (garbage results); not built into the product library
template <int ABL, int DS = 0, int BARPOS = 0>
int launch_abl(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + 127) / 128, tiles_n = (a.N + 159) / 160;
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1, 5, true, EPI_NONE, ABL, DS, BARPOS>), dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_m, tiles_n, tiles_m * tiles_n, 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#endif

}  // namespace

// variant: 0 = 128x128 REG (2x2 waves of 64x64)   1 = 128x160 REG (4x1 waves of 32x160)
//          7 = 128x128 DMA                         8 = 128x160 DMA            9 = 64x64 DMA (2x2 waves of 32x32)
//         10 = 128x96 DMA (4x1 waves of 32x96)          11 = 32x32 tiny-M kernel (K split over 8 waves, no LDS staging)
//         12 = 64x128 DMA (2x2 waves of 32x64)          13 = 128x64 DMA (4x1 waves of 32x64)   [round 3]
//         -1 = DMA, tile picked by a cost model over 256 CUs   (2 is the skinny kernel, see thmr_op_gemm)
int launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % BK) != 0) return -1;
    if ((a.lda % 4) != 0 || (a.ldw % 4) != 0) return -1;
    if (a.lda >= (1 << 22) || a.ldw >= (1 << 22)) return -1;     // per-lane byte offsets within a tile are 32-bit
    if (variant < 0) {
        // relative time ~ (rounds over 256 CUs) x tile area / tile efficiency.  A half-filled last round costs about one
        // tile time, not two (a block that is alone on its CU runs almost twice as fast), so rounds are counted over 256 CUs
        // although 512 blocks are resident.  Since the K loops carry no VALU work all tiles have nearly the same per-area
        // efficiency (fitted on profiles/r1_tile_sweep.log: B = 7 ... 64 on the four ViT shapes) and the choice is mostly tile
        // quantisation; small grids (the VQ decoder's M = B*21 convs) end up on the 64x64 tile.
        // Round 3 experiment (THMR_ALONE_PENALTY, default 1.0 = off): a penalty for grids of at most 256 blocks (one block per CU, where
        // nothing hides the 2-buffer pipeline's memory round trip per K tile).  It is right for the split-K launcher below, but as a
        // general rule it is wrong: 1.1 gains 1.6 % per call at 3 crops and LOSES 3 % at 5-6 and 9 % at 20 crops (fc2's 240 tiles of
        // 128x160 with their 160-tile K loops are the best choice there) — profiles/r3o_alone_penalty_ab.log.
        static const double alone = [] { const char* e = thmr_knob("THMR_ALONE_PENALTY"); return e ? atof(e) : 1.0; }();
        auto cost = [&](int BM, int BN, double eff) {
            const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
            return (double)((tiles + 255) / 256) * BM * BN / eff * (tiles <= 256 ? alone : 1.0);
        };
        const double c7 = cost(128, 128, 0.99), c8 = cost(128, 160, 1.0), c9 = cost(64, 64, 0.95), c10 = cost(128, 96, 0.985);
        variant = 8;
        double best = c8;
        if (c7 < best) { best = c7; variant = 7; }
        if (c10 < best) { best = c10; variant = 10; }
        if (c9 < best) { best = c9; variant = 9; }
        // (64x128 / 128x64, variants 12 / 13, were tried here in round 3: on the N = 1280 GEMMs at 7-8 crops they make 210-240 tiles
        // that run ONE per CU.  Stand-alone with warm weights that looked 4-6 % faster than 480 tiles of 64x64; in the pipeline, with
        // the weights streamed cold from HBM, a block that is alone on its CU pays a memory round trip per K tile and the call got
        // 7 % SLOWER — profiles/r3f_n1280_tile_sweep.log vs r3g_mid_batch_tiles_in_cost_model.log.  They serve the split-K launcher.)
        const long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        // at most one 64x64 block per CU: nothing hides the 2-buffer kernel's per-K-tile memory round trip, so the 4-deep
        // ring version of the same tile is used (same K order, bit-identical; measured 46 -> ~30 us on the head's B = 1 convs).
        // NOT beyond 256 tiles: with two blocks per CU the 2-buffer kernel is the faster one (fc2 at 8 crops, 480 tiles: 176 us
        // against 229 us on the ring kernel — a threshold of 512 was tried in round 2 and cost B = 8 11 %, profiles/r2ad_batch_sweep.jsonl)
        static const long ring_max_tiles = [] { const char* e = thmr_knob("THMR_RING_MAX_TILES"); return e ? atol(e) : 256L; }();   // A/B knob
        if (variant == 9 && tiles64 <= ring_max_tiles && epi != EPI_BIAS_POS) return launch_ring<4>(a, epi, 1, nullptr, s);
    }
#ifdef THMR_GEMM_ABLATION
    switch (variant) {     // 30 + ABL: timing-only ablations of the 128x160 DMA kernel (EPI_NONE)
        case 31: return launch_abl<1>(a, s);
        case 32: return launch_abl<2>(a, s);
        case 33: return launch_abl<3>(a, s);
        case 34: return launch_abl<4>(a, s);
        case 37: return launch_abl<7>(a, s);
        case 41: return launch_abl<0, 1>(a, s);    // DMA piece every MFMA / 2nd / 4th MFMA
        case 42: return launch_abl<0, 2>(a, s);
        case 44: return launch_abl<0, 4>(a, s);
        case 51: return launch_abl<0, 4, 10>(a, s);   // barrier pinned 10 / 20 / 2 MFMAs before the end of the tile
        case 52: return launch_abl<0, 4, 20>(a, s);
        case 53: return launch_abl<0, 4, 2>(a, s);
        default: break;
    }
#endif
    switch (variant) {
        case 11: return launch_tiny(a, epi, s);                      // 32x32 tiles, K split over the 8 waves of a workgroup
        case 0: return launch_cfg<2, 2, 2, 2, false>(a, epi, s);
        case 1: return launch_cfg<4, 1, 1, 5, false>(a, epi, s);
        case 8: return launch_cfg<4, 1, 1, 5, true>(a, epi, s);
        case 9: return launch_cfg<2, 2, 1, 1, true>(a, epi, s);
        case 10: return launch_cfg<4, 1, 1, 3, true>(a, epi, s);     // 128x96
        case 12: return launch_cfg<2, 2, 1, 2, true>(a, epi, s);     // 64x128 (2x2 waves of 32x64)
        case 13: return launch_cfg<4, 1, 1, 2, true>(a, epi, s);     // 128x64 (4x1 waves of 32x64)
        default: return launch_cfg<2, 2, 2, 2, true>(a, epi, s);
    }
}

// Split-K on the big LDS-DMA tiles (mid-size batches: the N = 1280 GEMMs of 7 ... ~23 crops have only 120-480 output tiles of
// 128x128, so proj / fc2 either ran on the 64x64 tile (0.82 of the big tiles' per-area rate) or left half the resident slots
// empty).  `ksplit` copies of the tile grid in ONE launch, copy sp reducing K slice sp into part[sp] (raw fp32 partial tiles,
// [ksplit][M][N]); the partials are summed in a fixed order by the residual + LayerNorm kernel that follows proj / fc2 anyway
// (launch_splitk_resid_ln), so split-K adds no launch.  Within a slice K is summed in the same order by every tile variant.
int launch_gemm_splitk(const GemmArgs& a0, int variant, int ksplit, float* part, hipStream_t s) {
    if (ksplit < 2 || part == nullptr) return -1;
    if (a0.M <= 0 || a0.N <= 0 || a0.K <= 0 || (a0.K % (BK * ksplit)) != 0) return -1;
    if ((a0.lda % 4) != 0 || (a0.ldw % 4) != 0 || a0.lda >= (1 << 22) || a0.ldw >= (1 << 22)) return -1;
    GemmArgs a = a0;
    a.ksplit = ksplit;
    a.C = part; a.ldc = a.N;
    a.bias = nullptr; a.resid = nullptr; a.ldr = 0; a.cs_out = nullptr;
    static const int forced_tile = [] { const char* e = thmr_knob("THMR_MID_TILE"); return e ? atoi(e) : -1; }();     // A/B knob
    if (variant < 0 && forced_tile > 0) variant = forced_tile;
    if (variant < 0) {
        // as launch_gemm's model, plus: a grid of at most 256 blocks runs one block per CU, where nothing hides the 2-buffer
        // pipeline's memory round trip per K tile (weights come cold from HBM in the pipeline): +10 %; and the 64x128 tile
        // (0.95), which turns the 210-240 tiles of 7-8 crops into 420-480 co-resident pairs
        auto cost = [&](int BM, int BN, double eff) {
            const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * ksplit;
            return (double)((tiles + 255) / 256) * BM * BN / eff * (tiles <= 256 ? 1.1 : 1.0);
        };
        const double c7 = cost(128, 128, 0.99), c8 = cost(128, 160, 1.0), c10 = cost(128, 96, 0.985), c12 = cost(64, 128, 0.95);
        variant = 8;
        double best = c8;
        if (c7 < best) { best = c7; variant = 7; }
        if (c10 < best) { best = c10; variant = 10; }
        if (c12 < best) { best = c12; variant = 12; }
    }
    switch (variant) {
        case 8: return launch_cfg<4, 1, 1, 5, true>(a, EPI_NONE, s);
        case 9: return launch_cfg<2, 2, 1, 1, true>(a, EPI_NONE, s);
        case 10: return launch_cfg<4, 1, 1, 3, true>(a, EPI_NONE, s);
        case 12: return launch_cfg<2, 2, 1, 2, true>(a, EPI_NONE, s);
        case 13: return launch_cfg<4, 1, 1, 2, true>(a, EPI_NONE, s);
        case 7: return launch_cfg<2, 2, 2, 2, true>(a, EPI_NONE, s);
        default: return -1;
    }
}

// small-M GEMM on 16x16x4 tiles (64 x 48 workgroup tile); epilogues none / bias / gelu / relu / qscale
int launch_gemm_ring16(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % BK) != 0) return -1;
    if ((a.lda % 4) != 0 || (a.ldw % 4) != 0 || a.lda >= (1 << 22) || a.ldw >= (1 << 22)) return -1;
    return launch_ring16(a, epi, s);
}

// ring = 4 | 8 (LDS ring depth: 64 KB -> 2 blocks/CU, 128 KB -> 1 block/CU with twice the prefetch distance);
// ksplit > 1 writes raw partial sums to part[ksplit][M][N] and applies NO epilogue (see launch_splitk_*).
int launch_gemm_ring(const GemmArgs& a, int epi, int ring, int ksplit, float* part, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || ksplit < 1 || (a.K % (BK * ksplit)) != 0) return -1;
    if ((a.lda % 4) != 0 || (a.ldw % 4) != 0 || (ksplit > 1 && part == nullptr)) return -1;
    if (a.lda >= (1 << 22) || a.ldw >= (1 << 22)) return -1;
    return ring == 8 ? launch_ring<8>(a, epi, ksplit, part, s) : launch_ring<4>(a, epi, ksplit, part, s);
}
