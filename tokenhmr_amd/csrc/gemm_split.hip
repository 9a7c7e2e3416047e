// fp32 GEMM on the bf16 matrix pipe:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)  with every fp32 operand carried as THREE bf16 pieces.
//
// Why: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, gemm_f32.hip) runs at 1/16 of the bf16 rate on gfx950 (157 vs 2500 TFLOP/s) and the
// ViT GEMMs of the reference (tokenhmr/lib/models/backbones/vit.py:82-87,104-126) already sit at 0.93 of that peak.  An fp32 number
// is the exact sum of three bf16 numbers, x = h + m + l (8 + 8 + 8 significand bits, each piece the round-to-nearest bf16 of what
// the pieces before it left), a product of two bf16 is exact in fp32, and
//     a.b = hh + (hm + mh) + (mm + hl + lh) + [ml + lm + ll]
// where the bracket is below 2^-23 of |a.b| — the size of ONE fp32 rounding, which an fp32 dot product of length K commits K times.
// So six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per 16 k give an fp32-grade product at 16 / 6 = 2.67 times the matrix rate.
// This is NOT the product path of the engine (which stays on exact-fp32 MFMA and is bitwise an fmaf chain); it is an op of the C ABI
// (thmr_op_split3 / thmr_op_gemm_split3) with its own parity tests against an fp64 reference, measured beside the fp32 kernel.
//
// Operand format ("split3"): X[R][K] fp32  ->  Xs[R][K/8][3][8] bf16: 8 consecutive k of a row as three adjacent 16-byte chunks
// (piece h, m, l).  One chunk is exactly one lane's A / B operand of v_mfma_f32_32x32x16_bf16 (8 consecutive k of one row), a row
// of a 32-deep K tile is 192 contiguous bytes, a row of the matrix is 6 K bytes.
//
// Kernel (gfx950): block tile (WM TM 32) x (WN TN 32), 8 waves of 64 x 64 on 128 x 256 by default, ONE workgroup per CU (two LDS
// stages of 72 KB), two waves per SIMD.
//   * staging: global_load_lds_dwordx4, saddr form (gemm_device.h); the LDS image of a stage is chunk-linear (row r = 12 chunks), and
//     since the LDS side of the copy is lane-linear the bank swizzle is applied to the SOURCE: physical slot p of row r holds logical
//     chunk (p - rot(r)) mod 12 with rot(r) = (r >> 2) & 3.  A row is 192 B = 48 banks, so the 16 lanes of one ds_read_b128 group
//     (16 consecutive rows, same logical chunk) would collide four ways; with the rotation they cover all 64 banks exactly once.
//   * per K tile and wave: 2 steps of 16 k; a step = 3 (TM + TN) ds_read_b128 and 6 TM TN MFMAs, accumulators walked round-robin.
//   * schedule of K tile kt (buffer kt & 1), no VALU instruction in the loop:
//       step 0 : MFMAs on fragment set 0, the reads of step 1's fragments in their shadow
//       barrier: this wave's copies of tile kt+1 (issued one step EARLIER, during step 1 of tile kt-1) have landed and its reads of
//                buffer kt & 1 have returned
//       step 1 : MFMAs on fragment set 1; in their shadow the reads of tile kt+1's step-0 fragments and the copies of tile kt+2 into
//                the buffer everybody just finished reading
//     so a copy has a whole step (>= 24 MFMAs of this wave plus its SIMD partner's) to land and no fragment read is exposed.
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"
#include "gemm_device.h"
#include "gemm_split_device.h"

namespace {

// X[rows][lds] fp32 -> split3 (row stride ldd fp32-equivalents = 6 ldd bytes); thread = 8 consecutive k of one row
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int64_t lds_, char* __restrict__ dst, int64_t ldd,
                                                     int64_t rows, int kg) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * kg) return;
    const int64_t r = idx / kg;
    const int g = (int)(idx - r * kg);
    const f32x4* p = reinterpret_cast<const f32x4*>(src + r * lds_ + (int64_t)g * 8);
    const f32x4 v0 = p[0], v1 = p[1];
    const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = bf16_rne(x[e]);
        const float r1 = x[e] - __uint_as_float(h[e] << 16);          // exact: the residual of a rounding fits the format
        m[e] = bf16_rne(r1);
        const float r2 = r1 - __uint_as_float(m[e] << 16);            // exact
        l[e] = bf16_rne(r2);
    }
    u32x4* o = reinterpret_cast<u32x4*>(dst + r * ldd * 6 + (int64_t)g * 48);
    o[0] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    o[1] = u32x4{m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
    o[2] = u32x4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}

// nn.LayerNorm over D = NV * 256 (vit.py:136,144; the arithmetic of rowops.hip::ln_wave_kernel, one wave per row) with the result
// written as a split3 operand: lane l holds 4 consecutive elements, so lanes 2 j and 2 j + 1 write the two 8-byte halves of the three
// chunks of k-group j — the fp32 copy of the normalised row is never written.
template <int NV>
__global__ __launch_bounds__(256) void ln_split3_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, char* __restrict__ y, int rows, float eps) {
    constexpr int D = NV * 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)row * D;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq += d * d;
        }
    const float var = wave_sum(sq) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    // The row's split3 image (6 D bytes) is assembled in a wave-private LDS buffer and then written as WHOLE lines: 1024 contiguous
    // bytes per store instruction.  Written straight from the registers a lane owns the 8-byte half of every third 16-byte chunk, and
    // such partial-line stores run at 3.8 instead of 6.2 TB/s (scripts/micro/store_patterns.hip, profiles/r4c_store_patterns.jsonl:
    // 24.8 vs 15.2 us for the 94 MB of a 64-crop operand).
    __shared__ __attribute__((aligned(16))) char rowbuf[4][D * 6];
    char* rb = rowbuf[threadIdx.x >> 6];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float o = (v[i][e] - mean) * rstd * g[e] + b[e];
            h[e] = bf16_rne(o);
            const float r1 = o - __uint_as_float(h[e] << 16);
            m[e] = bf16_rne(r1);
            const float r2 = r1 - __uint_as_float(m[e] << 16);
            l[e] = bf16_rne(r2);
        }
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        char* o8 = rb + (c >> 3) * 48 + (lane & 1) * 8;
        *reinterpret_cast<u32x2*>(o8) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2*>(o8 + 16) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *reinterpret_cast<u32x2*>(o8 + 32) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
    // wave-private buffer: the wave's own LDS writes are visible to its reads in program order (no barrier)
    char* yr = y + (int64_t)row * D * 6;
    constexpr int NCH = D * 6 / 16;                      // 16-byte chunks of the row (480)
#pragma unroll
    for (int i = 0; i < (NCH + 63) / 64; ++i) {
        const int ch = i * 64 + lane;
        if (ch < NCH) *reinterpret_cast<u32x4*>(yr + ch * 16) = *reinterpret_cast<const u32x4*>(rb + ch * 16);
    }
}

// ABL (timing-only experiments, results are garbage): bit0 no copies in the loop, bit1 no per-tile barrier, bit2 no fragment reads.
// RS: step 0 issues one fragment read every RS-th MFMA (0 = as early as possible: one per MFMA).
// ABLK: A is a ROW-BLOCKED split3 operand (GemmArgs::a_blk): its stage image in LDS is the memory image, [block of 32 rows][12 chunks of the K
// tile][32 rows][16 B] — linear copies, fragment reads of 512 contiguous bytes per 32 lanes (no swizzle needed).
template <int WM, int WN, int TM, int TN, int EPI, int ABL = 0, int RS = 0, bool ABLK = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_split3_kernel(GemmArgs a, int tiles_m, int tiles_n, int nwg) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_Q = BM * SLOTS / 64, B_Q = BN * SLOTS / 64;       // wave instructions (1 KiB each) per stage
    static_assert(A_Q % NW == 0 && B_Q % NW == 0, "tile / waves mismatch");
    constexpr int A_P = A_Q / NW, B_P = B_Q / NW, NP = A_P + B_P;     // copies per wave and K tile
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB;
    static_assert(2 * (A_STAGE + B_STAGE) <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(16))) char smem[2 * (A_STAGE + B_STAGE)];
    char* As = smem;                       // [2][BM][192]
    char* Bs = smem + 2 * A_STAGE;         // [2][BN][192]

    int tile_m, tile_n;
    int logical = logical_block(nwg, 0);
    int nk = a.K / SBK;
    int ksp = 0;
    if (a.ksplit > 1) {      // split-K: this block belongs to copy ksp of the tile grid and reduces K slice ksp into part[ksp] (wave-uniform)
        const int tiles = tiles_m * tiles_n;
        ksp = logical / tiles;
        logical -= ksp * tiles;
        nk /= a.ksplit;
        a.C += (int64_t)ksp * a.M * a.ldc;
    }
    tile_coords(tiles_m, tiles_n, logical, tile_m, tile_n);
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM * 32;
    const int wn0 = (wave % WN) * TN * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // ---- copies: wave instruction q = wave + i NW of an operand fills LDS chunks 64 q ... 64 q + 63 of its stage; lane -> chunk c,
    // row c / 12, physical slot c % 12, which holds logical chunk (slot - rot(row)) mod 12.  Rows past the edge are clamped.
    const int64_t arow = a.lda * 6, wrow = a.ldw * 6;                  // bytes per matrix row
    constexpr int A_KSTEP = ABLK ? SLOTS * 512 : ROWB;               // bytes a K tile advances the A source by (per 32-row block / per row)
    const char* Abase = reinterpret_cast<const char*>(a.A) + (int64_t)bm0 * arow + (int64_t)ksp * nk * A_KSTEP;
    const char* Wbase = reinterpret_cast<const char*>(a.W) + (int64_t)bn0 * wrow + (int64_t)ksp * nk * ROWB;
    uint32_t Aoff[A_P], Woff[B_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
        const int c = (wave + i * NW) * 64 + lane;
        if constexpr (ABLK) {      // LDS chunk c = (block c / 384, chunk-of-the-K-tile (c % 384) / 32, row-in-block c % 32): the memory order
            const int row = (c / 384) * 32 + (c & 31), rg = min(bm0 + row, a.M - 1) - bm0;
            Aoff[i] = (uint32_t)(rg >> 5) * (uint32_t)(a.lda * 192) + (uint32_t)((c % 384) >> 5) * 512u + (uint32_t)(rg & 31) * 16u;
        } else {
            const int row = c / SLOTS, slot = c - row * SLOTS;
            const int j = (slot + SLOTS - ((row >> 2) & 3)) % SLOTS;
            Aoff[i] = (uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)arow + (uint32_t)j * 16u;
        }
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
        const int c = (wave + i * NW) * 64 + lane, row = c / SLOTS, slot = c - row * SLOTS;
        const int j = (slot + SLOTS - ((row >> 2) & 3)) % SLOTS;
        Woff[i] = (uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)wrow + (uint32_t)j * 16u;
    }
    auto dma_piece = [&](int kt, int buf, int p) {                     // p, buf: compile-time after unrolling
        // wave-uniform, added on the SALU
        if (p < A_P) dma16_saddr(Abase + (int64_t)kt * A_KSTEP, Aoff[p], lds_addr_b(As + buf * A_STAGE + (wave + p * NW) * 1024));
        else dma16_saddr(Wbase + (int64_t)kt * ROWB, Woff[p - A_P], lds_addr_b(Bs + buf * B_STAGE + (wave + (p - A_P) * NW) * 1024));
    };

    // ---- fragments: step s of a K tile multiplies k-groups 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63); chunk j = 3 kgroup + piece
    // sits in physical slot (j + rot(row)) mod 12, rot(row) = (lrow >> 2) & 3 for every row this lane reads (tile offsets are x 32)
    uint32_t fo[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            fo[s][pc] = (uint32_t)lrow * ROWB + (uint32_t)((((2 * s + lhalf) * 3 + pc) + ((lrow >> 2) & 3)) % SLOTS) * 16u;
    const char* Afr = As + wm0 * ROWB;                                 // (row-blocked A: wm0 / 32 blocks of 12 x 512 bytes = the same offset)
    const char* Bfr = Bs + wn0 * ROWB;
    uint32_t foa[2][3];                                                // A fragments of a row-blocked stage: chunk (k-group, piece) x 512 + row x 16
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) foa[s][pc] = ABLK ? (uint32_t)(((2 * s + lhalf) * 3 + pc) * 512 + lrow * 16) : fo[s][pc];

    bf16x8 af[2][TM][3], bf[2][TN][3];
    constexpr int NR = 3 * (TM + TN);                                  // fragment reads per step
    // read r of step s of the tile in buffer `buf` into fragment set `set`: A pieces first, in the order the products use them
    auto read_one = [&](int buf, int s, int set, int r) {
        if (r < 3 * TM) {
            const int mi = r / 3, pc = r % 3;
            af[set][mi][pc] = *reinterpret_cast<const bf16x8*>(Afr + buf * A_STAGE + mi * 32 * ROWB + foa[s][pc]);
        } else {
            const int q = r - 3 * TM, ni = q / 3, pc = q % 3;
            bf[set][ni][pc] = *reinterpret_cast<const bf16x8*>(Bfr + buf * B_STAGE + ni * 32 * ROWB + fo[s][pc]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: tiles 0 and 1 in flight, tile 0's step-0 fragments in registers
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_piece(0, 0, p);
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_piece(min(1, nk - 1), 1, p);
    dma_wait_barrier();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(0, 0, 0, r);

    constexpr int G = NPROD * TM * TN;                                 // MFMAs per step
    constexpr int RS0 = RS > 0 ? RS : 1;                               // step 0: one read every RS0-th MFMA
    static_assert(NR * RS0 <= G && NR + NP <= G, "tile too small for the staging interleave");

    auto ktile = [&](int kt, auto bufc) {
        constexpr int buf = decltype(bufc){};
        const int kt2 = min(kt + 2, nk - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1 && !(ABL & 2)) {
                __builtin_amdgcn_sched_barrier(0);
                dma_wait_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < NPROD; ++p)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        const int idx = (p * TM + mi) * TN + ni;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][mi][piece_a(p)], bf[s][ni][piece_w(p)],
                                                                              acc[mi][ni], 0, 0, 0);
                        bool any = false;
                        if (s == 0) {
                            if (idx % RS0 == 0 && idx / RS0 < NR && !(ABL & 4)) { read_one(buf, 1, 1, idx / RS0); any = true; }
                        } else {
                            if (idx < NR) { if (!(ABL & 4)) { read_one(buf ^ 1, 0, 0, idx); any = true; } }
                            else if (idx - NR < NP && !(ABL & 1)) { dma_piece(kt2, buf, idx - NR); any = true; }
                        }
                        if (any) __builtin_amdgcn_sched_barrier(0);
                    }
        }
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        ktile(kt, IntC<0>{});
        ktile(kt + 1, IntC<1>{});
    }
    if (kt < nk) ktile(kt, IntC<0>{});
    if constexpr ((ABL & 4) != 0) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) asm volatile("" : "+v"(af[st][i][pc]));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) asm volatile("" : "+v"(bf[st][i][pc]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no LDS-DMA write may outlive the workgroup's LDS allocation

    if (a.c_split) {                        // wave-uniform: the result as the next GEMM's split3 operand
        constexpr int WT = TN * 32 + 4;
        static_assert(NW * TM * 32 * WT * 4 <= 2 * (A_STAGE + B_STAGE), "transpose tile does not fit the stage buffers");
        __syncthreads();                                               // every wave is done reading the stage buffers
        store_tile_split3<TM, TN, EPI>(a, acc, reinterpret_cast<float*>(smem) + wave * (TM * 32 * WT), bm0 + wm0, bn0 + wn0, lane);
        return;
    }
    store_tile<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
}

#ifdef THMR_EXPERIMENTS      // kernels that lost their A/B (numbers in the comments): experiments build only
// ---------------------------------------------------------------------------------------------------------------------
// Small-M split3 GEMM (few crops: the engine's split3 mode up to six crops) — the ring kernel of gemm_f32.hip on split3 operands:
// 64 x 64 tiles (2 x 2 waves of 32 x 32), ST-deep LDS ring of 24 KB stages fed by global_load_lds whose completion is tracked with
// s_waitcnt vmcnt(N), ONE barrier per 32-deep K tile in the middle of its 12 MFMAs, optional split-K into part[ksplit][M][N] (raw fp32
// partial tiles, summed in a fixed order by the residual + LayerNorm kernel that follows proj / fc2 anyway).  Per element K is summed
// in the order of gemm_split3_kernel (per 16 k: lh hl mm mh hm hh), so without split-K the two kernels are bit-identical.  A wave's
// chain is 12 MFMAs of 32 cycles per K tile instead of the fp32 ring kernel's 16 of 64: 0.375 of its matrix time.
template <int ST, int EPI, bool PARTIAL>
__global__ __launch_bounds__(256) void gemm_split3_ring_kernel(GemmArgs a, int tiles_m, int groups, int ksplit, float* part) {
    constexpr int NPW = 6;                                  // copies per wave and K tile (3 for A, 3 for W)
    constexpr int STAGE = 64 * ROWB;                        // 12 KB per operand and stage
    static_assert((ST & (ST - 1)) == 0 && ST >= 4, "ring depth must be a power of two >= 4");
    __shared__ __attribute__((aligned(16))) char smem[ST * 2 * STAGE];
    char* As = smem;                       // [ST][64][192]
    char* Bs = smem + ST * STAGE;          // [ST][64][192]

    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int gl = within / tiles_m, tile_m = within - gl * tiles_m;
    const int g = gl * 8 + xcd;
    if (g >= groups) return;
    const int tile_n = g / ksplit, sp = g - tile_n * ksplit;
    const int bm0 = tile_m * 64, bn0 = tile_n * 64;
    const int kper = a.K / ksplit, kbeg = sp * kper, nk = kper / SBK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    const int64_t arow = a.lda * 6, wrow = a.ldw * 6;
    const char* Abase = reinterpret_cast<const char*>(a.A) + (int64_t)bm0 * arow + (int64_t)(kbeg >> 3) * 48;
    const char* Wbase = reinterpret_cast<const char*>(a.W) + (int64_t)bn0 * wrow + (int64_t)(kbeg >> 3) * 48;
    uint32_t Aoff[3], Woff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = (wave + 4 * p) * 64 + lane, row = c / SLOTS, slot = c - row * SLOTS;
        const int j = (slot + SLOTS - ((row >> 2) & 3)) % SLOTS;
        Aoff[p] = (uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)arow + (uint32_t)j * 16u;
        Woff[p] = (uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)wrow + (uint32_t)j * 16u;
    }
    auto dma_tile = [&](int kt, auto stc) {      // K tile kt -> ring slot stc (= kt % ST, an integral constant)
        constexpr int st = decltype(stc){};
        const int64_t k0b = (int64_t)kt * ROWB;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            dma16_saddr(Abase + k0b, Aoff[p], lds_addr_b(As + st * STAGE + (wave + 4 * p) * 1024));
            dma16_saddr(Wbase + k0b, Woff[p], lds_addr_b(Bs + st * STAGE + (wave + 4 * p) * 1024));
        }
    };
    uint32_t fo[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            fo[s][pc] = (uint32_t)lrow * ROWB + (uint32_t)((((2 * s + lhalf) * 3 + pc) + ((lrow >> 2) & 3)) % SLOTS) * 16u;
    const char* Afr = As + wm0 * ROWB;
    const char* Bfr = Bs + wn0 * ROWB;
    bf16x8 af[2][3], bf[2][3];
    auto read_frags = [&](auto stc, int s, int set) {
        constexpr int st = decltype(stc){};
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            af[set][pc] = *reinterpret_cast<const bf16x8*>(Afr + st * STAGE + fo[s][pc]);
            bf[set][pc] = *reinterpret_cast<const bf16x8*>(Bfr + st * STAGE + fo[s][pc]);
        }
    };
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    // one step of 16 k: product 0, then the fragment reads of the NEXT step (nxt() may be empty) in the shadow of products 1-5
    auto step = [&](int set, auto nxt) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][piece_a(0)], bf[set][piece_w(0)], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        nxt();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 1; p < NPROD; ++p)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][piece_a(p)], bf[set][piece_w(p)], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: K tiles 0 .. ST-2 in flight; tile 0 must have landed (for every wave) before the first fragment read
    static_for<ST - 1>([&](auto t) {
        if (t < nk) dma_tile(t, t);
    });
    if (nk >= ST - 1) wait_vm_barrier<(ST - 2) * NPW>(); else wait_vm_barrier<0>();
    read_frags(IntC<0>{}, 0, 0);

    auto tile = [&](int kt, auto stc, bool last) {      // stc = kt % ST; `last` is a literal at every call site
        constexpr int S = decltype(stc){};
        step(0, [&] { read_frags(stc, 1, 1); });
        if (!last) {
            // tile kt+1 landed (in-order completion: at most the ST-3 younger tiles may still be in flight); after the barrier every
            // wave is past tile kt-1, whose ring slot receives tile kt+ST-1
            if (kt + ST - 2 < nk) wait_vm_barrier<(ST - 3) * NPW>(); else wait_vm_barrier<0>();
            if (kt + ST - 1 < nk) dma_tile(kt + ST - 1, IntC<(S + ST - 1) % ST>{});
        }
        step(1, [&] { if (!last) read_frags(IntC<(S + 1) % ST>{}, 0, 0); });
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
        if (a.c_split) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                           // every wave is done reading the ring
            store_tile_split3<1, 1, EPI>(a, acc, reinterpret_cast<float*>(smem) + wave * (32 * 36), bm0 + wm0, bn0 + wn0, lane);
            return;
        }
        store_tile<1, 1, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 block tile, 4 waves of 128 x 128 (ONE wave per SIMD, 256 accumulator registers), 16-deep stages — fewer LDS bytes per MFMA:
// a step needs 24 ds_read_b128 for 96 MFMAs (0.25 per MFMA against the 64 x 64 wave tile's 0.5) and the tile 125 B of copies per MFMA
// (192).  The schedule ablation of the 128 x 256 kernel prices the fragment reads at 20 % and the copies at 12.5 % of its time
// (profiles/r3v_split3_schedule_ablation.jsonl); this variant trades that against having no second wave per SIMD.
// A stage is 512 rows x 96 B = 48 KB (two stages); 6 chunks per row = 24 banks, rows r and r + 8 of a ds_read_b128 lane group collide,
// hence rot(r) = (r >> 3) & 1.  Per K tile (one step of 16 k): the copies of tile kt + 2 into the buffer whose fragments were read during
// the previous tile, the reads of tile kt + 1's fragments from the other buffer, 96 MFMAs; barrier.  Same K order per element as the other
// tiles (per 16 k: lh hl mm mh hm hh) -> bit-identical results.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_split3_wide_kernel(GemmArgs a, int tiles_m, int tiles_n, int nwg) {
    constexpr int TM = 4, TN = 4, WN = 2, NW = 4, BM = 256, BN = 256;
    constexpr int WSLOTS = 6, WROWB = 96;                             // 16 k: 2 k-groups x 3 pieces
    constexpr int A_STAGE = BM * WROWB, B_STAGE = BN * WROWB;         // 24 KB each
    constexpr int A_P = BM * WSLOTS / 64 / NW, B_P = BN * WSLOTS / 64 / NW, NP = A_P + B_P;     // 6 + 6 copies per wave and tile
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_STAGE + B_STAGE)];
    char* As = smem;
    char* Bs = smem + 2 * A_STAGE;

    int tile_m, tile_n;
    tile_coords(tiles_m, tiles_n, logical_block(nwg, 0), tile_m, tile_n);
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;
    const int nk = a.K / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM * 32, wn0 = (wave % WN) * TN * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    const int64_t arow = a.lda * 6, wrow = a.ldw * 6;
    const char* Abase = reinterpret_cast<const char*>(a.A) + (int64_t)bm0 * arow;
    const char* Wbase = reinterpret_cast<const char*>(a.W) + (int64_t)bn0 * wrow;
    uint32_t Aoff[A_P], Woff[B_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
        const int c = (wave + i * NW) * 64 + lane, row = c / WSLOTS, slot = c - row * WSLOTS;
        const int j = (slot + WSLOTS - ((row >> 3) & 1)) % WSLOTS;
        Aoff[i] = (uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)arow + (uint32_t)j * 16u;
        Woff[i] = (uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)wrow + (uint32_t)j * 16u;
    }
    auto dma_piece = [&](int kt, int buf, int p) {
        const int64_t k0b = (int64_t)kt * WROWB;
        if (p < A_P) dma16_saddr(Abase + k0b, Aoff[p], lds_addr_b(As + buf * A_STAGE + (wave + p * NW) * 1024));
        else dma16_saddr(Wbase + k0b, Woff[p - A_P], lds_addr_b(Bs + buf * B_STAGE + (wave + (p - A_P) * NW) * 1024));
    };
    uint32_t fo[3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
        fo[pc] = (uint32_t)lrow * WROWB + (uint32_t)(((lhalf * 3 + pc) + ((lrow >> 3) & 1)) % WSLOTS) * 16u;
    const char* Afr = As + wm0 * WROWB;
    const char* Bfr = Bs + wn0 * WROWB;
    bf16x8 af[2][TM][3], bf[2][TN][3];
    constexpr int NR = 3 * (TM + TN);
    auto read_one = [&](int buf, int set, int r) {
        if (r < 3 * TM) {
            const int mi = r / 3, pc = r % 3;
            af[set][mi][pc] = *reinterpret_cast<const bf16x8*>(Afr + buf * A_STAGE + mi * 32 * WROWB + fo[pc]);
        } else {
            const int q = r - 3 * TM, ni = q / 3, pc = q % 3;
            bf[set][ni][pc] = *reinterpret_cast<const bf16x8*>(Bfr + buf * B_STAGE + ni * 32 * WROWB + fo[pc]);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int p = 0; p < NP; ++p) dma_piece(0, 0, p);
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_piece(min(1, nk - 1), 1, p);
    dma_wait_barrier();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(0, 0, r);

    auto ktile = [&](int kt, auto bufc) {
        constexpr int buf = decltype(bufc){};                        // buffer and fragment set of tile kt
        const int kt2 = min(kt + 2, nk - 1);
#pragma unroll
        for (int p = 0; p < NPROD; ++p)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int idx = (p * TM + mi) * TN + ni;
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][mi][piece_a(p)], bf[buf][ni][piece_w(p)], acc[mi][ni], 0, 0, 0);
                    bool any = false;
                    if (idx < NP) { dma_piece(kt2, buf, idx); any = true; }
                    else if (idx - NP < NR) { read_one(buf ^ 1, buf ^ 1, idx - NP); any = true; }
                    if (any) __builtin_amdgcn_sched_barrier(0);
                }
        __builtin_amdgcn_sched_barrier(0);
        dma_wait_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        ktile(kt, IntC<0>{});
        ktile(kt + 1, IntC<1>{});
    }
    if (kt < nk) ktile(kt, IntC<0>{});

    if (a.c_split) {
        // one 32-row slice of the wave tile at a time through a wave-private 32 x 132 fp32 tile (the stage buffers are dead)
        constexpr int WT = TN * 32 + 4;
        static_assert(NW * 32 * WT * 4 <= 2 * (A_STAGE + B_STAGE), "transpose tile does not fit the stage buffers");
        float* T = reinterpret_cast<float*>(smem) + wave * (32 * WT);
        static_for<TM>([&](auto mic) {
            constexpr int mi = decltype(mic){};
            store_tile_split3<1, TN, EPI>(a, reinterpret_cast<f32x16(&)[1][TN]>(acc[mi]), T, bm0 + wm0 + mi * 32, bn0 + wn0, lane);
        });
        return;
    }
    store_tile<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
}

template <int WM, int WN, int TM, int TN, int ABL, int RS>
int launch_split3_abl(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_split3_kernel<WM, WN, TM, TN, EPI_NONE, ABL, RS>), dim3(nwg), dim3(WM * WN * 64), 0, s, a, tiles_m, tiles_n, nwg);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_split3_wide(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.ksplit > 1 || a.a_blk) return -1;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256, nwg = tiles_m * tiles_n;
    const dim3 grid(nwg), block(256);
#define THMR_WIDE_CASE(E)                                                                                       \
    case E:                                                                                                     \
        hipLaunchKernelGGL((gemm_split3_wide_kernel<E>), grid, block, 0, s, a, tiles_m, tiles_n, nwg);          \
        break;
    switch (epi) {
        THMR_WIDE_CASE(EPI_NONE)
        THMR_WIDE_CASE(EPI_BIAS)
        THMR_WIDE_CASE(EPI_BIAS_GELU)
        THMR_WIDE_CASE(EPI_BIAS_RESID)
        THMR_WIDE_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_WIDE_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#endif  // THMR_EXPERIMENTS

template <int WM, int WN, int TM, int TN>
int launch_split3_cfg(const GemmArgs& a, int epi, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, nwg = tiles_m * tiles_n * (a.ksplit > 1 ? a.ksplit : 1);
    const dim3 grid(nwg), block(WM * WN * 64);
    if (a.a_blk) {      // row-blocked A (fc2's operand): the two epilogues fc2 runs with — raw split-K partials and bias + residual
        if (epi == EPI_NONE) hipLaunchKernelGGL((gemm_split3_kernel<WM, WN, TM, TN, EPI_NONE, 0, 0, true>), grid, block, 0, s, a, tiles_m, tiles_n, nwg);
        else if (epi == EPI_BIAS_RESID) hipLaunchKernelGGL((gemm_split3_kernel<WM, WN, TM, TN, EPI_BIAS_RESID, 0, 0, true>), grid, block, 0, s, a, tiles_m, tiles_n, nwg);
        else return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
#define THMR_SPLIT_CASE(E)                                                                                                  \
    case E:                                                                                                                 \
        hipLaunchKernelGGL((gemm_split3_kernel<WM, WN, TM, TN, E>), grid, block, 0, s, a, tiles_m, tiles_n, nwg);    \
        break;
    switch (epi) {
        THMR_SPLIT_CASE(EPI_NONE)
        THMR_SPLIT_CASE(EPI_BIAS)
        THMR_SPLIT_CASE(EPI_BIAS_GELU)
        THMR_SPLIT_CASE(EPI_BIAS_RESID)
        THMR_SPLIT_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_SPLIT_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int launch_split3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int K, hipStream_t s) {
    if (rows <= 0 || K <= 0 || (K % 8) != 0 || (ld_src % 4) != 0 || (ld_dst % 8) != 0 || ld_dst < K) return -1;
    const int kg = K / 8;
    const int64_t n = rows * kg;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, reinterpret_cast<char*>(dst), ld_dst,
                       rows, kg);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_layernorm_split3(const float* x, const float* g, const float* b, void* y_split, int rows, int D, float eps, hipStream_t s) {
    if (rows <= 0 || D != 1280) return -1;
    hipLaunchKernelGGL(ln_split3_kernel<5>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, reinterpret_cast<char*>(y_split), rows, eps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int launch_split3_tiles(const GemmArgs& a, int epi, int variant, hipStream_t s);

// a.A / a.W point at split3 operands (lda / ldw = their row strides in fp32-equivalents, i.e. 6 lda bytes); C, bias, resid are fp32.
// variant: -1 = rule below   0 = 8 waves of 64x64 on 128x256   2 = 4 waves of 64x64 on 128x128   6 = 8 waves of 64x32 on 128x128   8 = 4 waves on 128x128, three-stage K ring   5 / 7 = 128x256 with the
// ragged last round as half tiles on 4 / 8 waves   (experiments build: 1 = 4 waves of 64x128 on 128x256, ...)
int launch_gemm_split3(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % SBK) != 0 || (a.lda % 8) != 0 || (a.ldw % 8) != 0) return -1;
    if (a.lda * 6 * 256 >= (int64_t(1) << 32) || a.ldw * 6 * 256 >= (int64_t(1) << 32)) return -1;     // 32-bit lane offsets within a tile
    if (a.cs_out != nullptr || a.ksplit > 1) return -1;
    if (a.c_split != nullptr && ((a.N % 8) != 0 || (a.ldcs % 8) != 0 || a.ldcs < a.N || epi == EPI_BIAS_RESID)) return -1;
    return launch_split3_tiles(a, epi, variant, s);
}

// Split-K on the split3 big tiles (the mode's 3 ... 31 crops: the N = 1280 GEMMs have 25-240 tiles of 128 x 256): `ksplit` copies of the tile
// grid in ONE launch, copy sp reducing K slice sp into part[sp][M][N] (raw fp32 partial tiles, no epilogue), summed in a fixed order by the
// residual + LayerNorm kernel that follows proj / fc2 anyway.  Tile by the same rule over tiles * ksplit (bit-identical either way).
int launch_gemm_split3_splitk(const GemmArgs& a0, int ksplit, float* part, hipStream_t s) {
    if (ksplit < 2 || part == nullptr) return -1;
    if (a0.M <= 0 || a0.N <= 0 || a0.K <= 0 || (a0.K % (SBK * ksplit)) != 0 || (a0.lda % 8) != 0 || (a0.ldw % 8) != 0) return -1;
    if (a0.lda * 6 * 256 >= (int64_t(1) << 32) || a0.ldw * 6 * 256 >= (int64_t(1) << 32) || a0.cs_out != nullptr || a0.c_split != nullptr) return -1;
    GemmArgs a = a0;
    a.ksplit = ksplit;
    a.C = part; a.ldc = a.N;
    a.bias = nullptr; a.resid = nullptr; a.ldr = 0;
    return launch_split3_tiles(a, EPI_NONE, -1, s);
}

// compute units of the current device (cached per device)
static int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cache[dev] = n;
    return n;
}

static int launch_split3_tiles(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    static const int forced = [] { const char* e = thmr_knob("THMR_SPLIT3_TILE"); return e ? atoi(e) : -1; }();     // A/B knob (0 / 2)
    if (variant < 0 && forced >= 0) variant = forced;
    bool by_rule = false;
    if (variant < 0) {
        // the tiles are bit-identical (same K order per element), so the choice is purely a matter of time:
        // 128 x 256 (8 waves) unless the whole grid of 128 x 128 tiles still fits ONE round of 256 CUs (the N = 1280 GEMMs at 16 crops, and
        // at 7-8 crops with their K split two ways: 240 workgroups instead of 120).  A rounds x tile-time model over both shapes (a
        // 128 x 128 tile takes 0.55 of a 128 x 256 one) was tried and is worse wherever it differs — 640 vs 669 crops/s at 15 crops, 665 vs
        // 707 at 48: a partly filled last round of big tiles costs less than a full round (profiles/r3ae_split3_tile_rule_ab.log)
        const long ks = a.ksplit > 1 ? a.ksplit : 1;
        const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * ks;
        // round 6: the 128 x 128 tile on EIGHT waves of 64 x 32 (variant 6; gemm_split16.hip) was built to put two waves on every SIMD where
        // that tile is what runs (few crops) and measured SLOWER than the four-wave form, same box, interleaved, whole path
        // (profiles/r6d_ab_narrow8_*): +5.9 % per call at 4 crops (fc2 +0.19 ms, fc1 +0.14, qkv +0.12 of 8.1), +3.7 % at 8 (fc2 +0.42), +2.9 % at 3,
        // +3.2 % at 5, +0.6 % at 16.  The few-crop K loop is not short of waves: it waits for its LDS-DMA copies (one 48 KB stage in flight per
        // CU against a 1.5 us loaded round trip = the 1.55 us per K tile measured), and eight waves add barrier cost without adding bytes
        // in flight.  Kept as variant 6 / THMR_SPLIT3_NARROW8=1 (bit-identical, tested).
        static const bool narrow8 = [] { const char* e = thmr_knob("THMR_SPLIT3_NARROW8"); return e && e[0] == '1'; }();
        // round 6: the four-wave 128 x 128 tile with a THREE-stage K ring (variant 8: two stages of copies in flight per CU)
        static const int ring3 = [] { const char* e = thmr_knob("THMR_SPLIT3_RING3"); return e ? atoi(e) : -1; }();
        const bool r3 = ring3 >= 0 ? ring3 != 0 : !(a.tile_opts & 4);
        const bool n8 = narrow8 || (a.tile_opts & 1);
        variant = t128 <= 256 ? (n8 ? (r3 ? 9 : 6) : r3 ? 8 : 2) : 0;
        by_rule = true;
    }
    if ((by_rule && variant == 0) || variant == 5 || variant == 7) {
        // round 5: the 128 x 256 grid with its ragged last round as 128 x 128 half tiles, where that applies (fc1 of a 64-crop batch: 1920
        // tiles = 7.5 rounds of 256 CUs; gemm_split16.hip gemm_split16_tail_kernel).  Bit-identical to the plain grid.  THMR_SPLIT3_TAIL=0
        // (experiments build) switches it off for the A/B; variant 5 asks for it explicitly and fails if the shape does not qualify.
        static const bool tail_off = [] { const char* e = thmr_knob("THMR_SPLIT3_TAIL"); return e && e[0] == '0'; }();
        // round 6: the half tiles on all eight waves (64 x 32 wave tiles) instead of four = variant 7 / THMR_SPLIT3_TAIL8=1: measured equal
        // (fc1 +0.09 ms of 22.1 per 64-crop step, whole step +0.1 %: profiles/r6d_ab_narrow8_or_tail8_b64.json); round 5's form stays
        static const bool tail8 = [] { const char* e = thmr_knob("THMR_SPLIT3_TAIL8"); return e && e[0] == '1'; }();
        if (variant == 5 || variant == 7 || !tail_off) {
            const int r = launch_split16_tiles_tail(a, epi, device_cus(), variant == 7 ? true : (tail8 || (a.tile_opts & 2)), s);
            if (r <= 0) return r;                 // launched, or failed
            if (variant == 5 || variant == 7) return -1;          // does not apply to this shape
        }
        if (variant == 5 || variant == 7) return -1;
    }
    if (by_rule && variant == 0 && !(a.tile_opts & 8) && !a.a_blk) {
        // round 6: a 128 x 256 grid of at most two rounds issues every copy of a K tile right behind the barrier (gemm_split16.hip FRONT): its
        // period is the weights' round trip, not the MFMA time.  Same box, per call (profiles/r6o_*): fc2 at 32 crops (240 tiles) 10.83 -> 10.38
        // ms, fc1 / fc2 at 16 crops -0.12 / -0.16, fc1 at 8 -0.08; equal at 64 crops (not used there: more rounds); the three-stage
        // 128 x 128 tile LOSES with it (12 copies in a row stall its one wave per SIMD: +0.1 ... +0.26 ms per class at 4 / 8 crops)
        const long ks = a.ksplit > 1 ? a.ksplit : 1;
        if ((long)((a.M + 127) / 128) * ((a.N + 255) / 256) * ks <= 512) variant = 10;
    }
    switch (variant) {
        // round 4: the product kernels multiply on v_mfma_f32_16x16x32_bf16 (gemm_split16.hip); the 32x32x16 kernels of this file are the
        // experiments build's variants 20 / 22 (and 1, 4, 3x below) — another grouping of k inside the MFMA, so equal to rounding only
        case 0: return launch_split16_tiles(a, epi, 0, s);
        case 2: return launch_split16_tiles(a, epi, 1, s);
        case 6: return launch_split16_tiles(a, epi, 2, s);
        case 8: return launch_split16_tiles(a, epi, 3, s);
        case 9: return launch_split16_tiles(a, epi, 4, s);
        case 10: return launch_split16_tiles(a, epi, 5, s);
        case 11: return launch_split16_tiles(a, epi, 6, s);
#ifdef THMR_EXPERIMENTS
        case 20: return launch_split3_cfg<2, 4, 2, 2>(a, epi, s);
        case 22: return launch_split3_cfg<2, 2, 2, 2>(a, epi, s);
        case 1: return launch_split3_cfg<2, 2, 2, 4>(a, epi, s);    // 4 waves of 64 x 128: 3 % slower (r3v)
        case 4: return launch_split3_wide(a, epi, s);               // 256 x 256, 4 waves of 128 x 128, 16-deep stages: the same rate (r3ak)
        // experiments on the default tile, EPI_NONE only: 3 = step-0 fragment reads every 2nd MFMA (the first version's schedule);
        // 31 / 32 / 34 / 37 = timing-only ablations (no copies / no barrier / no fragment reads / none of the three): garbage results
        case 3: return epi == EPI_NONE ? launch_split3_abl<2, 4, 2, 2, 0, 2>(a, s) : -1;
        case 31: return epi == EPI_NONE ? launch_split3_abl<2, 4, 2, 2, 1, 0>(a, s) : -1;
        case 32: return epi == EPI_NONE ? launch_split3_abl<2, 4, 2, 2, 2, 0>(a, s) : -1;
        case 34: return epi == EPI_NONE ? launch_split3_abl<2, 4, 2, 2, 4, 0>(a, s) : -1;
        case 37: return epi == EPI_NONE ? launch_split3_abl<2, 4, 2, 2, 7, 0>(a, s) : -1;
#endif
        default: return -1;
    }
}

#ifdef THMR_EXPERIMENTS
// small-M split3 GEMM: 64x64 tiles on a 4-deep LDS-DMA ring; ksplit > 1 writes raw partial sums to part[ksplit][M][N] and applies NO
// epilogue (launch_splitk_resid_ln / launch_splitk_epilogue then do)
int launch_gemm_split3_ring(const GemmArgs& a, int epi, int ksplit, float* part, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || ksplit < 1 || (a.K % (SBK * ksplit)) != 0 || (a.lda % 8) != 0 || (a.ldw % 8) != 0) return -1;
    if (a.lda * 6 * 64 >= (int64_t(1) << 32) || a.ldw * 6 * 64 >= (int64_t(1) << 32) || (ksplit > 1 && part == nullptr)) return -1;
    if (a.cs_out != nullptr || a.ksplit > 1 || a.a_blk) return -1;
    if (a.c_split != nullptr && ((a.N % 8) != 0 || (a.ldcs % 8) != 0 || a.ldcs < a.N || epi == EPI_BIAS_RESID || ksplit > 1)) return -1;
    const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 63) / 64;
    const int groups = tiles_n * ksplit;
    dim3 grid(8 * tiles_m * ((groups + 7) / 8)), block(256);
    if (ksplit > 1) {
        hipLaunchKernelGGL((gemm_split3_ring_kernel<4, EPI_NONE, true>), grid, block, 0, s, a, tiles_m, groups, ksplit, part);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
#define THMR_SRING_CASE(E)                                                                                              \
    case E:                                                                                                             \
        hipLaunchKernelGGL((gemm_split3_ring_kernel<4, E, false>), grid, block, 0, s, a, tiles_m, groups, 1, nullptr);   \
        break;
    switch (epi) {
        THMR_SRING_CASE(EPI_NONE)
        THMR_SRING_CASE(EPI_BIAS)
        THMR_SRING_CASE(EPI_BIAS_GELU)
        THMR_SRING_CASE(EPI_BIAS_RESID)
        THMR_SRING_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_SRING_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#endif  // THMR_EXPERIMENTS
