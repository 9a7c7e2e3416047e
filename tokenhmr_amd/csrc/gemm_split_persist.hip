// The split3 GEMM (gemm_split.hip: fp32 operands as three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16 k) as PERSISTENT workgroups
// over a tile stream.  Same 128 x 256 tile, same 8 waves of 64 x 64, same LDS image and same K loop as gemm_split3_kernel — and the same
// bits: every output element is ONE accumulator chain over k = 0 ... K-1 in the same order, here too.  What changes is who runs which
// part of which tile, and when (vit.py:82-87,104-126 are the four products this serves at 32 crops and more).
//
// Why (profiles/r3ah_split3_kernel_stats.csv): with one workgroup per tile and one workgroup per CU the four ViT GEMMs of a
// 64-crop batch have 1440 / 480 / 1920 / 480 tiles for 256 CUs = 5.625 / 1.875 / 7.5 / 1.875 rounds — every one of them idles 6.25 % of
// the chip in a ragged last round — and every tile pays its own pipeline fill (two K tiles of LDS-DMA latency) and its own epilogue with
// nothing running underneath (fc1's GELU + split3 epilogue: 812 us against 716 us for the same tiles at the qkv rate).
//
// Decomposition.  256 workgroups (one per CU), workgroup b = (xcd = b & 7, lane = b >> 3).  Lane i of the eight XCDs owns the tile list
// {i, i + 32, i + 64, ...} in the XCD-aware logical order of gemm_device.h::tile_coords — T_i tiles of nk K tiles each = T_i nk steps, cut into
// eight equal consecutive step ranges, one per XCD.  So the 32 workgroups of an XCD walk 32 NEIGHBOURING tiles (8 x 4: the L2 footprint of
// gemm_split3_kernel's dispatch order) in lockstep, all with the same range boundaries, and a tile that straddles a boundary is split ALONG K
// between the same lane of two neighbouring XCDs.
//
// The K split keeps the association: the first part (k tiles [0, p)) is computed by XCD x as the FIRST thing it does, its raw fp32
// accumulators go to a workspace slab (write-through stores) and a flag is published; XCD x + 1 runs the rest (k tiles [p, nk)) as the LAST
// thing it does, starting from those accumulators instead of zeros — the accumulator chain merely travels through memory, bit for bit.
// (Summing two independently accumulated halves would be another association: a crop's result would depend on which tile it lands in.)
// The producer never waits for anybody and the consumer needs the slab some hundred microseconds after it was written, so the spin on the
// flag is nominal; it is bounded all the same (error word, no hang).  Placement-independent: cdna_hip_programming.md Guideline 16, R1 form
// (sc1 payload stores, every storing wave drains before a workgroup barrier, one lane publishes; sc1 loads on the consumer).
//
// Pipeline.  The two-stage LDS-DMA pipeline runs ACROSS tile boundaries: while the last K tiles of one tile are multiplied the first two K
// tiles of the next one are already being copied, the epilogue's stores drain under the next tile's MFMAs, and the only exposed cost per
// tile is the epilogue's own VALU work.  For a split3 OUTPUT (fc1: GELU, then three bf16 pieces per value, 8 consecutive columns per
// 16-byte chunk) that requires an epilogue that leaves the stage buffers alone: MODE 2 multiplies with the operand ROLES swapped (the MFMA's
// "A" is the weight fragment), which leaves a lane with one ROW of the output and 4-column groups in its registers, and eight
// v_permlane32_swap per accumulator turn those into two groups of 8 consecutive columns — no LDS, no barrier.  MODE 1 keeps the LDS
// transposition of gemm_split3_kernel (coalesced 384-byte row pieces) and pays a pipeline drain + refill per tile for it.
#include <map>
#include <mutex>

#include "common.h"
#include "gemm_device.h"
#include "gemm_split_device.h"

namespace {

constexpr int PG = 32;                                     // stream lanes (workgroups) per XCD
constexpr int PBM = 128, PBN = 256;
constexpr int P_SLAB = PBM * PBN;                          // floats of one raw accumulator tile (128 KB)
constexpr int P_NWG = 8 * PG;

struct PersistWs {
    float* part;        // [8 * PG] slabs of P_SLAB floats: slab (x, i) = first part of the tile shared by lane i of XCDs x and x + 1
    unsigned* flag;     // [8 * PG] 0 / 1, set by the producer, cleared by the consumer (stream order separates launches); [8 * PG] = error word
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// MODE 2 epilogue: acc[mi][ni] holds the TRANSPOSED 32 x 32 tile (lane & 31 = row m of the output, registers = columns n:
// n = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)).  v_permlane32_swap(X = reg j of column block g, Y = reg j of block g + 1) exchanges X's upper
// 32 lanes with Y's lower 32: afterwards lanes 0-31 hold columns 8 h ... 8 h + 7 of blocks (g, g + 1) = 16 g' + 0 ... 7 and lanes 32-63
// columns 16 g' + 8 ... 15.  Then bias + activation + split3 per value and three 16-byte stores (48 contiguous bytes) per group.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile_split3_swapped(const GemmArgs& a, f32x16 (&acc)[TM][TN], int m0, int n0, int lane) {
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + mi * 32 + lrow;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][8 * gp + j]),
                                                                       __float_as_uint(acc[mi][ni][8 * gp + 4 + j]), false, false);
                    v[j] = __uint_as_float(sw.x);
                    v[4 + j] = __uint_as_float(sw.y);
                }
                const int n = n0 + ni * 32 + gp * 16 + lhalf * 8;
                float bias[8];
                if constexpr (EPI != EPI_NONE) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
                    bias[0] = b0[0]; bias[1] = b0[1]; bias[2] = b0[2]; bias[3] = b0[3];
                    bias[4] = b1[0]; bias[5] = b1[1]; bias[6] = b1[2]; bias[7] = b1[3];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) bias[u] = 0.f;
                }
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        const f32x2 gl = gelu_erf2(f32x2{v[u] + bias[u], v[u + 1] + bias[u + 1]});
                        v[u] = gl.x;
                        v[u + 1] = gl.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = gemm_epilogue<EPI>(a, v[u], bias[u], m, n + u);
                }
                uint32_t H[4], M[4], L[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) split3_pair(v[2 * u], v[2 * u + 1], H[u], M[u], L[u]);
                char* cb = reinterpret_cast<char*>(a.c_split);
                const bool blk = a.cs_blk != 0;
                *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 0, blk)) = u32x4{H[0], H[1], H[2], H[3]};
                *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 1, blk)) = u32x4{M[0], M[1], M[2], M[3]};
                *reinterpret_cast<u32x4*>(cb + split3_chunk_off(a.ldcs, m, n >> 3, 2, blk)) = u32x4{L[0], L[1], L[2], L[3]};
            }
        }
    }
}

// MODE: 0 = fp32 output (store_tile), pipeline continuous across tiles
//       1 = split3 output through the LDS transposition (store_tile_split3): pipeline drained and refilled per tile
//       2 = split3 output with swapped operand roles + v_permlane32_swap: pipeline continuous
// ABLK: A is a row-blocked split3 operand (GemmArgs::a_blk; see gemm_split3_kernel)
template <int EPI, int MODE, bool ABLK = false>
__global__ __launch_bounds__(512) void gemm_split3_persist_kernel(GemmArgs a, int tiles_m, int tiles_n, PersistWs ws) {
    constexpr int NW = 8, WN = 4, TM = 2, TN = 2;
    constexpr bool SWAP = MODE == 2, CONT = MODE != 1;
    constexpr int A_Q = PBM * SLOTS / 64, B_Q = PBN * SLOTS / 64;
    constexpr int A_P = A_Q / NW, B_P = B_Q / NW, NP = A_P + B_P;      // 3 + 6 copies per wave and K tile
    constexpr int A_STAGE = PBM * ROWB, B_STAGE = PBN * ROWB;
    static_assert(2 * (A_STAGE + B_STAGE) <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(16))) char smem[2 * (A_STAGE + B_STAGE)];
    char* As = smem;
    char* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM * 32;
    const int wn0 = (wave % WN) * TN * 32;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // ---- this workgroup's step range of its lane's tile list, as segments: [first part of a shared tile] [whole tiles] [rest of a shared tile]
    const int xcd = blockIdx.x & 7, ln = blockIdx.x >> 3;
    const int nk = a.K / SBK;
    const int T = (tiles_m * tiles_n - ln + PG - 1) / PG;              // >= 8 (launcher)
    const int S0 = (int)((int64_t)xcd * T * nk / 8), S1 = (int)((int64_t)(xcd + 1) * T * nk / 8);
    const int j0 = S0 / nk, k0 = S0 - j0 * nk;
    const int j1 = (S1 - 1) / nk, k1 = S1 - j1 * nk;
    const int has_pre = k1 < nk ? 1 : 0, has_post = k0 > 0 ? 1 : 0;    // a range is >= nk steps long, so the two are never the same tile
    const int jf0 = j0 + has_post, nfull = max(j1 - has_pre - jf0 + 1, 0);
    const int nseg = has_pre + nfull + has_post;
    // segment n -> tile j of the lane's list, K tiles [kb, ke), kind 0 whole / 1 first part (store the accumulators) / 2 rest (load them)
    auto seg_of = [&](int n, int& j, int& kb, int& ke, int& kind) {
        const int m = n - has_pre;
        if (has_pre && n == 0) { j = j1; kb = 0; ke = k1; kind = 1; }
        else if (m < nfull) { j = jf0 + m; kb = 0; ke = nk; kind = 0; }
        else { j = j0; kb = k0; ke = nk; kind = 2; }
    };
    auto tile_of = [&](int j, int& bm0, int& bn0) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, j * PG + ln, tm, tn);
        bm0 = __builtin_amdgcn_readfirstlane(tm * PBM);              // wave-uniform by construction (blockIdx and kernel arguments only)
        bn0 = __builtin_amdgcn_readfirstlane(tn * PBN);
    };

    // ---- copies (M % 128 == 0 and N % 256 == 0: no clamping, the lane offsets are the same for every tile)
    const int64_t arow = a.lda * 6, wrow = a.ldw * 6;
    uint32_t Aoff[A_P], Woff[B_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
        const int c = (wave + i * NW) * 64 + lane;
        if constexpr (ABLK) {
            Aoff[i] = (uint32_t)(c / 384) * (uint32_t)(a.lda * 192) + (uint32_t)(c % 384) * 16u;       // linear within a 32-row block
        } else {
            const int row = c / SLOTS, slot = c - row * SLOTS;
            Aoff[i] = (uint32_t)row * (uint32_t)arow + (uint32_t)((slot + SLOTS - ((row >> 2) & 3)) % SLOTS) * 16u;
        }
    }
    constexpr int A_KSTEP = ABLK ? SLOTS * 512 : ROWB;               // bytes a K tile advances the A source by
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
        const int c = (wave + i * NW) * 64 + lane, row = c / SLOTS, slot = c - row * SLOTS;
        Woff[i] = (uint32_t)row * (uint32_t)wrow + (uint32_t)((slot + SLOTS - ((row >> 2) & 3)) % SLOTS) * 16u;
    }
    // fetch cursor: the K tile the NEXT copy brings in (two ahead of the multiply), wave-uniform
    int fn = 0, fk = 0, fke = 0;
    const char *fA = nullptr, *fW = nullptr;
    auto fetch_seg = [&](int n) {
        int j, kb, ke, kind, bm0, bn0;
        seg_of(n, j, kb, ke, kind);
        tile_of(j, bm0, bn0);
        fk = kb;
        fke = ke;
        fA = reinterpret_cast<const char*>(a.A) + (int64_t)bm0 * arow + (int64_t)kb * A_KSTEP;
        fW = reinterpret_cast<const char*>(a.W) + (int64_t)bn0 * wrow + (int64_t)kb * ROWB;
    };
    auto fetch_advance = [&]() {
        if (fk + 1 < fke) { ++fk; fA += A_KSTEP; fW += ROWB; }
        else if (CONT && fn + 1 < nseg) fetch_seg(++fn);
        // else: past the end (of the segment in MODE 1): the last K tile is copied again, into a buffer nobody reads any more
    };
    auto dma_piece = [&](int buf, int p) {
        if (p < A_P) dma16_saddr(fA, Aoff[p], lds_addr_b(As + buf * A_STAGE + (wave + p * NW) * 1024));
        else dma16_saddr(fW, Woff[p - A_P], lds_addr_b(Bs + buf * B_STAGE + (wave + (p - A_P) * NW) * 1024));
    };

    // ---- fragments (as gemm_split3_kernel)
    uint32_t fo[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            fo[s][pc] = (uint32_t)lrow * ROWB + (uint32_t)((((2 * s + lhalf) * 3 + pc) + ((lrow >> 2) & 3)) % SLOTS) * 16u;
    const char* Afr = As + wm0 * ROWB;
    const char* Bfr = Bs + wn0 * ROWB;
    uint32_t foa[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) foa[s][pc] = ABLK ? (uint32_t)(((2 * s + lhalf) * 3 + pc) * 512 + lrow * 16) : fo[s][pc];
    bf16x8 af[2][TM][3], bf[2][TN][3];
    constexpr int NR = 3 * (TM + TN);
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
    constexpr int G = NPROD * TM * TN;
    static_assert(NR + NP <= G, "tile too small for the staging interleave");

    // one K tile out of buffer `buf` (gemm_split3_kernel's schedule): step 0's MFMAs over fragment set 0 with the reads of step 1's
    // fragments in their shadow; barrier; step 1's MFMAs with the reads of the next K tile's first fragments (other buffer) and the copies of
    // the K tile after next (into this buffer) in their shadow
    auto ktile = [&](auto bufc) {
        constexpr int buf = decltype(bufc){};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1) {
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
                        if constexpr (SWAP)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s][ni][piece_w(p)], af[s][mi][piece_a(p)], acc[mi][ni], 0, 0, 0);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][mi][piece_a(p)], bf[s][ni][piece_w(p)], acc[mi][ni], 0, 0, 0);
                        bool any = false;
                        if (s == 0) {
                            if (idx < NR) { read_one(buf, 1, 1, idx); any = true; }
                        } else {
                            if (idx < NR) { read_one(buf ^ 1, 0, 0, idx); any = true; }
                            else if (idx - NR < NP) { dma_piece(buf, idx - NR); any = true; }
                        }
                        if (any) __builtin_amdgcn_sched_barrier(0);
                    }
        }
        fetch_advance();
    };

    // fill: the first two K tiles of segment n into buffers 0 / 1, tile 0's first fragments into set 0
    auto fill = [&](int n) {
        fetch_seg(n);
        fn = n;
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(0, p);
        fetch_advance();
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(1, p);
        fetch_advance();
        dma_wait_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r) read_one(0, 0, 0, r);
    };

    static_assert((int64_t)P_NWG * P_SLAB * 4 < (int64_t(1) << 31), "slab offsets are 32-bit");
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ws.part, 0, P_NWG * P_SLAB * 4, 0x00020000);
    const uint32_t slab_lane = (uint32_t)(wave * (TM * TN * 4) * 1024 + lane * 16);     // a wave's 16 chunk rows of 1 KiB each
    int pub_pending = 0;         // the slab's stores are issued; the flag goes out after the next K tile's barrier (every wave drained)
    int par = 0;                 // buffer of the next K tile
    auto after_tile = [&]() {
        if (pub_pending) {
            if (tid == 0) __hip_atomic_store(ws.flag + xcd * PG + ln, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pub_pending = 0;
        }
    };

    if constexpr (CONT) fill(0);
    for (int n = 0; n < nseg; ++n) {
        int j, kb, ke, kind, bm0, bn0;
        seg_of(n, j, kb, ke, kind);
        tile_of(j, bm0, bn0);
        if constexpr (!CONT) {
            fill(n);
            par = 0;
        }
        if (kind == 2) {
            // the accumulators of k tiles [0, kb) from lane ln of the previous XCD: one thread polls one word, then sc1 loads
            if (tid == 0) {
                unsigned spins = 0;
                unsigned* f = ws.flag + (xcd - 1) * PG + ln;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) {                        // ~0.5 s: report, never hang
                        __hip_atomic_store(ws.flag + P_NWG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch (stream order)
            }
            asm volatile("s_barrier" ::: "memory");
            const uint32_t base = (uint32_t)((xcd - 1) * PG + ln) * (uint32_t)(P_SLAB * 4) + slab_lane;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (uint32_t)(((mi * TN + ni) * 4 + q) * 1024), 0, 16);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[mi][ni][q * 4 + c] = __uint_as_float(r[c]);
                    }
        } else {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        }

        int cnt = ke - kb;
        if (cnt > 0 && par) { ktile(IntC<1>{}); after_tile(); --cnt; par = 0; }
        for (; cnt >= 2; cnt -= 2) {
            ktile(IntC<0>{});
            after_tile();
            ktile(IntC<1>{});
            after_tile();
        }
        if (cnt) { ktile(IntC<0>{}); after_tile(); par = 1; }

        if (kind == 1) {
            // raw accumulators -> slab (xcd, ln), write-through; published after the next K tile's barrier
            const uint32_t base = (uint32_t)(xcd * PG + ln) * (uint32_t)(P_SLAB * 4) + slab_lane;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 v = {__float_as_uint(acc[mi][ni][q * 4]), __float_as_uint(acc[mi][ni][q * 4 + 1]),
                                         __float_as_uint(acc[mi][ni][q * 4 + 2]), __float_as_uint(acc[mi][ni][q * 4 + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, base + (uint32_t)(((mi * TN + ni) * 4 + q) * 1024), 0, 16);
                    }
            pub_pending = 1;
        } else if constexpr (MODE == 0) {
            store_tile<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lrow, lhalf);
        } else if constexpr (MODE == 2) {
            store_tile_split3_swapped<TM, TN, EPI>(a, acc, bm0 + wm0, bn0 + wn0, lane);
        } else {
            constexpr int WT = TN * 32 + 4;
            static_assert(NW * TM * 32 * WT * 4 <= 2 * (A_STAGE + B_STAGE), "transpose tile does not fit the stage buffers");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the copies past the segment's end have landed
            __syncthreads();                                           // every wave is done with the stage buffers
            store_tile_split3<TM, TN, EPI>(a, acc, reinterpret_cast<float*>(smem) + wave * (TM * 32 * WT), bm0 + wm0, bn0 + wn0, lane);
            __syncthreads();                                           // ... and with the transpose tiles, before the next fill
        }
        if constexpr (!CONT) {
            if (pub_pending) {                                         // MODE 1 has no next K tile to hide behind: drain and publish now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                after_tile();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no LDS-DMA write may outlive the workgroup's LDS allocation
    if (pub_pending) {                                                 // (a first part is never a range's last segment; kept for safety)
        __syncthreads();
        after_tile();
    }
}

template <int EPI, int MODE, bool ABLK = false>
int launch_persist_cfg(const GemmArgs& a, const PersistWs& ws, hipStream_t s) {
    const int tiles_m = a.M / PBM, tiles_n = a.N / PBN;
    hipLaunchKernelGGL((gemm_split3_persist_kernel<EPI, MODE, ABLK>), dim3(P_NWG), dim3(512), 0, s, a, tiles_m, tiles_n, ws);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

size_t gemm_split3_persist_ws_bytes() { return (size_t)P_NWG * P_SLAB * 4 + (P_NWG + 64) * sizeof(unsigned); }

bool gemm_split3_persist_ok(const GemmArgs& a) {
    // (round 6: M may be ragged — an odd number of crops — for the product kernel of gemm_split16.hip; the row-blocked operands hold whole
    // 32-row blocks, which 192 B rows always are)
    if (a.M <= 0 || a.N <= 0 || a.K < 2 * SBK || (a.K % SBK) != 0 || (a.M % 32) != 0 || (a.N % PBN) != 0) return false;
    if ((int64_t)((a.M + PBM - 1) / PBM) * (a.N / PBN) < P_NWG) return false;     // every lane's list holds >= 8 tiles: a range >= one tile
    if ((a.lda % 8) != 0 || (a.ldw % 8) != 0 || a.lda * 6 * 256 >= (int64_t(1) << 32) || a.ldw * 6 * 256 >= (int64_t(1) << 32)) return false;
    if (a.cs_out != nullptr || a.ksplit > 1) return false;
    if (a.c_split != nullptr && ((a.N % 8) != 0 || (a.ldcs % 8) != 0 || a.ldcs < a.N)) return false;
    return true;
}

// ws: gemm_split3_persist_ws_bytes() of device memory, ZEROED once when it is allocated; one launch at a time per workspace.  Layout: 256 slabs,
// then 256 flag words (gemm_split16.hip: the epoch of the launch that published the slab), then the control words: + 0 error, + 1 epoch,
// + 2 arrival counter, + 4 / + 5 the address of a host-mapped error word (gemm_split3_persist_bind_host_err; zero = none).  mode: 0 fp32 output; 1 / 2 split3 output (a.c_split) through LDS / through swapped operand roles.
// round 4: modes 0 / 2 run gemm_split16.hip's kernel (16x16x32 MFMAs; one epilogue for both outputs); the 32x32x16 kernels of this file are
// the experiments build's modes 10 (fp32 output) / 11 (split3 output through LDS) / 12 (split3 output, swapped roles)
int launch_gemm_split3_persist(const GemmArgs& a, int epi, int mode, void* ws_mem, hipStream_t s) {
    if (!gemm_split3_persist_ok(a) || ws_mem == nullptr) return -1;
    if (mode == 0 || mode == 2) {
        if ((mode == 0) != (a.c_split == nullptr)) return -1;
        return launch_split16_persist(a, epi, ws_mem, false, s);
    }
#ifndef THMR_EXPERIMENTS
    return -1;
#else
    mode -= 10;
    if ((mode == 0) != (a.c_split == nullptr) || mode < 0 || mode > 2 || (a.M % PBM) != 0) return -1;      // (the 32x32x16 kernels: whole row tiles only)
    // the 32x32x16 kernels of this file keep round 4's 0 / 1 flags (set by the producer, cleared by the consumer); the product kernel leaves
    // epochs in them: zero the flag words (not the control words) first
    if (hipMemsetAsync(reinterpret_cast<char*>(ws_mem) + (size_t)P_NWG * P_SLAB * 4, 0, P_NWG * sizeof(unsigned), s) != hipSuccess) return -2;
    if (a.c_split != nullptr && epi == EPI_BIAS_RESID) return -1;
    PersistWs ws;
    ws.part = reinterpret_cast<float*>(ws_mem);
    ws.flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws_mem) + (size_t)P_NWG * P_SLAB * 4);
    if (a.a_blk) {      // row-blocked A (fc2's operand): bias + residual, or no epilogue
        if (mode != 0) return -1;
        if (epi == EPI_BIAS_RESID) return launch_persist_cfg<EPI_BIAS_RESID, 0, true>(a, ws, s);
        if (epi == EPI_NONE) return launch_persist_cfg<EPI_NONE, 0, true>(a, ws, s);
        return -1;
    }
#define THMR_PERSIST_CASE(E, MD) \
    if (epi == E && mode == MD) return launch_persist_cfg<E, MD>(a, ws, s);
    THMR_PERSIST_CASE(EPI_NONE, 0)
    THMR_PERSIST_CASE(EPI_BIAS, 0)
    THMR_PERSIST_CASE(EPI_BIAS_RESID, 0)
    THMR_PERSIST_CASE(EPI_BIAS_QSCALE, 0)
    THMR_PERSIST_CASE(EPI_BIAS_GELU, 0)
    THMR_PERSIST_CASE(EPI_BIAS_GELU, 2)
    THMR_PERSIST_CASE(EPI_NONE, 2)
    THMR_PERSIST_CASE(EPI_BIAS_GELU, 1)      // split3 output through the LDS transposition: 796 vs 784 us on the fc1 shape (profiles/r4a_split3_gemm_persistent_b64.jsonl)
    THMR_PERSIST_CASE(EPI_NONE, 1)
#undef THMR_PERSIST_CASE
    return -1;
#endif  // THMR_EXPERIMENTS
}

bool gemm_split3_persist_narrow_ok(const GemmArgs& a) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;                                    // split-K: the stream's units are (tile, K slice) pairs, raw partial sums
    if (a.M <= 0 || a.N <= 0 || (a.K % (SBK * ks)) != 0 || a.K / ks < 3 * SBK || (a.N % 128) != 0) return false;      // (M may be ragged: a multiple of 192 rows)
    if ((int64_t)((a.M + 127) / 128) * (a.N / 128) * ks < P_NWG) return false;     // every lane's list holds >= 8 units: a range >= one unit
    if ((a.lda % 8) != 0 || (a.ldw % 8) != 0 || a.lda * 6 * 128 >= (int64_t(1) << 32) || a.ldw * 6 * 128 >= (int64_t(1) << 32)) return false;
    if (a.cs_out != nullptr || a.a_blk) return false;
    if (ks > 1 && (a.c_split != nullptr || a.bias != nullptr || a.resid != nullptr)) return false;
    if (a.c_split != nullptr && ((a.N % 8) != 0 || (a.ldcs % 8) != 0 || a.ldcs < a.N)) return false;
    return true;
}

int launch_gemm_split3_persist_narrow(const GemmArgs& a, int epi, void* ws_mem, hipStream_t s) {
    if (!gemm_split3_persist_narrow_ok(a) || ws_mem == nullptr || (a.ksplit > 1 && epi != EPI_NONE)) return -1;
    return launch_split16_persist(a, epi, ws_mem, true, s);
}

// split-K through the 128 x 128 stream: the arguments of launch_gemm_split3_splitk (raw partial sums of K slice sp into part[sp][M][N])
int launch_gemm_split3_splitk_stream(const GemmArgs& a0, int ksplit, float* part, void* ws_mem, hipStream_t s) {
    if (ksplit < 2 || part == nullptr) return -1;
    GemmArgs a = a0;
    a.ksplit = ksplit;
    a.C = part; a.ldc = a.N;
    a.bias = nullptr; a.resid = nullptr; a.ldr = 0;
    return launch_gemm_split3_persist_narrow(a, EPI_NONE, ws_mem, s);
}

// the workspace's error word (a consumer's bounded spin ran out): 0 = none.  Synchronises the stream.
int gemm_split3_persist_error(void* ws_mem, hipStream_t s, unsigned* err_out) {
    unsigned* flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws_mem) + (size_t)P_NWG * P_SLAB * 4);
    if (hipMemcpyAsync(err_out, flag + P_NWG, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess) return -2;
    if (hipStreamSynchronize(s) != hipSuccess) return -2;
    return 0;
}

// the host-mapped word a timed-out consumer also writes: *host_err_slot = its device-visible address (the slot must outlive the copy: the
// engine keeps it in its struct).  Enqueued on `s` behind the memset that zeroed the workspace.
int gemm_split3_persist_bind_host_err(void* ws_mem, unsigned* const* host_err_slot, hipStream_t s) {
    unsigned* flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws_mem) + (size_t)P_NWG * P_SLAB * 4);
    static_assert(sizeof(unsigned*) == 8, "the workspace keeps the address in two words");
    return hipMemcpyAsync(flag + P_NWG + 4, host_err_slot, sizeof(unsigned*), hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -2;
}

// grow-never workspace per (device, stream) for the stateless operators (thmr_op_gemm_split3 with a persistent variant)
void* gemm_split3_persist_op_ws(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, void*>, void*> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::unique_lock<std::mutex> lk(mu);
    void*& p = pool[{dev, (void*)s}];
    if (!p) {
        if (hipMalloc(&p, gemm_split3_persist_ws_bytes()) != hipSuccess) { p = nullptr; return nullptr; }
        if (hipMemset(p, 0, gemm_split3_persist_ws_bytes()) != hipSuccess) { (void)hipFree(p); p = nullptr; return nullptr; }
    }
    return p;
}
