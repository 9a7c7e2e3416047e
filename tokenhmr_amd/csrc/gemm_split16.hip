// The split3 GEMM on v_mfma_f32_16x16x32_bf16 — ONE kernel template for both decompositions: one workgroup per tile (any shape, split-K
// copies of the grid) and 256 persistent workgroups over a tile stream (gemm_split_persist.hip's scheme: a ragged last round split along
// K, the accumulators handed from one XCD's workgroup to the next through memory).  Same operand format, same LDS stage image and
// LDS-DMA staging as gemm_split.hip (vit.py:82-87,104-126 are the products it serves).
//
// Why another MFMA shape (scripts/micro/mfma_bf16_sustained.hip, profiles/r4j_mfma_bf16_16x16x32_vs_32x32x16_sustained.log).  The split3
// GEMMs run at the clock the part grants under 256 CUs of back-to-back bf16 MFMAs (1.4-1.5 GHz, HISTORY.md 11.2), so what counts is
// flops per joule — and on random operand bits the part SUSTAINS 2.09-2.12 PFLOP/s with 16x16x32 against 1.83 with 32x32x16 (register
// operands), 1.81 against 1.67 with this kernel's 24 fragment reads and 9 LDS-DMA copies per K tile beside them.
//
// Wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16 (NI = 4; NI = 2: 64 x 32, the 128 x 128 tile on eight waves — round 6, below); one MFMA consumes a whole 32-deep K tile: lane (l & 15, g = l >> 4) of an operand
// fragment holds the 16-byte chunk of row l & 15, k-group g — exactly one chunk of the split3 format.  Operand ROLES are swapped for every
// product (the MFMA's "A" is the weight fragment): an accumulator then holds, per lane, 4 CONSECUTIVE output columns n = 4 g ... 4 g + 3 of
// output row m = l & 15 — one 16-byte store per accumulator for an fp32 result (a 16-byte load for the residual, the bias), and for a
// split3 result four v_permlane16_swap per accumulator pair complete the 8-column chunks (lane pairs g, g + 1): no LDS epilogue exists.
// Per K tile and wave: 96 MFMAs (6 piece products x 16 tiles), 24 ds_read_b128, its share of the LDS-DMA copies, ONE barrier.
//
// Schedule of K tile t (LDS buffer t & 1), no VALU instruction in the loop.  Weight fragments are double-buffered in registers (48 x 2),
// activation fragments roll through one set (48): block mi = 0..3 multiplies activation tile mi with the four weight tiles (24 MFMAs) —
//   block 0 : + the read of tile t's last activation fragment (its registers were busy until block 3 of tile t - 1); then the barrier:
//             every wave's copies of tile t + 1 have landed and its reads of buffer t & 1 are done
//   blocks 1-3 : + the reads of tile t + 1's weight fragments (4 per block) and activation fragments 0-2 (one per block, into the registers
//             block mi - 1 just left) from the other buffer, and the copies of tile t + 2 into buffer t & 1
// so every read has at least a block (24 MFMAs + the SIMD partner's) to return and a copy a whole K tile to land.
// Per output element K is summed tile by tile, per tile the six piece products in the order lh hl mm mh hm hh (each over its 32 k inside one
// MFMA): the same for every instantiation, decomposition and batch size — bit-identical results among them (tests), and NOT bit-identical to
// the 32x32x16 kernels of gemm_split.hip (another grouping of k inside the MFMA), which are kept in the experiments build for the A/B.
#include <map>
#include <mutex>

#include "common.h"
#include "gemm_device.h"
#include "gemm_split_device.h"

namespace {

constexpr int QG = 32;                                     // stream lanes (workgroups) per XCD of the persistent decomposition
constexpr int QBM = 128;
constexpr int Q_SLAB = 128 * 256;                          // floats of one raw accumulator tile (128 KB)
constexpr int Q_NWG = 8 * QG;

// Hand-over protocol (round 5): a flag holds the EPOCH of the launch that published its slab.  Every workgroup reads the workspace's
// epoch word at its start (ep = epoch + 1, never 0), producers store ep, consumers wait for == ep; the last workgroup to finish — an arrival
// counter, one device-scope atomic per workgroup — advances the epoch word and zeroes the counter.  Nothing is "re-armed": a producer that
// publishes AFTER its consumer's bounded wait ran out leaves a flag of a finished epoch, which no later launch can mistake for its own (round
// 4's 0 / 1 flags stayed 1 in that case and fed every later launch the previous launch's accumulators).  It is device state, so a launch
// captured in a hipGraph replays correctly.  A timeout is written to the workspace's error word AND to the host-mapped word whose address the
// workspace carries (words W_HOST, W_HOST + 1: gemm_split3_persist_bind_host_err; null = none), which the engine tests on entry of the next call (engine.hip check_ready).
struct Ws16 {
    float* part;        // [8 * QG] slabs: slab (x, i) = the accumulators of the first part of the tile shared by lane i of XCDs x and x + 1
    unsigned* flag;     // [8 * QG] epochs; then the control words below
};
// When the flag goes out — measured (round 5, same-box interleaved against the round-4 library, fc2 class of a 64-crop step = 32 launches;
// profiles/r5d_*, r5e_*, r5g_*, r5h_*; round 4 = 20.5-20.9 ms box to box):
//   from inside the K loop, after the K tile that follows the slab stores — that tile's wait + barrier drains every wave's write-through
//       stores under its MFMAs — with the cold branch's operands (flag address, epoch) PARKED IN LDS: what ships, r4 + 0.14 ms
//   the same branch with its operands in registers (round 4's form): 22.9-24.0 ms in this code base — hipcc keeps them in scratch and
//       reloads two register pairs in every trip, a vmcnt(0) in front of the LDS-DMA pipeline (round 4's own source compiled without
//       that by luck of its register allocation; the first build of round 5 did not: fc2 +12 %, found by the first same-box A/B)
//   the flag right behind the slab stores + one explicit wait for them, nothing in the loop          r4 + 0.5 ms
//   the flag after the first K tile of the next segment, peeled out of the loop                      r4 + 1.5
//   the flag after the next segment's whole K loop (consumers of 1.9-tile ranges starve)             r4 + 1.2
//   the epoch protocol itself: free (0 / 1 flags without any arrival 21.30 ms, epoch advanced by one workgroup without arrival atomics
//       21.42, full protocol 21.46 at the same publish point); arrival at the workgroup's end instead of under the slab loads: the same
constexpr int W_ERR = Q_NWG, W_EPOCH = Q_NWG + 1, W_DONE = Q_NWG + 2, W_HOST = Q_NWG + 4;      // W_HOST: 8-byte aligned (the flag array is)

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// the epilogue arithmetic on the 4 consecutive columns n ... n + 3 of row m a lane holds
__device__ __forceinline__ f32x4 bias4(const GemmArgs& a, int n) {
    f32x4 b;
    if (n + 3 < a.N) b = *reinterpret_cast<const f32x4*>(a.bias + n);
    else
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = a.bias[min(n + r, a.N - 1)];
    return b;
}
template <int EPI>
__device__ __forceinline__ f32x4 epi4(const GemmArgs& a, f32x4 v, f32x4 b, int m, int n) {
    if constexpr (EPI == EPI_NONE) return v;
    v = v + b;
    if constexpr (EPI == EPI_BIAS_GELU) {
        const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
        v = f32x4{g0.x, g0.y, g1.x, g1.y};
    }
    if constexpr (EPI == EPI_BIAS_RELU) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    if constexpr (EPI == EPI_BIAS_RESID) {
        const float* rp = a.resid + (int64_t)m * a.ldr + n;
        f32x4 rs;
        if (n + 3 < a.N && (a.ldr & 3) == 0) rs = *reinterpret_cast<const f32x4*>(rp);
        else
#pragma unroll
            for (int r = 0; r < 4; ++r) rs[r] = n + r < a.N ? rp[r] : 0.f;
        v = rs + v;
    }
    if constexpr (EPI == EPI_BIAS_QSCALE)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (n + r < a.qcols) ? v[r] * a.qscale : v[r];
    if constexpr (EPI == EPI_BIAS_POS) {      // patch embed (vit.py:327): ((acc + bias) + pos[1 + token]) + pos[0]; resid = pos_embed [193][N], N % 4 == 0
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(a.resid + (int64_t)(1 + m % 192) * a.N + n);
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(a.resid + n);
        v = (v + p1) + p0;
    }
    return v;
}

// Rotation of a stage-image row's 12 chunks (rows of 192 bytes = 12 x 16 bytes; a lane reads chunk 3 g + pc of row l15).  gfx950 serves a
// ds_read_b128 in four groups of 16 lanes that are NOT 16 consecutive lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) — i.e. eight lanes of one g with rows in {0-3, 12-15} together with eight lanes of the next g with rows
// 4-11.  Four consecutive rows cover one residue class mod 4 of the 16-byte bank slots (192 = 12 slots: row bases 0, 12, 8, 4 mod 16),
// the class being (chunk + rot) mod 4, so a group is conflict-free iff its four row quads land in four different classes:
//   {rot(q0), rot(q3), rot(q1) + 3, rot(q2) + 3} and {rot(q1), rot(q2), rot(q0) + 3, rot(q3) + 3} both distinct mod 4  <=>  rot = (0, 2, 0, 2).
// (The first version rotated by (row >> 2) & 3, right for groups of 16 CONSECUTIVE lanes: every fragment read was a 2-way conflict —
// SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, 8 instead of 4 LDS cycles per read, profiles/r4ah_pmc_lds.json.)
__device__ __forceinline__ constexpr int rot16(int row) { return ((row >> 2) & 1) * 2; }

template <int WN, int NI = 4, int NST = 2>
constexpr int split16_lds_bytes() { return NST * (QBM * ROWB + WN * NI * 16 * ROWB); }    // NST stages of (A tile + W tile): 147,456 / 98,304 bytes at two, 147,456 for three 128 x 128 stages

// The kernel body (one workgroup of 2 WN waves; `smem` = its split16_lds_bytes<WN>() of LDS).  given_tile / half: -1 / -1 = the tile comes
// from the block index (the plain kernels below); the tail kernel passes the tile — and, for its 128 x 128 half tiles, which half of the
// 128 x 256 tile `given_tile` of the WIDE grid (tiles_n = that grid's) this workgroup computes.
// NI = 16-column MFMA tiles per wave along N.  (WN, NI) = (4, 4): 128 x 256 on 8 waves; (2, 4): 128 x 128 on 4 waves; (4, 2): 128 x 128 on
// EIGHT waves of 64 x 32 (round 6: 18 fragment reads per 48 MFMAs and wave; bit-identical, same K order per element).  Built on the
// hypothesis that the four-wave form (one wave per SIMD, matrix pipe ~45 % busy at few crops: fc1 at 4 crops = 240 workgroups x 40 K tiles
// in 62 us = 1.55 us per K tile against 0.7 us of MFMA issue, profiles/r6a_class_profile_small_batches.log) lacks a second wave per SIMD.
// Measured, it does not: the eight-wave form is 3-6 % SLOWER per call at 3-8 crops and equal as the half-tile tail at 64
// (profiles/r6d_ab_narrow8_or_tail8_*.json; launch_split3_tiles in gemm_split.hip keeps the four-wave form).  What the few-crop K loop
// waits for is its LDS-DMA copies — bytes in flight per CU, not waves (see the NST = 3 ring below).
// NST = LDS stages of the K ring (round 6).  Two (every instantiation up to round 5): the copies of K tile t + 1 are issued during tile t - 1 and
// waited for at tile t's barrier — ONE stage (48 KB of a 128 x 128 tile pair, 72 KB of a 128 x 256 one) in flight per CU for about one K-tile
// period.  With many tiles per CU that period is the MFMA time (2.2 us per wide K tile at 64 crops) and the round trip hides under it; at few
// crops the weights come cold from HBM to 90-240 workgroups, the round trip (~1.5 us loaded) IS the period, and the matrix pipe idles for
// half of it (1.55 us per 128 x 128 K tile against 0.7 us of MFMA issue at 4 crops).  Three stages (128 x 128 tile only: 3 x 48 KB = 144 KB)
// put the copies of tile t + 2 in flight as well: tile t's barrier waits with vmcnt(NP) — everything but the newest tile's copies — so a
// copy has two periods to land.  Fragment registers still alternate between two sets, so the K loop is unrolled by six (stage = t mod 3,
// set = t mod 2, all LDS offsets immediates).  Same K order per element: bit-identical.  Measured, whole path, same box, interleaved
// (profiles/r6e_ab_ring2_vs_ring3_*): 386 -> 459 crops/s at 3 crops, 497 -> 558 at 4, 473 -> 527 at 5, 645 -> 672 at 8, 770 -> 776 at 16.
// The 128 x 256 tile has no LDS for a third stage (3 x 72 KB).  Tried for it and not kept (commit 6294ec7, profiles/r6f_*): warming L2 with
// the weight tile several K tiles ahead of the copy cursor — one dword LDS-DMA touch per lane and K tile into a sink — is SLOWER at every
// batch size (+2.5 % per call at 8 crops, +7.8 % at 4, +2.7 % at 16, +3.8 % at 32, +2.5 % at 64): the extra copy and the later arrival of
// the real ones cost more than an L2 hit saves.
// FRONT (round 6): where in K tile t the copies of tile t + NST are issued.  false: spread over blocks 1-3, one every few MFMAs (the last
// ones ~80 % into the tile: they have under half a K-tile period to land before the next tile's barrier — plenty where the period is the MFMA
// time and the round trip short: 64 crops).  true: ALL of them in block 1, right behind the barrier that freed their stage: every copy has
// ~0.95 of a period.  For the few-crop calls, whose period IS the round trip of cold weights (see NST).  Same results either way.
template <int WN, int NI, int EPI, bool PERSIST, bool ABLK, int NST = 2, bool FRONT = false>
__device__ __forceinline__ void split16_body(GemmArgs& a, int tiles_m, int tiles_n, int nwg, const Ws16& ws, char* smem, int given_tile, int half) {
    constexpr int NW = 2 * WN, BN = WN * NI * 16;
    static_assert((NST == 2 && (!PERSIST || WN == 4)) || (NST == 3 && WN * NI == 8 && (!PERSIST || WN == 2)),
                  "two stages (persistent: the 128 x 256 tile), or three under the 128 x 128 tile (persistent: its four-wave form)");
    static_assert(NI == 4 || NI == 2, "wave tile 64 x 64 or 64 x 32");
    static_assert(!PERSIST || NI == 4, "the persistent decomposition uses the 64 x 64 wave tile");
    constexpr int A_Q = QBM * SLOTS / 64, B_Q = BN * SLOTS / 64;
    static_assert(A_Q % NW == 0 && B_Q % NW == 0, "tile / waves mismatch");
    constexpr int A_P = A_Q / NW, B_P = B_Q / NW, NP = A_P + B_P;      // copies per wave and K tile: 3 + 6 (8 waves), 6 + 6 (4 waves)
    constexpr int A_STAGE = QBM * ROWB, B_STAGE = BN * ROWB;
    constexpr int A_KSTEP = ABLK ? SLOTS * 512 : ROWB;                 // bytes a K tile advances the A source by
    static_assert(NST * (A_STAGE + B_STAGE) <= 160 * 1024 && NST * (A_STAGE + B_STAGE) == split16_lds_bytes<WN, NI, NST>(), "LDS");
    static_assert((NST - 1) * A_STAGE + 3 * 16 * ROWB < 65536 && (NST - 1) * B_STAGE + (NI - 1) * 16 * ROWB < 65536, "fragment offsets must stay 16-bit immediates");

    char* As = smem;
    char* Bs = smem + NST * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * (NI * 16);
    const int l15 = lane & 15, g = lane >> 4;
    // K tiles of one unit of work.  The 128 x 128 stream also takes split-K launches (round 6): its units are (tile, K slice) pairs, unit u =
    // slice u / tiles of tile u % tiles — the same raw partial sums into part[slice] as the grid copies of the per-tile kernel, bit for bit
    const int ksn = (PERSIST && NST == 3 && a.ksplit > 1) ? a.ksplit : 1;
    const int nk_all = a.K / SBK / ksn;
    float* const c_base = a.C;

    // ---- segments: (tile, K tiles [kb, ke), kind 0 whole / 1 first part: store the accumulators / 2 rest: start from stored accumulators)
    int xcd = 0, ln = 0, j0 = 0, k0 = 0, j1 = 0, k1 = 0, has_pre = 0, has_post = 0, jf0 = 0, nfull = 1, nseg = 1;
    int pt_tile = 0, pt_kb = 0, pt_ke = nk_all;                        // the one segment of the per-tile decomposition
    if constexpr (PERSIST) {
        xcd = blockIdx.x & 7;
        ln = blockIdx.x >> 3;
        const int T = (tiles_m * tiles_n * ksn - ln + QG - 1) / QG;    // >= 8 (launcher): a range is at least one unit long
        const int S0 = (int)((int64_t)xcd * T * nk_all / 8), S1 = (int)((int64_t)(xcd + 1) * T * nk_all / 8);
        j0 = S0 / nk_all; k0 = S0 - j0 * nk_all;
        j1 = (S1 - 1) / nk_all; k1 = S1 - j1 * nk_all;
        has_pre = k1 < nk_all ? 1 : 0;
        has_post = k0 > 0 ? 1 : 0;
        jf0 = j0 + has_post;
        nfull = max(j1 - has_pre - jf0 + 1, 0);
        nseg = has_pre + nfull + has_post;
    } else {
        int logical = given_tile >= 0 ? given_tile : logical_block(nwg, 0);
        if (a.ksplit > 1) {      // copy ksp of the tile grid reduces K slice ksp into part[ksp] (raw, no epilogue)
            const int tiles = tiles_m * tiles_n, ksp = logical / tiles;
            logical -= ksp * tiles;
            const int nks = nk_all / a.ksplit;
            pt_kb = ksp * nks;
            pt_ke = pt_kb + nks;
            a.C += (int64_t)ksp * a.M * a.ldc;
        }
        pt_tile = logical;
    }
    auto seg_of = [&](int n, int& j, int& kb, int& ke, int& kind) {
        if constexpr (PERSIST) {
            const int m = n - has_pre;
            if (has_pre && n == 0) { j = j1; kb = 0; ke = k1; kind = 1; }
            else if (m < nfull) { j = jf0 + m; kb = 0; ke = nk_all; kind = 0; }
            else { j = j0; kb = k0; ke = nk_all; kind = 2; }
        } else {
            j = pt_tile; kb = pt_kb; ke = pt_ke; kind = 0;
        }
    };
    // K slice of stream unit j of this lane (0 without split-K)
    auto slice_of = [&](int j) { return ksn > 1 ? (j * QG + ln) / (tiles_m * tiles_n) : 0; };
    auto tile_of = [&](int j, int& bm0, int& bn0) {
        int tm, tn;
        int t = PERSIST ? j * QG + ln : j;
        if (ksn > 1) t -= slice_of(j) * (tiles_m * tiles_n);
        tile_coords(tiles_m, tiles_n, t, tm, tn);
        bm0 = __builtin_amdgcn_readfirstlane(tm * QBM);
        bn0 = __builtin_amdgcn_readfirstlane(half >= 0 ? tn * 256 + half * 128 : tn * BN);
    };

    // ---- copies: wave instruction q = wave + i NW of an operand fills LDS chunks 64 q ... 64 q + 63 of its stage (rows past the edge clamped)
    const int64_t arow = a.lda * 6, wrow = a.ldw * 6;
    uint32_t Aoff[A_P], Woff[B_P];
    auto set_offsets = [&](int bm0, int bn0) {
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int c = (wave + i * NW) * 64 + lane;
            if constexpr (ABLK) {
                const int row = (c / 384) * 32 + (c & 31), rg = min(bm0 + row, a.M - 1) - bm0;
                Aoff[i] = (uint32_t)(rg >> 5) * (uint32_t)(a.lda * 192) + (uint32_t)((c % 384) >> 5) * 512u + (uint32_t)(rg & 31) * 16u;
            } else {
                const int row = c / SLOTS, slot = c - row * SLOTS;
                Aoff[i] = (uint32_t)(min(bm0 + row, a.M - 1) - bm0) * (uint32_t)arow + (uint32_t)((slot + SLOTS - rot16(row)) % SLOTS) * 16u;
            }
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int c = (wave + i * NW) * 64 + lane, row = c / SLOTS, slot = c - row * SLOTS;
            Woff[i] = (uint32_t)(min(bn0 + row, a.N - 1) - bn0) * (uint32_t)wrow + (uint32_t)((slot + SLOTS - rot16(row)) % SLOTS) * 16u;
        }
    };
    // fetch cursor: the K tile the NEXT copy brings in (two ahead of the multiply), wave-uniform
    int fn = 0, fk = 0, fke = 0;
    const char *fA = nullptr, *fW = nullptr;
    auto fetch_seg = [&](int n) {
        int j, kb, ke, kind, bm0, bn0;
        seg_of(n, j, kb, ke, kind);
        tile_of(j, bm0, bn0);
        fk = kb;
        fke = ke;
        const int k_first = kb + (ksn > 1 ? slice_of(j) * nk_all : 0);      // the unit's K slice starts at slice * nk_all
        fA = reinterpret_cast<const char*>(a.A) + (int64_t)bm0 * arow + (int64_t)k_first * A_KSTEP;
        fW = reinterpret_cast<const char*>(a.W) + (int64_t)bn0 * wrow + (int64_t)k_first * ROWB;
        // the streams take a ragged M (an odd number of crops: 192 B rows): the row clamps of the LAST row tile differ, so the copy offsets
        // follow the segment the cursor enters (a cold branch of the K loop; the per-tile instantiations keep one set of offsets)
        if constexpr (PERSIST) {
            if ((a.M & (QBM - 1)) != 0) set_offsets(bm0, bn0);
        }
    };
    auto fetch_advance = [&]() {
        if (fk + 1 < fke) { ++fk; fA += A_KSTEP; fW += ROWB; }
        else if (PERSIST && fn + 1 < nseg) fetch_seg(++fn);
        // else: past the end: the last K tile is copied again, into a buffer nobody reads any more
    };
    auto dma_piece = [&](int buf, int p) {
        if (p < A_P) dma16_saddr(fA, Aoff[p], lds_addr_b(As + buf * A_STAGE + (wave + p * NW) * 1024));
        else dma16_saddr(fW, Woff[p - A_P], lds_addr_b(Bs + buf * B_STAGE + (wave + (p - A_P) * NW) * 1024));
    };

    // ---- fragments: lane (l15, g) reads chunk (k-group g, piece pc) of row l15 of a 16-row tile.  Row-major stage image: physical slot
    // (3 g + pc + rot16(row)) % 12, rot16(row) = rot16(l15) for every tile (tile offsets are multiples of 16).
    // Each operand has ONE per-lane LDS address per piece with the stage base and the wave's tile offset folded in, so that every fragment read
    // of the K loop is {that register} + {immediate below 64 KB}: A: buf * 24 KB + tile offset <= 33,792; W: buf * 48 KB + ni * 3 KB <=
    // 58,368.  (With the wave-uniform parts left to the compiler it kept three registers and re-added the part that exceeds the 16-bit
    // offset field inside the loop — three VALU instructions per trip in the persistent instantiations, round 5.)
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
    uint32_t fow[3], foa[3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        const uint32_t rowmaj = (uint32_t)l15 * ROWB + (uint32_t)((3 * g + pc + rot16(l15)) % SLOTS) * 16u;
        fow[pc] = lds_addr_b(Bs) + (uint32_t)(wn0 * ROWB) + rowmaj;
        foa[pc] = lds_addr_b(As) + (uint32_t)(wm0 * ROWB) + (ABLK ? (uint32_t)((3 * g + pc) * 512 + l15 * 16) : rowmaj);      // (row-blocked A: wm0 / 32 blocks of 12 x 512 bytes — the same offset)
    }
    bf16x8 af[4][3], wf[2][NI][3];                                     // activation fragments (rolling), weight fragments of this / the next K tile
    auto read_a = [&](int buf, int mi, int pc) {
        const int off = ABLK ? (mi >> 1) * (SLOTS * 512) + (mi & 1) * 256 : mi * 16 * ROWB;
        af[mi][pc] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(foa[pc] + (uint32_t)(buf * A_STAGE + off)));
    };
    auto read_w = [&](int buf, int set, int ni, int pc) {
        wf[set][ni][pc] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(fow[pc] + (uint32_t)(buf * B_STAGE + ni * 16 * ROWB)));
    };
    f32x4 acc[4][NI];

    // one K tile out of buffer `buf` (weight fragment set `buf`)
    constexpr int NRB = 3 + NI;                                        // reads per block 1-3: one activation tile (3 pieces) + NI of the 3 NI weight fragments
    constexpr int DB = (NP + 2) / 3;                                   // copies per block 1-3
    static_assert(NRB + (FRONT ? NP : DB) <= NPROD * NI, "block too small for the staging interleave");
    static_assert(!FRONT || !PERSIST, "front-loaded copies: the per-tile decomposition");
    int par = 0;                 // fragment set (and, with two stages, LDS buffer) of the next K tile
    int pub_pending = 0;         // the slab's stores are issued; the flag goes out after the next K tile's barrier (every wave has drained them)
    // wait for every copy of this wave but the newest `keep` tiles' (the compiler does not count LDS-DMA copies: explicit), + its LDS reads; barrier
    auto ring_wait_barrier = [&](auto keepc) {
        constexpr int keep = decltype(keepc){};
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(keep * NP) : "memory");
    };
    // K tile t out of LDS stage `buf` = t mod NST with weight fragment set `set` = t mod 2: reads tile t + 1 from stage buf + 1, copies tile t + NST into stage buf
    auto ktile = [&](auto bufc, auto setc) {
        constexpr int buf = decltype(bufc){}, set = decltype(setc){}, nbuf = (buf + 1) % NST;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int p = 0; p < NPROD; ++p)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int idx = p * NI + ni;
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][ni][piece_w(p)], af[mi][piece_a(p)], acc[mi][ni], 0, 0, 0);
                    bool any = false;
                    if (mi == 0) {
                        if (idx < 3) { read_a(buf, 3, idx); any = true; }                       // this tile's last activation fragment
                    } else {
                        if (idx < 3) { read_a(nbuf, mi - 1, idx); any = true; }                 // next tile: activation tile mi - 1 ...
                        else if (idx < NRB) {                                                   // ... and NI of its 3 NI weight fragments
                            const int q = (mi - 1) * NI + (idx - 3);
                            read_w(nbuf, set ^ 1, q / 3, q % 3);
                            any = true;
                        } else if (!FRONT && idx - NRB < DB && (mi - 1) * DB + (idx - NRB) < NP) {
                            dma_piece(buf, (mi - 1) * DB + (idx - NRB));                        // K tile t + NST, into this stage
                            any = true;
                        } else if (FRONT && mi == 1 && idx - NRB < NP) {
                            dma_piece(buf, idx - NRB);                                          // ... all of it right behind the barrier
                            any = true;
                        }
                    }
                    if (any) __builtin_amdgcn_sched_barrier(0);
                }
            if (mi == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NST == 2) dma_wait_barrier();
                else if constexpr (PERSIST) {
                    // the K tile behind a slab store drains EVERYTHING (the publish that follows it relies on the stores having left, and
                    // stores need not retire in order with the copies)
                    if (pub_pending) dma_wait_barrier();
                    else ring_wait_barrier(IntC<NST - 2>{});
                } else ring_wait_barrier(IntC<NST - 2>{});           // tile t + 1 has landed; the copies of t + 2 may still be in flight
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_advance();
    };
    // The persistent three-stage form.  Its segments start at whatever point of the ring the previous one ended on, and stage x fragment set
    // has period six: with both compile-time (the six-fold unrolled loop of the per-tile kernel) every segment would begin and end with up to
    // five single steps out of a switch, whose register assignments the allocator reconciles with accumulation-register moves — measured as a
    // first version: 1.45 us per K tile against the per-tile kernel's 1.15.  Here the STAGE is a run-time scalar (three stage offsets added to
    // the per-lane fragment bases once per K tile: nine vector adds against 96 MFMAs — the one K loop of this file that carries vector
    // instructions) and only the fragment set stays compile-time: the loop is the two-stage kernel's, two K tiles per trip.
    int stage = 0;               // (K tiles this workgroup has multiplied) mod NST
    auto ktile_rt = [&](auto setc) {
        constexpr int set = decltype(setc){};
        static_assert(!ABLK, "row-major A only");
        const int nstage = stage == NST - 1 ? 0 : stage + 1;
        uint32_t ac[3], an[3], wn[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            ac[pc] = foa[pc] + (uint32_t)(stage * A_STAGE);
            an[pc] = foa[pc] + (uint32_t)(nstage * A_STAGE);
            wn[pc] = fow[pc] + (uint32_t)(nstage * B_STAGE);
        }
        const uint32_t dA = lds_addr_b(As) + (uint32_t)(stage * A_STAGE), dB = lds_addr_b(Bs) + (uint32_t)(stage * B_STAGE);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int p = 0; p < NPROD; ++p)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int idx = p * NI + ni;
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][ni][piece_w(p)], af[mi][piece_a(p)], acc[mi][ni], 0, 0, 0);
                    bool any = false;
                    if (mi == 0) {
                        if (idx < 3) { af[3][idx] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(ac[idx] + (uint32_t)(3 * 16 * ROWB))); any = true; }
                    } else {
                        if (idx < 3) { af[mi - 1][idx] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(an[idx] + (uint32_t)((mi - 1) * 16 * ROWB))); any = true; }
                        else if (idx < NRB) {
                            const int q = (mi - 1) * NI + (idx - 3);
                            wf[set ^ 1][q / 3][q % 3] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(wn[q % 3] + (uint32_t)((q / 3) * 16 * ROWB)));
                            any = true;
                        } else if (idx - NRB < DB && (mi - 1) * DB + (idx - NRB) < NP) {
                            const int pp = (mi - 1) * DB + (idx - NRB);
                            if (pp < A_P) dma16_saddr(fA, Aoff[pp], dA + (uint32_t)((wave + pp * NW) * 1024));
                            else dma16_saddr(fW, Woff[pp - A_P], dB + (uint32_t)((wave + (pp - A_P) * NW) * 1024));
                            any = true;
                        }
                    }
                    if (any) __builtin_amdgcn_sched_barrier(0);
                }
            if (mi == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (pub_pending) dma_wait_barrier();                 // behind a slab store: drain everything (see ktile)
                else ring_wait_barrier(IntC<NST - 2>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_advance();
        stage = nstage;
    };
    // fill: the first two K tiles of segment n into buffers 0 / 1, every fragment of tile 0 into registers
    auto fill = [&](int n) {
        fetch_seg(n);
        fn = n;
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(0, p);
        fetch_advance();
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(1, p);
        fetch_advance();
        if constexpr (NST == 3) {
#pragma unroll
            for (int p = 0; p < NP; ++p) dma_piece(2, p);
            fetch_advance();
            ring_wait_barrier(IntC<2>{});                            // tile 0 has landed
        } else {
            dma_wait_barrier();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                read_a(0, t, pc);
                if (t < NI) read_w(0, 0, t, pc);
            }
    };

    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ws.part, 0, Q_NWG * Q_SLAB * 4, 0x00020000);
    const uint32_t slab_lane = (uint32_t)(wave * 16 * 1024 + lane * 16);       // a wave's 16 accumulators of 1 KiB each
    __shared__ uint32_t pub_slot[4];          // {flag address lo, hi, epoch}: what the cold publish branch needs, parked in LDS so that NOTHING
                                              // but `pub_pending` is live across the K loop on its behalf (see the measurements above)
    // this launch's epoch (see Ws16): read once (requested here, consumed after the first fill so that its round trip runs under the
    // prologue's copies); every workgroup reads it before any workgroup can have arrived
    unsigned ep_raw = 0u, ep = 1u;
    if constexpr (PERSIST) ep_raw = __hip_atomic_load(ws.flag + W_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // arrival (thread 0): a workgroup arrives once it no longer needs the epoch — after its consumer wait, or at its end if it has none; the
    // last of the launch's Q_NWG arrivals closes the epoch.  `arrived_old` = the counter value this workgroup's arrival returned.
    auto close_epoch_if_last = [&](unsigned arrived_old) {
        if (arrived_old == (unsigned)(Q_NWG - 1)) {
            __hip_atomic_store(ws.flag + W_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws.flag + W_EPOCH, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    {
        int j, kb, ke, kind, bm0, bn0;
        seg_of(0, j, kb, ke, kind);
        tile_of(j, bm0, bn0);
        set_offsets(bm0, bn0);          // persistent: M % 128 == 0 and N % 256 == 0, the offsets are the same for every tile
    }
    fill(0);
    if constexpr (PERSIST) {
        ep = __builtin_amdgcn_readfirstlane(ep_raw) + 1u;
        if (ep == 0u) ep = 1u;
        if (tid == 0) {
            const uint64_t fa = (uint64_t)(uintptr_t)(ws.flag + xcd * QG + ln);
            pub_slot[0] = (uint32_t)fa; pub_slot[1] = (uint32_t)(fa >> 32); pub_slot[2] = ep;
        }
    }
    for (int n = 0; n < nseg; ++n) {
        int j, kb, ke, kind, bm0, bn0;
        seg_of(n, j, kb, ke, kind);
        tile_of(j, bm0, bn0);
        if constexpr (PERSIST && NST == 3) {
            if (ksn > 1) a.C = c_base + (int64_t)slice_of(j) * a.M * a.ldc;      // this unit's partial-sum plane
        }
        if (PERSIST && kind == 2) {
            // the accumulators of K tiles [0, kb) from lane ln of the previous XCD: one thread polls one word, then sc1 loads
            unsigned arrived_old = 0u;
            if (tid == 0) {
                unsigned spins = 0;
                unsigned* f = ws.flag + (xcd - 1) * QG + ln;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ep) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) {                        // ~0.5 s: report (device word + host-mapped word), never hang
                        __hip_atomic_store(ws.flag + W_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unsigned* const he = *reinterpret_cast<unsigned* const*>(ws.flag + W_HOST);
                        if (he) __hip_atomic_store(he, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
                // this workgroup's last use of the epoch: arrive now — the atomic's round trip runs under the slab loads below
                arrived_old = __hip_atomic_fetch_add(ws.flag + W_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_barrier" ::: "memory");
            const uint32_t base = (uint32_t)((xcd - 1) * QG + ln) * (uint32_t)(Q_SLAB * 4) + slab_lane;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (uint32_t)((mi * 4 + ni) * 1024), 0, 16);
                    acc[mi][ni] = f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
                }
            if (tid == 0) close_epoch_if_last(arrived_old);
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // the publish: a scalar test per K tile; the cold branch reads its operands from LDS (thread 0 wrote them, thread 0 reads them)
        auto after_tile = [&]() {
            if constexpr (PERSIST) {
                if (pub_pending) {
                    if (tid == 0) {
                        const uint32_t lo = pub_slot[0], hi = pub_slot[1], e = pub_slot[2];
                        __hip_atomic_store(reinterpret_cast<unsigned*>((uintptr_t)(((uint64_t)hi << 32) | lo)), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    pub_pending = 0;
                }
            }
        };
        int cnt = ke - kb;
        if constexpr (NST == 3 && !PERSIST) {
            // one segment per workgroup: stage = t mod 3, fragment set = t mod 2
            for (; cnt >= 6; cnt -= 6) {
                ktile(IntC<0>{}, IntC<0>{});
                ktile(IntC<1>{}, IntC<1>{});
                ktile(IntC<2>{}, IntC<0>{});
                ktile(IntC<0>{}, IntC<1>{});
                ktile(IntC<1>{}, IntC<0>{});
                ktile(IntC<2>{}, IntC<1>{});
            }
            if (cnt > 0) ktile(IntC<0>{}, IntC<0>{});
            if (cnt > 1) ktile(IntC<1>{}, IntC<1>{});
            if (cnt > 2) ktile(IntC<2>{}, IntC<0>{});
            if (cnt > 3) ktile(IntC<0>{}, IntC<1>{});
            if (cnt > 4) ktile(IntC<1>{}, IntC<0>{});
        } else if constexpr (NST == 3) {
            // persistent: run-time stage, fragment set = par (the two-stage kernel's loop)
            if (cnt > 0 && par) {
                ktile_rt(IntC<1>{});
                after_tile();
                --cnt; par = 0;
            }
            for (; cnt >= 2; cnt -= 2) {
                ktile_rt(IntC<0>{});
                after_tile();
                ktile_rt(IntC<1>{});
                after_tile();
            }
            if (cnt) { ktile_rt(IntC<0>{}); after_tile(); par = 1; }
        } else {
            if (cnt > 0 && par) {
                ktile(IntC<1>{}, IntC<1>{});
                after_tile();
                --cnt; par = 0;
            }
            for (; cnt >= 2; cnt -= 2) {
                ktile(IntC<0>{}, IntC<0>{});
                after_tile();
                ktile(IntC<1>{}, IntC<1>{});
                after_tile();
            }
            if (cnt) { ktile(IntC<0>{}, IntC<0>{}); after_tile(); par = 1; }
        }

        const int m0 = bm0 + wm0, n0 = bn0 + wn0;
        if (PERSIST && kind == 1) {
            // raw accumulators -> slab (xcd, ln), write-through; published by `after_tile` once the next K tile's wait + barrier has passed
            const uint32_t base = (uint32_t)(xcd * QG + ln) * (uint32_t)(Q_SLAB * 4) + slab_lane;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const u32x4 v = {__float_as_uint(acc[mi][ni][0]), __float_as_uint(acc[mi][ni][1]), __float_as_uint(acc[mi][ni][2]),
                                     __float_as_uint(acc[mi][ni][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, base + (uint32_t)((mi * 4 + ni) * 1024), 0, 16);
                }
            pub_pending = 1;
        } else if (a.c_split != nullptr) {
            // the result as a split3 operand: bias + activation, then lanes (g, g + 1) complete each other's 8-column chunks
            // (v_permlane16_swap of accumulator pair (ni, ni + 1): even g ends with a chunk of tile ni, odd g with one of tile ni + 1).
            // One address per lane; the 24 stores sit at offsets that are the same for every lane (row step, row-block step, chunk step).
            const bool blk = a.cs_blk != 0;
            const int mb = m0 + l15, nb = n0 + (g & 1) * 16 + (g & 2) * 4;         // this lane's first row; first of its 8 columns for ni = 0
            char* cb = reinterpret_cast<char*>(a.c_split) + split3_chunk_off(a.ldcs, mb, nb >> 3, 0, blk);
            // mi -> row + 16 mi: row-blocked, 256 bytes inside the 32-row block for odd mi and one block (ld * 192 bytes) per two; row-major 16 rows
            const int64_t step_m1 = blk ? 256 : a.ldcs * 96, step_m2 = blk ? a.ldcs * 192 : a.ldcs * 192;
            const int step_n = blk ? 4 * 3 * 512 : 4 * 48, step_p = blk ? 512 : 16;            // ni -> + 32 columns = 4 k-groups; piece
            f32x4 bv[NI];                                  // the bias of this lane's columns, once (the stores below would force a reload per row tile)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bv[ni] = EPI == EPI_NONE ? f32x4{0.f, 0.f, 0.f, 0.f} : bias4(a, n0 + ni * 16 + 4 * g);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = mb + mi * 16, mc = min(m, a.M - 1);
                char* cm = cb + (mi & 1) * step_m1 + (mi >> 1) * step_m2;
#pragma unroll
                for (int ni = 0; ni < NI; ni += 2) {
                    f32x4 x = epi4<EPI>(a, acc[mi][ni], bv[ni], mc, n0 + ni * 16 + 4 * g);
                    f32x4 y = epi4<EPI>(a, acc[mi][ni + 1], bv[ni + 1], mc, n0 + (ni + 1) * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
                        x[r] = __uint_as_float(sw.x);
                        y[r] = __uint_as_float(sw.y);
                    }
                    uint32_t H[4], M[4], L[4];
                    split3_pair(x[0], x[1], H[0], M[0], L[0]);
                    split3_pair(x[2], x[3], H[1], M[1], L[1]);
                    split3_pair(y[0], y[1], H[2], M[2], L[2]);
                    split3_pair(y[2], y[3], H[3], M[3], L[3]);
                    if (m < a.M && nb + ni * 16 < a.N) {
                        char* o = cm + (ni >> 1) * step_n;
                        *reinterpret_cast<u32x4*>(o) = u32x4{H[0], H[1], H[2], H[3]};
                        *reinterpret_cast<u32x4*>(o + step_p) = u32x4{M[0], M[1], M[2], M[3]};
                        *reinterpret_cast<u32x4*>(o + 2 * step_p) = u32x4{L[0], L[1], L[2], L[3]};
                    }
                }
            }
        } else {
            const bool vec = (a.ldc & 3) == 0 && ((uintptr_t)a.C & 15) == 0;
            f32x4 bv[NI];
            if constexpr (!PERSIST)                       // (the persistent instantiations are short of registers here: they load it per row tile)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bv[ni] = EPI == EPI_NONE ? f32x4{0.f, 0.f, 0.f, 0.f} : bias4(a, n0 + ni * 16 + 4 * g);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + mi * 16 + l15;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + ni * 16 + 4 * g;
                    if constexpr (PERSIST) bv[ni] = EPI == EPI_NONE ? f32x4{0.f, 0.f, 0.f, 0.f} : bias4(a, n);
                    const f32x4 v = epi4<EPI>(a, acc[mi][ni], bv[ni], min(m, a.M - 1), n);
                    if (m < a.M) {
                        float* cp = a.C + (int64_t)m * a.ldc + n;
                        if (vec && n + 3 < a.N) *reinterpret_cast<f32x4*>(cp) = v;
                        else
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n + r < a.N) cp[r] = v[r];
                    }
                }
            }
        }
        if constexpr (PERSIST) {
            // The next tile's fragments were prefetched by the last K tile above; they are read AGAIN here, after the epilogue, so that
            // they are dead across it: 84 registers the GELU + split3 epilogue otherwise spills around (113-115 scratch registers in the
            // first build, and the persistent fc1 at 728 us against the per-tile kernel's 702, profiles/r4k_split3_gemm_b64_mfma16.jsonl).
            // 21 LDS reads per tile; the copies they read landed before the last K tile's barrier.
            if (n + 1 < nseg) {
                auto refill = [&](auto bufc) {
                    constexpr int buf = decltype(bufc){};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            if (t < 3) read_a(buf, t, pc);
                            if (t < NI) read_w(buf, buf, t, pc);
                        }
                };
                if constexpr (NST == 3) {
                    auto refill_rt = [&](auto setc) {
                        constexpr int set = decltype(setc){};
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) {
                                if (t < 3) af[t][pc] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(foa[pc] + (uint32_t)(stage * A_STAGE + t * 16 * ROWB)));
                                wf[set][t][pc] = *reinterpret_cast<lds_frag_ptr>((uintptr_t)(fow[pc] + (uint32_t)(stage * B_STAGE + t * 16 * ROWB)));
                            }
                    };
                    if (par) refill_rt(IntC<1>{}); else refill_rt(IntC<0>{});
                } else if (par) refill(IntC<1>{}); else refill(IntC<0>{});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no LDS-DMA write may outlive the workgroup's LDS allocation
    if constexpr (PERSIST) {
        if (pub_pending) {                                             // (a first part is never a range's last segment: >= 8 tiles per lane; kept for safety)
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ws.flag + xcd * QG + ln, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // a workgroup without a consumer segment (first XCD; ranges that start on a tile boundary) arrives here
        if (!has_post && tid == 0) close_epoch_if_last(__hip_atomic_fetch_add(ws.flag + W_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}

// WN = waves along N (4: 128 x 256 tile, 8 waves; 2: 128 x 128, 4 waves).  PERSIST: 256 workgroups over a tile stream (WN = 4; M % 128 == 0,
// N % 256 == 0) instead of one workgroup per (tile, K slice).  ABLK: A is a row-blocked split3 operand (GemmArgs::a_blk).
// (__launch_bounds__(512) for the 4-wave instantiation too: told that a workgroup has 256 threads hipcc budgets 512 registers per lane,
// parks fragments in AGPRs and copies them back inside the K loop — ~40 v_accvgpr moves per 192 MFMAs.  A bound of 512 threads = 256
// registers gives it the 8-wave instantiation's allocation: none.)
template <int WN, int NI, int EPI, bool PERSIST, bool ABLK, int NST = 2, bool FRONT = false>
__global__ __launch_bounds__(512) void gemm_split16_kernel(GemmArgs a, int tiles_m, int tiles_n, int nwg, Ws16 ws) {
    __shared__ __attribute__((aligned(16))) char smem[split16_lds_bytes<WN, NI, NST>()];
    split16_body<WN, NI, EPI, PERSIST, ABLK, NST, FRONT>(a, tiles_m, tiles_n, nwg, ws, smem, -1, -1);
}

// The persistent 128 x 128 stream (four waves, three-stage ring): its own entry point with __launch_bounds__(256) — under the 512-thread
// bound of the kernel above (256 registers per lane) the segment bookkeeping on top of the ring's 250 registers spills 1200-1380 registers
// to scratch; with 256 threads declared the allocator may use the accumulation registers as well (one wave per SIMD: 512 per lane).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_split16_persist_narrow_kernel(GemmArgs a, int tiles_m, int tiles_n, int nwg, Ws16 ws) {
    __shared__ __attribute__((aligned(16))) char smem[split16_lds_bytes<2, 4, 3>()];
    split16_body<2, 4, EPI, true, false, 3>(a, tiles_m, tiles_n, nwg, ws, smem, -1, -1);
}

// One workgroup per 128 x 256 tile EXCEPT the ragged last round, which runs as 128 x 128 half tiles (round 5).  With T
// tiles on C CUs (one 147 KB workgroup per CU) the last round holds T mod C tiles and the other CUs idle for a whole tile time: fc1 of a
// 64-crop batch is 1920 tiles = 7.5 rounds on 256 CUs (6.25 % of the launch idle).  Block b runs on XCD b % 8 and XCD x owns the contiguous
// logical tiles [x q, (x + 1) q), q = T / 8, dispatched in order `within` = b / 8.  Here the first tail_from = 32 floor(q / 32) of them stay
// 8-wave workgroups on wide tiles; each of the remaining rem = q mod 32 <= 16 tiles becomes TWO blocks whose waves 0-3 run the 4-wave body on
// one 128 x 128 half (waves 4-7 exit at once): 2 rem <= 32 blocks per XCD, one per CU, each with half the matrix work — the last round takes
// about half a tile time instead of a whole one.  Same K order per element as every other instantiation: bit-identical results (tests).
template <int EPI, bool ABLK, bool TAIL8>
__global__ __launch_bounds__(512) void gemm_split16_tail_kernel(GemmArgs a, int tiles_m, int tiles_n, int q, int tail_from) {
    __shared__ __attribute__((aligned(16))) char smem[split16_lds_bytes<4>()];
    const Ws16 none{nullptr, nullptr};
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;           // workgroup-uniform
    if (within < tail_from) {
        split16_body<4, 4, EPI, false, ABLK>(a, tiles_m, tiles_n, 0, none, smem, xcd * q + within, -1);
    } else {
        const int h = within - tail_from;
        if constexpr (TAIL8) {                                          // round 6: the half tile on all eight waves (64 x 32 wave tiles)
            split16_body<4, 2, EPI, false, ABLK>(a, tiles_m, tiles_n, 0, none, smem, xcd * q + tail_from + (h >> 1), h & 1);
        } else {
            if (threadIdx.x >= 256) return;                             // (an ended wave no longer counts at s_barrier)
            split16_body<2, 4, EPI, false, ABLK>(a, tiles_m, tiles_n, 0, none, smem, xcd * q + tail_from + (h >> 1), h & 1);
        }
    }
}

template <int WN, int NI, int EPI, bool PERSIST, bool ABLK, int NST, bool FRONT = false>
int launch16(const GemmArgs& a, const Ws16& ws, hipStream_t s) {
    constexpr int BN = WN * NI * 16;
    const int tiles_m = (a.M + QBM - 1) / QBM, tiles_n = (a.N + BN - 1) / BN;
    const int nwg = PERSIST ? Q_NWG : tiles_m * tiles_n * (a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL((gemm_split16_kernel<WN, NI, EPI, PERSIST, ABLK, NST, FRONT>), dim3(nwg), dim3(2 * WN * 64), 0, s, a, tiles_m, tiles_n, nwg, ws);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int WN, int NI, bool PERSIST, int NST = 2, bool FRONT = false>
int dispatch16(const GemmArgs& a, int epi, const Ws16& ws, hipStream_t s) {
    if (a.a_blk) {      // row-blocked A (fc2's operand): bias + residual, or no epilogue
        if constexpr (FRONT) return -1;
        if (epi == EPI_BIAS_RESID) return launch16<WN, NI, EPI_BIAS_RESID, PERSIST, true, NST>(a, ws, s);
        if (epi == EPI_NONE) return launch16<WN, NI, EPI_NONE, PERSIST, true, NST>(a, ws, s);
        return -1;
    }
    switch (epi) {
        case EPI_NONE: return launch16<WN, NI, EPI_NONE, PERSIST, false, NST, FRONT>(a, ws, s);
        case EPI_BIAS: return launch16<WN, NI, EPI_BIAS, PERSIST, false, NST, FRONT>(a, ws, s);
        case EPI_BIAS_GELU: return launch16<WN, NI, EPI_BIAS_GELU, PERSIST, false, NST, FRONT>(a, ws, s);
        case EPI_BIAS_RESID: return launch16<WN, NI, EPI_BIAS_RESID, PERSIST, false, NST, FRONT>(a, ws, s);
        case EPI_BIAS_QSCALE: return launch16<WN, NI, EPI_BIAS_QSCALE, PERSIST, false, NST, FRONT>(a, ws, s);
        case EPI_BIAS_POS:
            if constexpr (!PERSIST) return (a.N % 4) == 0 && a.resid != nullptr && a.c_split == nullptr ? launch16<WN, NI, EPI_BIAS_POS, false, false, NST, FRONT>(a, ws, s) : -1;
            return -1;
        default: return -1;
    }
}

template <int EPI, bool ABLK>
int launch16_tail(const GemmArgs& a, int tiles_m, int tiles_n, int q, int tail_from, bool tail8, hipStream_t s) {
    const int nwg = 8 * (tail_from + 2 * (q - tail_from));
#ifdef THMR_EXPERIMENTS
    if (tail8) hipLaunchKernelGGL((gemm_split16_tail_kernel<EPI, ABLK, true>), dim3(nwg), dim3(512), 0, s, a, tiles_m, tiles_n, q, tail_from);
    else
#else
    if (tail8) return -1;
#endif
    hipLaunchKernelGGL((gemm_split16_tail_kernel<EPI, ABLK, false>), dim3(nwg), dim3(512), 0, s, a, tiles_m, tiles_n, q, tail_from);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// one workgroup per tile (and K slice): shape 0 = 128 x 256 tile on 8 waves, 1 = 128 x 128 on 4 waves, 2 = 128 x 128 on 8 waves of 64 x 32,
// 3 = 128 x 128 on 4 waves with a THREE-stage K ring (two stages of copies in flight), 4 = the same ring under 8 waves of 64 x 32,
// 5 / 6 = shapes 0 / 3 with every copy of a K tile issued right behind the barrier (FRONT).
// a.ksplit > 1: raw partial sums (epi must be EPI_NONE)
int launch_split16_tiles(const GemmArgs& a, int epi, int shape, hipStream_t s) {
    if (a.c_split != nullptr && epi == EPI_BIAS_RESID) return -1;
    const Ws16 none{nullptr, nullptr};
    switch (shape) {
        case 0: return dispatch16<4, 4, false>(a, epi, none, s);
        case 1: return dispatch16<2, 4, false>(a, epi, none, s);
        case 3: return dispatch16<2, 4, false, 3>(a, epi, none, s);
        case 5: return dispatch16<4, 4, false, 2, true>(a, epi, none, s);      // the 128 x 256 tile with front-loaded copies (few crops)
#ifdef THMR_EXPERIMENTS
        case 6: return dispatch16<2, 4, false, 3, true>(a, epi, none, s);      // the three-stage 128 x 128 tile likewise: slower (profiles/r6o_*)      // the eight-wave forms lost their A/B (3-6 % slower with two stages, 2-3 % with three: profiles/r6d_*, r6g_*)
        case 2: return dispatch16<4, 2, false>(a, epi, none, s);
        case 4: return dispatch16<4, 2, false, 3>(a, epi, none, s);      // eight waves of 64 x 32 AND the three-stage ring
#endif
        default: return -1;
    }
}

// The 128 x 256 tiling with its ragged last round as 128 x 128 half tiles (gemm_split16_tail_kernel).  `cus` = compute units of the device.
// Applies when the tile count divides by the 8 XCDs, an XCD's share q leaves 1 ... cus / 16 tiles after its full rounds and N % 256 == 0;
// returns 1 (nothing launched) when it does not — the caller then launches the plain grid.
int launch_split16_tiles_tail(const GemmArgs& a, int epi, int cus, bool tail8, hipStream_t s) {
    if (a.ksplit > 1 || (a.N % 256) != 0 || cus < 16 || (cus % 8) != 0) return 1;
    if (a.c_split != nullptr && epi == EPI_BIAS_RESID) return -1;
    const int tiles_m = (a.M + QBM - 1) / QBM, tiles_n = a.N / 256, T = tiles_m * tiles_n, per = cus / 8;
    if ((T % 8) != 0) return 1;
    const int q = T / 8, full = q / per, rem = q - full * per;
    if (full < 1 || rem < 1 || 2 * rem > per) return 1;
    const int tf = full * per;
    if (a.a_blk) {
        if (epi == EPI_BIAS_RESID) return launch16_tail<EPI_BIAS_RESID, true>(a, tiles_m, tiles_n, q, tf, tail8, s);
        if (epi == EPI_NONE) return launch16_tail<EPI_NONE, true>(a, tiles_m, tiles_n, q, tf, tail8, s);
        return -1;
    }
    switch (epi) {
        case EPI_NONE: return launch16_tail<EPI_NONE, false>(a, tiles_m, tiles_n, q, tf, tail8, s);
        case EPI_BIAS: return launch16_tail<EPI_BIAS, false>(a, tiles_m, tiles_n, q, tf, tail8, s);
        case EPI_BIAS_GELU: return launch16_tail<EPI_BIAS_GELU, false>(a, tiles_m, tiles_n, q, tf, tail8, s);
        case EPI_BIAS_RESID: return launch16_tail<EPI_BIAS_RESID, false>(a, tiles_m, tiles_n, q, tf, tail8, s);
        case EPI_BIAS_QSCALE: return launch16_tail<EPI_BIAS_QSCALE, false>(a, tiles_m, tiles_n, q, tf, tail8, s);
        default: return 1;
    }
}

// 256 persistent workgroups (gemm_split3_persist_ok shapes); ws = gemm_split3_persist_ws_bytes() of zeroed device memory.
// narrow (round 6): the same tile stream over 128 x 128 tiles on four waves with the three-stage K ring (gemm_split3_persist_narrow_ok
// shapes) — for the few-crop calls whose 128 x 128 grid is MORE than one round (257 ... ~1000 tiles): 1.4 rounds of one-workgroup-per-tile
// cost two K-loop latencies, the stream costs 1.4.
int launch_split16_persist(const GemmArgs& a, int epi, void* ws_mem, bool narrow, hipStream_t s) {
    if (a.c_split != nullptr && epi == EPI_BIAS_RESID) return -1;
    Ws16 ws;
    ws.part = reinterpret_cast<float*>(ws_mem);
    ws.flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws_mem) + (size_t)Q_NWG * Q_SLAB * 4);
    if (narrow) {
        if (a.a_blk) return -1;
        const int tiles_m = (a.M + 127) / 128, tiles_n = a.N / 128;
        switch (epi) {      // what the engine runs this way: qkv, fc1 (split3 output), and raw / bias / residual for the tests
            case EPI_NONE: hipLaunchKernelGGL((gemm_split16_persist_narrow_kernel<EPI_NONE>), dim3(Q_NWG), dim3(256), 0, s, a, tiles_m, tiles_n, Q_NWG, ws); break;
            case EPI_BIAS: hipLaunchKernelGGL((gemm_split16_persist_narrow_kernel<EPI_BIAS>), dim3(Q_NWG), dim3(256), 0, s, a, tiles_m, tiles_n, Q_NWG, ws); break;
            case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_split16_persist_narrow_kernel<EPI_BIAS_GELU>), dim3(Q_NWG), dim3(256), 0, s, a, tiles_m, tiles_n, Q_NWG, ws); break;
            case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm_split16_persist_narrow_kernel<EPI_BIAS_RESID>), dim3(Q_NWG), dim3(256), 0, s, a, tiles_m, tiles_n, Q_NWG, ws); break;
            case EPI_BIAS_QSCALE: hipLaunchKernelGGL((gemm_split16_persist_narrow_kernel<EPI_BIAS_QSCALE>), dim3(Q_NWG), dim3(256), 0, s, a, tiles_m, tiles_n, Q_NWG, ws); break;
            default: return -1;
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    return dispatch16<4, 4, true>(a, epi, ws, s);
}
