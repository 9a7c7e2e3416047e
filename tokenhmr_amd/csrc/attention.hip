// ViT global self-attention for one (crop, head): softmax(q k^T) v over N = 192 tokens, d = 80.
//
// Replaces tokenhmr/lib/models/backbones/vit.py:113-122 (Attention.forward between the qkv and proj
// Linears).  All 32 blocks are global attention over 192 tokens (no windowing in the reference).
// Input  qkv (B,192,3840) = [q(16x80) | k(16x80) | v(16x80)] per token, q already scaled by 80^-0.5 in the
// QKV GEMM epilogue (vit.py:116).  Output (B,192,1280) with column h*80+d (vit.py:122 transpose+reshape).
//
// gfx950 design: one 256-thread workgroup per (b,h), TWO workgroups per CU.  K and V of the head time-share ONE 66 KB LDS
// buffer, split into two 96-key halves that are staged by LDS-DMA (global_load_lds, saddr form: no VGPR round trip, no ds_write, no per-copy VALU) and
// software-pipelined against the matrix work inside the workgroup:
//     DMA K[0:96], K[96:192] | S(keys 0..95) | DMA V[0:96] over the dead K half | S(keys 96..191) | DMA V[96:192] |
//     softmax | P.V(keys 0..95) | P.V(keys 96..191) | store
// so only the first K half's latency is exposed; completion is tracked with s_waitcnt vmcnt(N) (LDS-DMA returns in order).
// (The first version staged K and V through 60 VGPRs per thread with the loads issued phase by phase: a workgroup that was
// alone on its CU spent 23 of its 42 us outside the MFMA phases, and two co-resident workgroups run in lockstep, so their
// load / softmax / store phases coincide instead of overlapping — profiles/r1_attention_experiments.log.)
// Each wave owns 16*QT query rows and keeps their whole score tile in registers (QT = 3: 48x192 = 144 VGPRs) — no KV loop
// and no online softmax is needed at N = 192.
//   S^T = K Q^T  with v_mfma_f32_16x16x4_f32 (A = K rows from LDS via ds_read_b128 + the k-permutation
//                trick, B = Q fragments held in registers).  The swapped product leaves every query's
//                192 scores in 4 lanes x 48 registers, so row max/sum are 47 in-lane ops + 2 xor-shuffles.
//   P = softmax  fp32; e = v_exp_f32(fma(s, log2 e, -m log2 e)), un-normalised into P.V, one reciprocal per row applied to the outputs.
//   O^T = V^T P^T  P registers are directly the MFMA B operand (lane group g <-> key 16*kt + 4g + r); V rows come
//                from LDS with conflict-free ds_read_b32 (row stride 84); each lane ends with 4 consecutive d of one
//                query, so the output goes out as 16-byte stores.
// d = 80 = 5 tiles of 16 and 80 = 20 k-steps of 4: the 16x16x4 shape wastes no MFMA work.
// LDS rows: K stride 88 floats = 22 DMA slots of 16 B (20 data + 2 pad), V stride 84 floats = 21 slots (20 + 1): a wave's DMA
// instruction fills 64 consecutive slots, i.e. lane l of instruction q carries slot 64q + l = (row, column) by division.
#include <cstdlib>

#include "common.h"
#include "attention_device.h"

namespace {

constexpr int KS = 88;   // K row stride in LDS (floats): conflict-free for ds_read_b128 lane groups
constexpr int VS = 84;   // V row stride in LDS (floats): conflict-free for ds_read_b32 (rows 4 apart)
constexpr int HALF = NTOK / 2;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Stage 96 rows x 80 floats (row stride QKV_LD in global memory) into LDS rows of SLOTS*4 floats.  Wave w issues the DMA
// instructions q = w, w+4, ...: at most (NI+3)/4 each; K halves take 33 instructions (wave 0 issues 9, the others 8), V halves 32.
// The per-lane source offset of instruction i ((row, column) of LDS slot 64 q + lane, a division by the slots of a padded row)
// is the same for both halves of K (and of V), so it is computed ONCE into dma_offsets and every copy is then the saddr form of
// global_load_lds_dwordx4: {wave-uniform 64-bit base, SALU} + {that 32-bit offset} — zero VALU instructions per copy.  The first
// version recomputed (row, column) and a 64-bit address for each of the 34 copies: 13 VALU instructions each, a third of the
// kernel's non-MFMA vector work, and on gfx950 VALU time is matrix-pipe time (profiles/r1_mfma_valu_microbench.log).
template <int SLOTS, int NW>
struct DmaOff {
    static constexpr int NSLOT = HALF * SLOTS, NI = (NSLOT + 63) / 64, PER = (NI + NW - 1) / NW;
    static constexpr int MINPER = NI / NW;       // every wave has at least this many copies of a half in flight
    uint32_t off[PER];
};
template <int SLOTS, int NW>
__device__ __forceinline__ DmaOff<SLOTS, NW> dma_offsets(int wave, int lane) {
    DmaOff<SLOTS, NW> d;
#pragma unroll
    for (int i = 0; i < DmaOff<SLOTS, NW>::PER; ++i) {
        const int sl = min((i * NW + wave) * 64 + lane, DmaOff<SLOTS, NW>::NSLOT - 1);   // past-the-end slots re-fetch the last one
        const int row = sl / SLOTS, c = sl - row * SLOTS;
        d.off[i] = ((uint32_t)row * (uint32_t)QKV_LD + (uint32_t)(c < HD / 4 ? c : 0) * 4u) * 4u;   // pad slots re-fetch column 0 (never read)
    }
    return d;
}
// M0 is written inside the asm (it cannot be declared as a clobber); this file uses no LDS-DMA builtin and nothing else that
// reads M0, and the copies are invisible to the compiler's vmcnt bookkeeping: every consumer waits with wait_vm_barrier<N>().
__device__ __forceinline__ void dma16_saddr(const float* base, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_base) : "memory");
}
template <int SLOTS, int NW>
__device__ __forceinline__ void dma_half(const float* __restrict__ src, float* lds, int wave, const DmaOff<SLOTS, NW>& d) {
    const uint32_t l0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
#pragma unroll
    for (int i = 0; i < DmaOff<SLOTS, NW>::PER; ++i) {
        const int q = i * NW + wave;                 // wave-uniform
        if (q < DmaOff<SLOTS, NW>::NI) dma16_saddr(src, d.off[i], l0 + (uint32_t)q * 1024u);
    }
}

// everything but the N youngest VMEM operations of this wave has completed, then the workgroup barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Diagnostics (scripts/micro/attn_timeline.hip only; the product instantiations use DBG = 0 and compile to the same code as before):
//   DBG & 1: wave 0 of every workgroup stamps s_memrealtime (100 MHz) at the phase boundaries and records HW_ID / XCC_ID
//   DBG & 4: the round-1 block order (logical index = blockIdx.x, heads of a crop spread over the eight XCDs)
//   DBG & 2: the workgroup that holds the odd threadgroup slot of its CU (HW_ID.TG_ID & 1) starts `delay_ticks` later when it belongs
//            to the first `first_round` workgroups of the grid (de-phasing experiment)
struct AttnDbg {
    unsigned long long* tl;     // [grid][16]
    int delay_ticks, first_round;
};
#define THMR_ATTN_STAMP(i)                                  \
    if constexpr ((DBG & 1) != 0) {                           \
        __builtin_amdgcn_sched_barrier(0);                  \
        stamp[i] = wall_clock64();                          \
        __builtin_amdgcn_sched_barrier(0);                  \
    }

// QT = 16-query tiles per wave, NW = waves per workgroup.  (3, 4): one 256-thread workgroup covers all 192 queries of a
// (crop, head) (batched path); (1, 4): three workgroups of 64 queries each, all staging the same K / V (few crops: B*16
// workgroups cannot occupy 256 CUs); (1, 12): one 768-thread workgroup, 12 waves of 16 queries, K / V staged once.  Every query
// is computed by the same instruction sequence in all of them, so the variants are bit-identical.
// SPLIT: `out` is a split3 operand [B*192][1280/8][3][8] bf16 (gemm_split.hip) instead of fp32 — the same values, each written as three
// bf16 pieces (a lane's 4 consecutive d are the 8-byte half of the three chunks of one k-group)
template <int QT, int NW, int DBG = 0, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64, NW == 12 ? 6 : 2) void vit_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, AttnDbg dbg) {
    constexpr int QB = 12 / (QT * NW);      // query blocks per (crop, head)
    static_assert(QB * QT * NW == 12, "192 queries = QB workgroups x NW waves x QT tiles of 16");
    typedef DmaOff<KS / 4, NW> KOff;
    typedef DmaOff<VS / 4, NW> VOff;
    __shared__ __attribute__((aligned(16))) float smem[NTOK * KS];   // K halves, later overwritten by the V halves
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement, used for speed only), so the logical index walks each XCD's
    // share of the grid contiguously: the 16 heads of a crop run on ONE XCD at about the same time.  A head's rows are 320-byte
    // pieces at a 15,360-byte stride that straddle 128-byte lines shared with the neighbouring heads; spread over eight L2s every
    // boundary line was fetched twice and HBM saw scattered 320-byte bursts (profiles/r2n_attn_timeline.log: the first MFMA of a
    // workgroup waited 11 us for Q + K[0:96]).  Together, the heads of a crop stream whole 15 KB token rows through one L2.
    int logical;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, within = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
        logical = (DBG & 4) ? (int)blockIdx.x : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int bh = logical / QB, qb = logical - bh * QB;
    const int b = bh / NH, h = bh % NH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)b * NTOK * QKV_LD + h * HD;
    unsigned long long stamp[(DBG & 1) ? 12 : 1];
    unsigned hw_id = 0;
    if constexpr (DBG != 0) hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    THMR_ATTN_STAMP(0)
    if constexpr ((DBG & 2) != 0) {
        if ((int)blockIdx.x < dbg.first_round && ((hw_id >> 16) & 1u) != 0) {       // wave-uniform
            const unsigned long long t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < (long long)dbg.delay_ticks) __builtin_amdgcn_s_sleep(16);
        }
    }
    THMR_ATTN_STAMP(1)

    // ---- Q fragments first (oldest VMEM operations of the wave): B operand of S^T = K Q^T.  B[kslot g][j = query l15];
    //      with the k-permutation, register qf[qt][j][t] = Q[q0 + 16 qt + l15][16 j + 4 g + t] ----
    const int q0 = (qb * NW + wave) * 16 * QT;
    f32x4 qf[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            qf[qt][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
    // ---- both K halves go out now; the first one is awaited, the second lands under the first half of S ----
    {
        const KOff kd = dma_offsets<KS / 4, NW>(wave, lane);
        dma_half<KS / 4, NW>(base + DIM, smem, wave, kd);
        dma_half<KS / 4, NW>(base + (int64_t)HALF * QKV_LD + DIM, smem + HALF * KS, wave, kd);
    }
    const VOff vd = dma_offsets<VS / 4, NW>(wave, lane);     // 8 registers (NW = 4) that live across the S phases

    // ---- S^T tiles: s[qt][kt][r] = S[q0 + 16 qt + l15][16 kt + 4 g + r] ----
    f32x4 s[QT][12];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K fragment read per (kt, j) step feeding 4*QT MFMAs; software-pipelined: the fragment of step i+1 is read BEFORE
    // the MFMAs of step i are issued (pinned with sched_barrier; hipcc's own schedule is read -> s_waitcnt lgkmcnt(0) -> MFMAs)
    auto s_half = [&](int hf) {          // hf is a literal at both call sites
        f32x4 ka = *reinterpret_cast<const f32x4*>(&smem[(hf * HALF + l15) * KS + g * 4]);
#pragma unroll
        for (int i = 0; i < 30; ++i) {
            const int kt = hf * 6 + i / 5, j = i % 5;
            f32x4 kn = ka;
            if (i + 1 < 30)
                kn = *reinterpret_cast<const f32x4*>(&smem[((hf * 6 + (i + 1) / 5) * 16 + l15) * KS + ((i + 1) % 5) * 16 + g * 4]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qf[qt][j][t], s[qt][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ka = kn;
        }
    };
    wait_vm_barrier<KOff::MINPER>();   // Q and K[0:96] have landed (at most this wave's MINPER youngest = K[96:192] copies are in flight)
    THMR_ATTN_STAMP(2)
    s_half(0);
    THMR_ATTN_STAMP(3)
    wait_vm_barrier<0>();        // K[96:192] landed; every wave is done with the first K half ...
    dma_half<VS / 4, NW>(base + 2 * DIM, smem, wave, vd);                                        // ... which V[0:96] overwrites
    THMR_ATTN_STAMP(4)
    s_half(1);
    THMR_ATTN_STAMP(5)
    wait_vm_barrier<VOff::MINPER>();   // every wave is done with the second K half (V[0:96] may still be in flight)
    dma_half<VS / 4, NW>(base + (int64_t)HALF * QKV_LD + 2 * DIM, smem + HALF * VS, wave, vd);
    THMR_ATTN_STAMP(6)

    // ---- softmax over the 192 keys of each query (4 lanes x 48 registers per query), under the V copies.
    //      e_j = 2^(s_j log2e - m log2e) as ONE packed fma per score PAIR (v_pk_fma_f32) + v_exp_f32; the rounding of the
    //      constant m log2e is common to the whole row and cancels in e / sum.  The probabilities are NOT normalised here:
    //      P.V accumulates the un-normalised e and the 80 outputs of a query are scaled by 1/sum afterwards — 20 multiplies
    //      per lane instead of 48, and the row sums are packed adds.  (Round 1: sub, mul, exp, add, mul per score = 1353
    //      non-MFMA VALU instructions per wave, profiles/r1_pmc_attention.json; VALU time is matrix-pipe time on gfx950.) ----
    constexpr float LOG2E = 1.44269504088896340736f;
    float inv[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float m = s[qt][0][0];
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][kt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const f32x2 c2 = splat2(-(m * LOG2E)), l2 = splat2(LOG2E);
        f32x2 sum2 = splat2(0.f);
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 t = __builtin_elementwise_fma(f32x2{s[qt][kt][2 * h], s[qt][kt][2 * h + 1]}, l2, c2);
                const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                s[qt][kt][2 * h] = e.x;
                s[qt][kt][2 * h + 1] = e.y;
                sum2 += e;
            }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        inv[qt] = 1.0f / sum;
    }

    // ---- O^T = V^T P^T: A[i = d l15][kslot g] = V[16 kt + 4 g + r][16 dt + l15], B[kslot g][j = query l15] = P register.
    //      The transposed product leaves each lane with 4 CONSECUTIVE d of one query -> 16-byte output stores. ----
    f32x4 o[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto pv_half = [&](int hf) {         // pipelined like s_half: the five V values of key step i+1 are read before step i's MFMAs
        float vc[5], vn[5];
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) vc[dt] = smem[(hf * HALF + g * 4) * VS + l15 + dt * 16];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int kt = hf * 6 + i / 4, r = i % 4;
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
                vn[dt] = (i + 1 < 24) ? smem[((hf * 6 + (i + 1) / 4) * 16 + g * 4 + (i + 1) % 4) * VS + l15 + dt * 16] : 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[dt], s[qt][kt][r], o[qt][dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) vc[dt] = vn[dt];
        }
    };
    THMR_ATTN_STAMP(7)
    wait_vm_barrier<VOff::MINPER>();   // V[0:96] landed for every wave
    THMR_ATTN_STAMP(8)
    pv_half(0);
    THMR_ATTN_STAMP(9)
    wait_vm_barrier<0>();        // V[96:192] landed
    THMR_ATTN_STAMP(10)
    pv_half(1);
    THMR_ATTN_STAMP(11)

    // ---- normalise + store: D layout of 16x16: col = lane&15 -> query (the lane that holds this query's 1/sum),
    //      row = 4*(lane>>4) + reg -> d  (one float4 per tile) ----
    float* obase = out + (int64_t)b * NTOK * DIM + h * HD;
    if constexpr (SPLIT) {
        store_o_split3<QT>(reinterpret_cast<char*>(out), (int64_t)b * NTOK + q0, h * HD, l15, g, o, inv);
    } else {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
                *reinterpret_cast<f32x4*>(obase + (int64_t)(q0 + qt * 16 + l15) * DIM + dt * 16 + g * 4) = o[qt][dt] * inv[qt];
    }
    if constexpr ((DBG & 1) != 0) {
        if (tid == 0) {
            unsigned long long* t = dbg.tl + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) t[i] = stamp[i];
            t[12] = wall_clock64();
            t[13] = hw_id;
            t[14] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));        // HW_REG_XCC_ID
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Persistent form of vit_attention_kernel<3, 4>: at most 512 workgroups (two per CU) each walk their (crop, head) items, and the
// NEXT item's K is fetched under the CURRENT item's P.V phase:
//     ... | P.V(keys 0..95) | DMA K'[0:96] over the dead V half | P.V(keys 96..191) | DMA K'[96:192] | normalise + store |
//     Q' -> registers | S'(keys 0..95) | ...
// Why: at B = 64 the plain kernel's grid is two rounds of 512 workgroups, and every workgroup waited ~10 us for Q + K[0:96]
// before its first MFMA (profiles/r2n_attn_timeline.log) — 66 MB requested by 512 workgroups at once; 20 of the kernel's
// 29 us above its MFMA floor.  Here only a workgroup's FIRST item pays that.  With at most 512 items (one per workgroup) it is
// still ~3 % faster than the plain kernel — one copy-offset register instead of 17, 240 registers — and serves those grids too
// (launch_vit_attention).  Per query the instruction sequence is the one of vit_attention_kernel, so results are bit-identical
// (scripts/micro/attn_timeline.hip and tests/test_gpu_ops.py check it).
// VMEM operations per wave and item, in issue order (in-order return): V[0:96] x8, V[96:192] x8, K'[0:96] x8, K'[96:192] x8, output
// stores x15, Q' x15.  The copies are invisible to the compiler and waited for by hand (vmcnt); the stores and Q' are ordinary
// instructions whose count the hand-written vmcnt(30) relies on (ISA-level guard: tests/test_host_logic.py).
constexpr int kAttnDephaseUs = 0;     // start-up delay of the odd threadgroup slot's workgroup (see the kernel)
__device__ __forceinline__ void barrier_only() { asm volatile("s_barrier" ::: "memory"); }

template <int DBG = 0, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void vit_attention_persistent_kernel(const float* __restrict__ qkv, float* __restrict__ out, int nitems, AttnDbg dbg) {
    constexpr int QT = 3, NW = 4;
    // LDS image (K halves, later the V halves): rows of PS = 84 floats = 21 slots of 16 B (20 data + 1 pad) — conflict-free both for the
    // ds_read_b128 of K fragments (16 consecutive rows start at 16 distinct multiples of 4 banks) and the ds_read_b32 of V.  One
    // LDS-DMA wave instruction fills 64 consecutive slots = exactly three rows + the first slot of the fourth, so EVERY copy of K and
    // V uses the same per-lane source offset (lane -> row lane / 21, column lane % 21; lane 63 fetches what the next instruction's
    // lane 0 fetches again) and only the wave-uniform base moves: one offset register instead of the 17 of vit_attention_kernel.
    // A half = 32 instructions (8 per wave, wave w issues 8 w ... 8 w + 7); slot 2016 after each half is a dump slot for the last
    // instruction's lane 63, which is clamped to the instruction's first row (it must not read past the tensor).
    constexpr int PS = 84, HSLOTS = HALF * (PS / 4) + 1;      // slots per half incl. the dump slot
    constexpr int CPW = 8;                                    // copies per wave and half
    __shared__ __attribute__((aligned(16))) float smem[2 * HSLOTS * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // items of XCD x: [x * per_xcd, (x + 1) * per_xcd) (nitems = 16 B is a multiple of 8); this workgroup takes every wpx-th of them
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, wpx = gridDim.x >> 3, per_xcd = nitems >> 3;
    int it = within;                                    // index within the XCD's share
    if (it >= per_xcd) return;
    auto item_base = [&](int i) {
        const int bh = xcd * per_xcd + i;
        return qkv + (int64_t)(bh / NH) * NTOK * QKV_LD + (bh % NH) * HD;
    };
    const float* base = item_base(it);
    unsigned long long stamp[(DBG & 1) ? 12 : 1];
    const unsigned hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    THMR_ATTN_STAMP(0)
    // De-phasing: the two workgroups of a CU would otherwise walk their items in lockstep (same phase at the same time: both in
    // softmax / store / waiting for Q' with the matrix pipe idle, then both in S or P.V at half speed each).  The one on the CU's odd
    // threadgroup slot (HW_ID.TG_ID; the co-resident pair always differs in it, scripts/micro/attn_timeline.hip) starts
    // dbg.delay_ticks (10 ns each) late, so one's vector / memory phases fall under the other's MFMA phases; it also halves the
    // burst of Q + K requests at kernel start.  A placement assumption used for speed only: results do not depend on it.
    if (dbg.delay_ticks > 0 && ((hw_id >> 16) & 1u) != 0) {          // wave-uniform
        const unsigned long long t0 = wall_clock64();
        while ((long long)(wall_clock64() - t0) < (long long)dbg.delay_ticks) __builtin_amdgcn_s_sleep(16);
    }
    THMR_ATTN_STAMP(1)

    const int q0 = wave * 16 * QT;
    f32x4 qf[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            qf[qt][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
    uint32_t doff, doff_last;
    {
        const int r = lane / (PS / 4), c = lane - r * (PS / 4);
        doff = ((uint32_t)r * (uint32_t)QKV_LD + (uint32_t)(c < HD / 4 ? c : 0) * 4u) * 4u;
        doff_last = lane == 63 ? 0u : doff;
    }
    // 96 rows starting at src (row stride QKV_LD) -> half hf of the LDS image
    auto dma_rows = [&](const float* src, int hf) {
        const uint32_t l0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)smem + (uint32_t)(hf * HSLOTS + wave * CPW * 63) * 16u;
        const float* s0 = src + (int64_t)(wave * CPW * 3) * QKV_LD;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if (i == CPW - 1 && wave == NW - 1) dma16_saddr(s0 + (int64_t)(i * 3) * QKV_LD, doff_last, l0 + (uint32_t)i * 1008u);
            else dma16_saddr(s0 + (int64_t)(i * 3) * QKV_LD, doff, l0 + (uint32_t)i * 1008u);
        }
    };
    dma_rows(base + DIM, 0);
    dma_rows(base + (int64_t)HALF * QKV_LD + DIM, 1);

    constexpr float LOG2E = 1.44269504088896340736f;
    wait_vm_barrier<CPW>();   // first item: Q and K[0:96] landed
    for (;;) {
        const bool has_next = it + wpx < per_xcd;                    // wave-uniform
        const float* nbase = has_next ? item_base(it + wpx) : base;
        THMR_ATTN_STAMP(2)
        f32x4 s[QT][12];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 12; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto s_half = [&](int hf) {
            f32x4 ka = *reinterpret_cast<const f32x4*>(&smem[hf * HSLOTS * 4 + l15 * PS + g * 4]);
#pragma unroll
            for (int i = 0; i < 30; ++i) {
                const int kt = hf * 6 + i / 5, j = i % 5;
                f32x4 kn = ka;
                if (i + 1 < 30)
                    kn = *reinterpret_cast<const f32x4*>(&smem[hf * HSLOTS * 4 + (((i + 1) / 5) * 16 + l15) * PS + ((i + 1) % 5) * 16 + g * 4]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qf[qt][j][t], s[qt][kt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ka = kn;
            }
        };
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(1);
        s_half(0);
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(0);
        THMR_ATTN_STAMP(3)
        wait_vm_barrier<0>();          // K[96:192] landed (and the previous item's stores); every wave is done with the first K half
        dma_rows(base + 2 * DIM, 0);
        THMR_ATTN_STAMP(4)
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(1);
        s_half(1);
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(0);
        THMR_ATTN_STAMP(5)
        barrier_only();                // every wave is done with the second K half
        dma_rows(base + (int64_t)HALF * QKV_LD + 2 * DIM, 1);
        THMR_ATTN_STAMP(6)

        float inv[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float m = s[qt][0][0];
#pragma unroll
            for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][kt][r]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const f32x2 c2 = splat2(-(m * LOG2E)), l2 = splat2(LOG2E);
            f32x2 sum2 = splat2(0.f);
#pragma unroll
            for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 t = __builtin_elementwise_fma(f32x2{s[qt][kt][2 * h], s[qt][kt][2 * h + 1]}, l2, c2);
                    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    s[qt][kt][2 * h] = e.x;
                    s[qt][kt][2 * h + 1] = e.y;
                    sum2 += e;
                }
            float sum = sum2.x + sum2.y;
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            inv[qt] = 1.0f / sum;
        }

        f32x4 o[QT][5];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto pv_half = [&](int hf) {
            float vc[5], vn[5];
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) vc[dt] = smem[hf * HSLOTS * 4 + (g * 4) * PS + l15 + dt * 16];
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int kt = hf * 6 + i / 4, r = i % 4;
#pragma unroll
                for (int dt = 0; dt < 5; ++dt)
                    vn[dt] = (i + 1 < 24) ? smem[hf * HSLOTS * 4 + (((i + 1) / 4) * 16 + g * 4 + (i + 1) % 4) * PS + l15 + dt * 16] : 0.f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[dt], s[qt][kt][r], o[qt][dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) vc[dt] = vn[dt];
            }
        };
        THMR_ATTN_STAMP(7)
        wait_vm_barrier<CPW>();   // V[0:96] landed for every wave
        THMR_ATTN_STAMP(8)
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(1);
        pv_half(0);
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(0);
        THMR_ATTN_STAMP(9)
        wait_vm_barrier<0>();              // V[96:192] landed; every wave is done with the first V half ...
        if (has_next) dma_rows(nbase + DIM, 0);       // ... which the next item's K[0:96] overwrites
        THMR_ATTN_STAMP(10)
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(1);
        pv_half(1);
        if constexpr ((DBG & 8) != 0) __builtin_amdgcn_s_setprio(0);
        THMR_ATTN_STAMP(11)
        barrier_only();                    // every wave is done with the second V half
        if (has_next) dma_rows(nbase + (int64_t)HALF * QKV_LD + DIM, 1);
        {
            const int bh = xcd * per_xcd + it;
            float* obase = out + (int64_t)(bh / NH) * NTOK * DIM + (bh % NH) * HD;
            if constexpr (SPLIT) {       // 24 stores per item instead of 15 (see the wait below)
                store_o_split3<QT>(reinterpret_cast<char*>(out), (int64_t)(bh / NH) * NTOK + q0, (bh % NH) * HD, l15, g, o, inv);
            } else {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int dt = 0; dt < 5; ++dt)
                        *reinterpret_cast<f32x4*>(obase + (int64_t)(q0 + qt * 16 + l15) * DIM + dt * 16 + g * 4) = o[qt][dt] * inv[qt];
            }
        }
        if constexpr ((DBG & 1) != 0) {
            if (tid == 0 && it == within) {          // timeline of the first item only
                unsigned long long* t = dbg.tl + (size_t)blockIdx.x * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) t[i] = stamp[i];
                t[12] = wall_clock64();
                t[13] = hw_id;
                t[14] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            }
        }
        if (!has_next) break;
        asm volatile("" ::: "memory");      // the stores stay above the loads
        // The next item's Q rows are fetched AFTER the stores, into the registers the output accumulators just left: their latency
        // (~2 us) is exposed, but K'[0:96] — the larger part of what the first MFMA needs — is on chip already.  Fetching Q' earlier
        // (under P.V, or before the stores) was built twice: the fragments then have to live beside the accumulators AND end up
        // where the scores are not, which hipcc can only do with 60 + 144 + 60 registers and 38-44 spills.
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int j = 0; j < 5; ++j)
                qf[qt][j] = *reinterpret_cast<const f32x4*>(nbase + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
        asm volatile("" ::: "memory");
        wait_vm_barrier<SPLIT ? kSplitStores3 + 15 : 30>();   // this wave's K'[0:96] and K'[96:192] copies landed: only the 15 (split3: 24) stores and the 15 Q' loads are younger
        it += wpx;
        base = nbase;
    }
}

// ---- one or two crops: split the KEYS of a (crop, head, 16-query block) over the 4 waves ----
// At one crop the kernels above have 48 workgroups, and what a workgroup takes is one wave's dependent chain: 240 MFMAs for
// S (16 queries x 192 keys x 80) + 240 for P.V = 15,360 matrix-pipe cycles = 6.4 us, after a staging round trip — 12.5 us per
// launch, 0.40 ms of a 4.1 ms call, with 80 % of the CUs idle.  Here a workgroup owns 16 queries of one (crop, head) and wave w
// owns keys [48 w, 48 w + 48): 60 + 60 MFMAs per wave, 12 B workgroups per (crop, head) = 192 per crop.  No LDS staging: a K / V
// element is used by exactly one wave, so the fragments go global -> VGPR in MFMA operand layout (16-byte loads for Q and K with the
// k-permutation trick, dword loads for V^T), all requested before the first MFMA.  Each wave runs its own softmax over its 48 keys
// (local max m_w, local sum l_w, un-normalised O_w); the four partial results are merged through LDS the flash-attention way,
//   O = sum_w O_w 2^((m_w - M) log2 e) / sum_w l_w 2^((m_w - M) log2 e),   M = max_w m_w,
// in wave order (deterministic).  The association of the key sum differs from the kernels above (to fp32 rounding): the engine uses
// it for one and two crops — its own regime (engine.hip kKeysplitMaxB; results agree with the other kernels' to < 1e-5).  From
// three crops on it loses: 12 / QT workgroups per (crop, head) each fetch all of K and V, four times the 64-query kernel's traffic.
// QT = 16-query tiles per workgroup (1, 2, 3): every query is computed by the same instruction sequence whatever QT, so the
// variants are bit-identical and the choice per batch size is free.  QT = 1 has the shortest chain (one crop: 192 workgroups);
// larger QT re-reads K / V less often (12 / QT workgroups per (crop, head) each fetch all of K and V).
template <int QT, bool SPLIT = false>
__global__ __launch_bounds__(256) void vit_attention_keysplit_kernel(const float* __restrict__ qkv, float* __restrict__ out) {
    constexpr int OS = 84;                                   // row stride of the partial-output tile in LDS (floats)
    constexpr int QB = 12 / QT;                              // workgroups per (crop, head)
    constexpr float LOG2E = 1.44269504088896340736f;
    __shared__ __attribute__((aligned(16))) float so[4][16 * QT][OS];
    __shared__ float sm[4][16 * QT], sl[4][16 * QT];
    const int bh = blockIdx.x / QB, qb = blockIdx.x - bh * QB;
    const int b = bh / NH, h = bh % NH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)b * NTOK * QKV_LD + h * HD;
    const int q0 = qb * 16 * QT, k0 = wave * 48;
    // every operand fragment of this wave, requested up front: Q (B operand of S^T = K Q^T), K (A operand), V^T (A operand of O^T = V^T P^T)
    f32x4 qf[QT][5], kf[3][5];
    float vf[3][4][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < 5; ++j) qf[qt][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            kf[kt][j] = *reinterpret_cast<const f32x4*>(base + DIM + (int64_t)(k0 + kt * 16 + l15) * QKV_LD + j * 16 + g * 4);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
                vf[kt][r][dt] = base[2 * DIM + (int64_t)(k0 + kt * 16 + g * 4 + r) * QKV_LD + dt * 16 + l15];
    __builtin_amdgcn_sched_barrier(0);       // all loads are issued before the first MFMA (hipcc otherwise sinks each load to its use: 20 round trips)
    // S^T tiles: s[qt][kt][r] = S[q0 + 16 qt + l15][k0 + 16 kt + 4 g + r]
    f32x4 s[QT][3];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][j][t], qf[qt][j][t], s[qt][kt], 0, 0, 0);
    // this wave's softmax over its 48 keys of each of its queries (4 lanes x 12 registers per query)
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mm = s[qt][0][0];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mm = fmaxf(mm, s[qt][kt][r]);
        mm = fmaxf(mm, __shfl_xor(mm, 16, 64));
        mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
        const float ml = mm * LOG2E;
        float ll = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[qt][kt][r] = __builtin_amdgcn_exp2f(fmaf(s[qt][kt][r], LOG2E, -ml));
                ll += s[qt][kt][r];
            }
        ll += __shfl_xor(ll, 16, 64);
        ll += __shfl_xor(ll, 32, 64);
        m[qt] = mm; l[qt] = ll;
    }
    // O_w^T tiles: o[qt][dt][i] = sum over this wave's keys of e * V, for d = 16 dt + 4 g + i, query 16 qt + l15
    f32x4 o[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][r][dt], s[qt][kt][r], o[qt][dt], 0, 0, 0);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if (g == 0) { sm[wave][qt * 16 + l15] = m[qt]; sl[wave][qt * 16 + l15] = l[qt]; }
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) *reinterpret_cast<f32x4*>(&so[wave][qt * 16 + l15][dt * 16 + g * 4]) = o[qt][dt];
    }
    __syncthreads();
    if constexpr (SPLIT) {
        // the same merge, eight consecutive d per thread, written as the three whole 16-byte chunks of one k-group (out = split3 operand)
        for (int idx = tid; idx < 16 * QT * (HD / 8); idx += 256) {
            const int q = idx / (HD / 8), d = (idx - q * (HD / 8)) * 8;
            const float M = fmaxf(fmaxf(sm[0][q], sm[1][q]), fmaxf(sm[2][q], sm[3][q]));
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            float L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float a = __builtin_amdgcn_exp2f((sm[w][q] - M) * LOG2E);
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&so[w][q][d]), p1 = *reinterpret_cast<const f32x4*>(&so[w][q][d + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0[e] = fmaf(p0[e], a, acc0[e]);
                    acc1[e] = fmaf(p1[e], a, acc1[e]);
                }
                L = fmaf(sl[w][q], a, L);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[e] = acc0[e] / L;
                acc1[e] = acc1[e] / L;
            }
            store_split3_oct(reinterpret_cast<char*>(out) + ((int64_t)b * NTOK + q0 + q) * (DIM * 6), h * HD + d, acc0, acc1);
        }
        return;
    }
    // merge the four partial results in wave order; 16 QT queries x 80 outputs, 5 QT per thread, consecutive threads = consecutive d
#pragma unroll
    for (int jj = 0; jj < 5 * QT; ++jj) {
        const int idx = tid + jj * 256, q = idx / HD, d = idx - q * HD;
        const float M = fmaxf(fmaxf(sm[0][q], sm[1][q]), fmaxf(sm[2][q], sm[3][q]));
        float acc = 0.f, L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __builtin_amdgcn_exp2f((sm[w][q] - M) * LOG2E);
            acc = fmaf(so[w][q][d], a, acc);
            L = fmaf(sl[w][q], a, L);
        }
        out[((int64_t)b * NTOK + q0 + q) * DIM + h * HD + d] = acc / L;
    }
}

}  // namespace

int launch_vit_attention_keysplit_qt(const float* qkv, float* out, int B, int want_qt, hipStream_t s) {
    if (B <= 0) return -1;
    // 16 / 32 / 48 queries per workgroup are bit-identical: one crop takes the shortest chain, more crops the fewer K / V re-reads
    static const int forced_qt = [] { const char* e = thmr_knob("THMR_ATTN_KEYSPLIT_QT"); return e ? atoi(e) : 0; }();   // A/B knob
    const int qt = want_qt ? want_qt : forced_qt ? forced_qt : (B == 1 ? 1 : B <= 2 ? 2 : 3);      // measured: profiles/r3j_attention_keysplit_ab.log
    if (qt == 1) hipLaunchKernelGGL(vit_attention_keysplit_kernel<1>, dim3(B * NH * 12), dim3(256), 0, s, qkv, out);
    else if (qt == 2) hipLaunchKernelGGL(vit_attention_keysplit_kernel<2>, dim3(B * NH * 6), dim3(256), 0, s, qkv, out);
    else hipLaunchKernelGGL(vit_attention_keysplit_kernel<3>, dim3(B * NH * 4), dim3(256), 0, s, qkv, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_vit_attention_keysplit(const float* qkv, float* out, int B, hipStream_t s) { return launch_vit_attention_keysplit_qt(qkv, out, B, 0, s); }

int launch_vit_attention_variant(const float* qkv, float* out, int B, int variant, hipStream_t s);

int launch_vit_attention(const float* qkv, float* out, int B, hipStream_t s) { return launch_vit_attention_variant(qkv, out, B, 0, s); }

// split3 output (the engine's split3 mode): the engine's batch-size rule — key-split kernel for one and two crops, else kernels 1 / 5
int launch_vit_attention_split3(const float* qkv, void* out_split, int B, hipStream_t s) {
    if (B <= 0) return -1;
    float* out = reinterpret_cast<float*>(out_split);
    const AttnDbg nodbg{nullptr, 0, 0};
    if (B == 1) {
        hipLaunchKernelGGL((vit_attention_keysplit_kernel<1, true>), dim3(B * NH * 12), dim3(256), 0, s, qkv, out);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (B == 2) {
        hipLaunchKernelGGL((vit_attention_keysplit_kernel<2, true>), dim3(B * NH * 6), dim3(256), 0, s, qkv, out);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (B <= 10 || (B >= 17 && B <= 24)) {
        hipLaunchKernelGGL((vit_attention_kernel<1, 4, 0, true>), dim3(B * NH * 3), dim3(256), 0, s, qkv, out, nodbg);
    } else {
        const AttnDbg dph{nullptr, kAttnDephaseUs * 100, 0};
        hipLaunchKernelGGL((vit_attention_persistent_kernel<0, true>), dim3(min(B * NH, 512)), dim3(256), 0, s, qkv, out, B * NH, dph);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// variant 0 = the rule below (or THMR_ATTN_VARIANT); 1 / 3 / 5 / 12 / 6 force a kernel (unit tests, A/B)
int launch_vit_attention_variant(const float* qkv, float* out, int B, int want, hipStream_t s) {
    if (B <= 0) return -1;
    // while 48*B workgroups of 64 queries still fit the 512 resident slots (2 per CU) they finish sooner than 16*B of 192
    static const int forced = [] { const char* e = thmr_knob("THMR_ATTN_VARIANT"); return e ? atoi(e) : 0; }();   // A/B knob (scripts/)
    // The variants are bit-identical, so the choice is purely a matter of time (profiles/r2ab_attn_variant_sweep.log, us per launch):
    //   1 = three 64-query workgroups per (crop, head): wins while its 48 B workgroups fill the 512 resident slots evenly — up to 10
    //       crops (one round: 23-25 us vs 30-32) and again for 17-24 crops (two rounds: 44-54 us, where 16 B workgroups of 192
    //       queries put two on some CUs and one on others: 56-58 us);
    //   5 = the persistent kernel everywhere else; with at most 512 items every workgroup has one item and it is still ~3 % faster
    //       than the plain 192-query kernel (3): one copy-offset register instead of 17, no spills; above 512 items it fetches the
    //       next item under the current one.
    //   6 = key-split (vit_attention_keysplit_kernel): another association of the key sum, A/B through the knob only here; the engine
    //       selects it for its whole small-batch regime through launch_vit_attention_keysplit
    const int variant = want ? want : forced ? forced : ((B <= 10 || (B >= 17 && B <= 24)) ? 1 : 5);
    if (variant >= 61 && variant <= 63) return launch_vit_attention_keysplit_qt(qkv, out, B, variant - 60, s);     // key-split with 16 / 32 / 48 queries per workgroup
    if (variant != 1 && variant != 3 && variant != 5 && variant != 6 && variant != 12) return -1;
    const AttnDbg nodbg{nullptr, 0, 0};
    if (variant == 6) return launch_vit_attention_keysplit(qkv, out, B, s);
    if (variant == 5) {
        static const int dephase_us = [] { const char* e = thmr_knob("THMR_ATTN_DEPHASE_US"); return e ? atoi(e) : kAttnDephaseUs; }();   // A/B knob
        const AttnDbg dph{nullptr, dephase_us * 100, 0};
        hipLaunchKernelGGL((vit_attention_persistent_kernel<0>), dim3(min(B * NH, 512)), dim3(256), 0, s, qkv, out, B * NH, dph);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (variant == 1) hipLaunchKernelGGL((vit_attention_kernel<1, 4>), dim3(B * NH * 3), dim3(256), 0, s, qkv, out, nodbg);
#ifdef THMR_EXPERIMENTS
    else if (variant == 12) hipLaunchKernelGGL((vit_attention_kernel<1, 12>), dim3(B * NH), dim3(768), 0, s, qkv, out, nodbg);   // A/B only: measured no faster (profiles/r2c_attention_variants.log)
#else
    else if (variant == 12) return -1;            // 12 waves of 16 queries: experiments build only
#endif
    else hipLaunchKernelGGL((vit_attention_kernel<3, 4>), dim3(B * NH), dim3(256), 0, s, qkv, out, nodbg);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
