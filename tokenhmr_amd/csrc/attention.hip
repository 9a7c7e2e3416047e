// ViT global self-attention for one (crop, head): softmax(q k^T) v over N = 192 tokens, d = 80.
//
// Replaces tokenhmr/lib/models/backbones/vit.py:113-122 (Attention.forward between the qkv and proj
// Linears).  All 32 blocks are global attention over 192 tokens (no windowing in the reference).
// Input  qkv (B,192,3840) = [q(16x80) | k(16x80) | v(16x80)] per token, q already scaled by 80^-0.5 in the
// QKV GEMM epilogue (vit.py:116).  Output (B,192,1280) with column h*80+d (vit.py:122 transpose+reshape).
//
// gfx950 design: one 256-thread workgroup per (b,h), TWO workgroups per CU.  K and V of the head time-share ONE 66 KB LDS
// buffer, split into two 96-key halves that are staged by LDS-DMA (global_load_lds: no VGPR round trip, no ds_write) and
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
//   P = softmax  fp32; exp(x) = v_exp_f32(x * log2 e) (1 ulp) and one reciprocal per row.
//   O^T = V^T P^T  P registers are directly the MFMA B operand (lane group g <-> key 16*kt + 4g + r); V rows come
//                from LDS with conflict-free ds_read_b32 (row stride 84); each lane ends with 4 consecutive d of one
//                query, so the output goes out as 16-byte stores.
// d = 80 = 5 tiles of 16 and 80 = 20 k-steps of 4: the 16x16x4 shape wastes no MFMA work.
// LDS rows: K stride 88 floats = 22 DMA slots of 16 B (20 data + 2 pad), V stride 84 floats = 21 slots (20 + 1): a wave's DMA
// instruction fills 64 consecutive slots, i.e. lane l of instruction q carries slot 64q + l = (row, column) by division.
#include "common.h"

namespace {

constexpr int NTOK = 192, HD = 80, NH = 16, DIM = 1280, QKV_LD = 3840;
constexpr int KS = 88;   // K row stride in LDS (floats): conflict-free for ds_read_b128 lane groups
constexpr int VS = 84;   // V row stride in LDS (floats): conflict-free for ds_read_b32 (rows 4 apart)
constexpr int HALF = NTOK / 2;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Stage 96 rows x 80 floats (row stride QKV_LD in global memory) into LDS rows of SLOTS*4 floats.  Wave w issues the DMA
// instructions q = w, w+4, ...: at most (NI+3)/4 each; K halves take 33 instructions (wave 0 issues 9, the others 8), V halves 32.
template <int SLOTS>
__device__ __forceinline__ void dma_half(const float* __restrict__ src, float* lds, int wave, int lane) {
    constexpr int NSLOT = HALF * SLOTS, NI = (NSLOT + 63) / 64;
#pragma unroll
    for (int i = 0; i < (NI + 3) / 4; ++i) {
        const int q = i * 4 + wave;                  // wave-uniform
        if (q < NI) {
            const int sl = q * 64 + lane, row = sl / SLOTS, c = sl - row * SLOTS;
            if (sl < NSLOT)                          // pad slots re-fetch column 0 (never read)
                __builtin_amdgcn_global_load_lds((gbl_void*)(src + (int64_t)row * QKV_LD + (c < HD / 4 ? c : 0) * 4),
                                                 (lds_void*)(lds + q * 256), 16, 0, 0);
        }
    }
}

// everything but the N youngest VMEM operations of this wave has completed, then the workgroup barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// QT = 16-query tiles per wave: 3 -> one workgroup covers all 192 queries of a (crop, head) (batched path); 1 -> three
// workgroups of 64 queries each, all staging the same K / V (few crops: B*16 workgroups cannot occupy 256 CUs).  Every query
// is computed by the same instruction sequence either way, so the two variants are bit-identical.
template <int QT>
__global__ __launch_bounds__(256, 2) void vit_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out) {
    constexpr int QB = 3 / QT;      // query blocks per (crop, head)
    __shared__ __attribute__((aligned(16))) float smem[NTOK * KS];   // K halves, later overwritten by the V halves
    const int bh = blockIdx.x / QB, qb = blockIdx.x - bh * QB;
    const int b = bh / NH, h = bh % NH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)b * NTOK * QKV_LD + h * HD;

    // ---- Q fragments first (oldest VMEM operations of the wave): B operand of S^T = K Q^T.  B[kslot g][j = query l15];
    //      with the k-permutation, register qf[qt][j][t] = Q[q0 + 16 qt + l15][16 j + 4 g + t] ----
    const int q0 = (qb * 4 + wave) * 16 * QT;
    f32x4 qf[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            qf[qt][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
    // ---- both K halves go out now; the first one is awaited, the second lands under the first half of S ----
    dma_half<KS / 4>(base + DIM, smem, wave, lane);
    dma_half<KS / 4>(base + (int64_t)HALF * QKV_LD + DIM, smem + HALF * KS, wave, lane);

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
    wait_vm_barrier<8>();        // Q and K[0:96] have landed (at most this wave's 8 youngest = K[96:192] copies are in flight)
    s_half(0);
    wait_vm_barrier<0>();        // K[96:192] landed; every wave is done with the first K half ...
    dma_half<VS / 4>(base + 2 * DIM, smem, wave, lane);                                        // ... which V[0:96] overwrites
    s_half(1);
    wait_vm_barrier<8>();        // every wave is done with the second K half (V[0:96] may still be in flight)
    dma_half<VS / 4>(base + (int64_t)HALF * QKV_LD + 2 * DIM, smem + HALF * VS, wave, lane);

    // ---- softmax over the 192 keys of each query (4 lanes x 48 registers per query), under the V copies ----
    constexpr float LOG2E = 1.44269504088896340736f;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float m = s[qt][0][0];
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][kt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f((s[qt][kt][r] - m) * LOG2E);
                s[qt][kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qt][kt][r] *= inv;
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
    wait_vm_barrier<8>();        // V[0:96] landed for every wave
    pv_half(0);
    wait_vm_barrier<0>();        // V[96:192] landed
    pv_half(1);

    // ---- store: D layout of 16x16: col = lane&15 -> query, row = 4*(lane>>4) + reg -> d  (one float4 per tile) ----
    float* obase = out + (int64_t)b * NTOK * DIM + h * HD;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 5; ++dt)
            *reinterpret_cast<f32x4*>(obase + (int64_t)(q0 + qt * 16 + l15) * DIM + dt * 16 + g * 4) = o[qt][dt];
}

}  // namespace

int launch_vit_attention(const float* qkv, float* out, int B, hipStream_t s) {
    if (B <= 0) return -1;
    // while 48*B workgroups of 64 queries still fit the 512 resident slots (2 per CU) they finish sooner than 16*B of 192
    if (B <= 10) hipLaunchKernelGGL(vit_attention_kernel<1>, dim3(B * NH * 3), dim3(256), 0, s, qkv, out);
    else hipLaunchKernelGGL(vit_attention_kernel<3>, dim3(B * NH), dim3(256), 0, s, qkv, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
