// ViT global self-attention for one (crop, head): softmax(q k^T) v over N = 192 tokens, d = 80.
//
// Replaces tokenhmr/lib/models/backbones/vit.py:113-122 (Attention.forward between the qkv and proj
// Linears).  All 32 blocks are global attention over 192 tokens (no windowing in the reference).
// Input  qkv (B,192,3840) = [q(16x80) | k(16x80) | v(16x80)] per token, q already scaled by 80^-0.5 in the
// QKV GEMM epilogue (vit.py:116).  Output (B,192,1280) with column h*80+d (vit.py:122 transpose+reshape).
//
// gfx950 design: one 256-thread workgroup per (b,h), TWO workgroups per CU (2 waves per SIMD) so that one
// workgroup's load / softmax phases run under the other's MFMAs.  K and V of the head time-share ONE 66 KB
// LDS buffer: K is staged first; after S = QK^T the V rows are fetched into the (now dead) Q registers while
// the softmax runs, then written over K.  Each wave owns 48 query rows = 3 tiles of 16 and keeps the whole
// 48x192 score tile in registers (144 VGPRs) — no KV loop and no online softmax is needed at N = 192.
//   S^T = K Q^T  with v_mfma_f32_16x16x4_f32 (A = K rows from LDS via ds_read_b128 + the k-permutation
//                trick, B = Q fragments held in registers).  The swapped product leaves every query's
//                192 scores in 4 lanes x 48 registers, so row max/sum are 47 in-lane ops + 2 xor-shuffles.
//   P = softmax  fp32; exp(x) = v_exp_f32(x * log2 e) (1 ulp) and one reciprocal per row.
//   O^T = V^T P^T  P registers are directly the MFMA B operand (lane group g <-> key 16*kt + 4g + r); V rows come
//                from LDS with conflict-free ds_read_b32 (row stride 84); each lane ends with 4 consecutive d of one
//                query, so the output goes out as 16-byte stores.
// d = 80 = 5 tiles of 16 and 80 = 20 k-steps of 4: the 16x16x4 shape wastes no MFMA work.
#include "common.h"

namespace {

constexpr int NTOK = 192, HD = 80, NH = 16, DIM = 1280, QKV_LD = 3840;
constexpr int KS = 88;   // K row stride in LDS (floats): conflict-free for ds_read_b128 lane groups
constexpr int VS = 84;   // V row stride in LDS (floats): conflict-free for ds_read_b32 (rows 4 apart)
constexpr int F4_PER_THREAD = NTOK * (HD / 4) / 256;   // 15 float4 of a 192x80 tile per thread

// QT = 16-query tiles per wave: 3 -> one workgroup covers all 192 queries of a (crop, head) (batched path); 1 -> three
// workgroups of 64 queries each, all staging the same K / V (few crops: B*16 workgroups cannot occupy 256 CUs).  Every query
// is computed by the same instruction sequence either way, so the two variants are bit-identical.
template <int QT>
__global__ __launch_bounds__(256, 2) void vit_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out) {
    constexpr int QB = 3 / QT;      // query blocks per (crop, head)
    __shared__ __attribute__((aligned(16))) float smem[NTOK * KS];   // K, later overwritten by V
    const int bh = blockIdx.x / QB, qb = blockIdx.x - bh * QB;
    const int b = bh / NH, h = bh % NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)b * NTOK * QKV_LD + h * HD;

    // ---- stage K (coalesced: 20 consecutive threads read one 320 B row) ----
    f32x4 stg[F4_PER_THREAD];
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = tid + i * 256, row = idx / (HD / 4), c4 = idx % (HD / 4);
        stg[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)row * QKV_LD + DIM + c4 * 4);
    }
    // ---- Q fragments: B operand of S^T = K Q^T.  B[kslot g][j = query l15]; with the k-permutation,
    //      register qf[qt][j][t] = Q[q0 + 16 qt + l15][16 j + 4 g + t] ----
    const int q0 = (qb * 4 + wave) * 16 * QT;
    f32x4 qf[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            qf[qt][j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + qt * 16 + l15) * QKV_LD + j * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = tid + i * 256, row = idx / (HD / 4), c4 = idx % (HD / 4);
        *reinterpret_cast<f32x4*>(&smem[row * KS + c4 * 4]) = stg[i];
    }
    __syncthreads();

    // ---- S^T tiles: s[qt][kt][r] = S[q0 + 16 qt + l15][16 kt + 4 g + r] ----
    f32x4 s[QT][12];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // QT = 1 (few crops, mostly one workgroup per CU): software-pipelined — the K fragment of step i+1 is read BEFORE the
    // MFMAs of step i are issued (pinned with sched_barrier; hipcc's own schedule is read -> s_waitcnt lgkmcnt(0) -> MFMAs,
    // which exposes the LDS latency every step): +10 %.  QT = 3: the co-resident workgroup already covers that latency and
    // the extra live registers cost more than they save (profiles/r1_attention_experiments.log), so the plain loop is kept.
    constexpr bool PIPE = QT == 1;
    {
        f32x4 ka = *reinterpret_cast<const f32x4*>(&smem[l15 * KS + g * 4]);
#pragma unroll
        for (int i = 0; i < 60; ++i) {
            const int kt = i / 5, j = i % 5;
            f32x4 kn = ka;
            if constexpr (PIPE) {
                if (i + 1 < 60)
                    kn = *reinterpret_cast<const f32x4*>(&smem[(((i + 1) / 5) * 16 + l15) * KS + ((i + 1) % 5) * 16 + g * 4]);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                ka = *reinterpret_cast<const f32x4*>(&smem[(kt * 16 + l15) * KS + j * 16 + g * 4]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qf[qt][j][t], s[qt][kt], 0, 0, 0);
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                ka = kn;
            }
        }
    }

    // ---- fetch V into registers (in flight during the softmax) ----
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = tid + i * 256, row = idx / (HD / 4), c4 = idx % (HD / 4);
        stg[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)row * QKV_LD + 2 * DIM + c4 * 4);
    }

    // ---- softmax over the 192 keys of each query (4 lanes x 48 registers per query) ----
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

    // ---- V over K in LDS: every wave must be done reading K first ----
    __syncthreads();
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = tid + i * 256, row = idx / (HD / 4), c4 = idx % (HD / 4);
        *reinterpret_cast<f32x4*>(&smem[row * VS + c4 * 4]) = stg[i];
    }
    __syncthreads();

    // ---- O^T = V^T P^T: A[i = d l15][kslot g] = V[16 kt + 4 g + r][16 dt + l15], B[kslot g][j = query l15] = P register.
    //      The transposed product leaves each lane with 4 CONSECUTIVE d of one query -> 16-byte output stores. ----
    f32x4 o[QT][5];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (QT = 1: software-pipelined like the S phase — the five V values of key step i+1 are read before the MFMAs of step i)
    {
        float vc[5], vn[5];
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) vc[dt] = smem[(g * 4) * VS + l15 + dt * 16];
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            const int kt = i / 4, r = i % 4;
            if constexpr (PIPE) {
#pragma unroll
                for (int dt = 0; dt < 5; ++dt)
                    vn[dt] = (i + 1 < 48) ? smem[(((i + 1) / 4) * 16 + g * 4 + (i + 1) % 4) * VS + l15 + dt * 16] : 0.f;
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) vc[dt] = smem[(kt * 16 + g * 4 + r) * VS + l15 + dt * 16];
            }
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[dt], s[qt][kt][r], o[qt][dt], 0, 0, 0);
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) vc[dt] = vn[dt];
            }
        }
    }

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
