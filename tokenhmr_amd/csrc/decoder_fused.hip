// The whole 1-token transformer decoder of the SMPL token head in ONE persistent kernel.
//
// Replaces, for all `depth` (6) layers, tokenhmr/lib/models/components/pose_transformer.py:349-357 (TransformerDecoder.forward
// with the zero token of token_head.py:91), :191-201 (TransformerCrossAttn), :75-86 (1-token self-attention == to_out(v)),
// :111-124 (CrossAttention over the 192 image tokens), :40-52 (FeedForward), :33-37 (PreNorm), plus the four read-out Linears
// (token_head.py:99-105) and the first Linear of the token classifier (token_classifier.py:71-73 mixer_trans.ff.0).
//
// Why one kernel.  With one query token per crop every Linear of the decoder is a GEMV-class product (M = #crops rows), and a
// layer is a chain of 7 such products each of which needs the FULL rows of its predecessor: 42 strictly dependent steps that
// move 96 MB of weights (12 us of HBM time) and ~3 GFLOP.  Round 1 ran them as ~70 separate launches of 5-30 us each
// (0.8 ms per 64 crops, profiles/r1_kernel_stats.csv): the time was launch-to-launch latency and per-kernel load latency,
// not work.  Here 64 ... 256 workgroups of 8 waves stay resident for the whole decoder and meet at a grid barrier between steps:
//   * a step's work items are (16-column tile x 16-row sub-tile), dealt over as many workgroups as there are items (what bounds a
//     step is bytes per CU: ~30 GB/s of misses each); the 8 waves of a workgroup split K 8 ways (v_mfma_f32_16x16x4_f32), so a
//     wave's whole share of the weight stream (<= 8 x 16-byte loads per lane) and of the activations is requested up front and
//     the step costs ~one memory round trip; the 8 partial tiles are summed through LDS in a FIXED order (deterministic, and a
//     crop's result does not depend on the batch it rides in);
//   * LayerNorm is the prologue of the step that consumes it and reads its input ONCE: two-pass row statistics from the A
//     fragments already in registers (per-wave partial sums exchanged through LDS, fixed wave order);
//   * cross-attention is one wave per (crop, head): coalesced 256-byte K / V rows, 3 scores per lane, 16-lane reductions;
//   * the grid barrier is flag based with no read-modify-write and no cache maintenance; the activations that cross it use
//     device-scope atomics.  It needs all workgroups resident: at most one 512-thread workgroup per CU, several such kernels of
//     different engines are chained by the host (engine.hip launch_decoder_serialised), and the poll is BOUNDED: on timeout the
//     kernel sets an error word (thmr_engine_status) and exits instead of hanging the GPU.
#include <cstdlib>

#include "mixer_device.h"

namespace {

constexpr int E = 1024, INNER = 512, DMLP = 1024, TOKK = 192, NHEAD = 8;
constexpr int NBLK = 64, NWAVE = 8, NS = 8;    // persistent grid: 64..256 workgroups x 8 waves; NS = most 16-k steps per wave (K = 1024)
constexpr float LN_EPS = 1e-5f;

// ---- grid barrier ------------------------------------------------------------------------------------------------------
// sync[0] release word (epoch published by workgroup 0), sync[1] generation = epoch when the previous kernel of this engine
// finished, sync[2] exit counter, sync[3] sticky error (1 = barrier timeout), sync[256 + b] arrival flag of workgroup b.
//
// No read-modify-write and no cache maintenance: every workgroup stores the epoch into its own flag word, the (up to 256)
// threads of workgroup 0 poll one flag each, then workgroup 0 publishes the epoch and one thread per workgroup polls that word.
// The data that crosses the barrier (the few-KB activation rows) moves with DEVICE-SCOPE relaxed atomic stores / loads
// (sc1: coherent across the 8 XCDs' L2s by themselves), so the barrier only has to ORDER: each wave drains its stores
// (s_waitcnt) before the workgroup barrier that precedes the flag store.  Measured on MI355X (scripts/micro/grid_barrier.hip,
// profiles/r2_grid_barrier_microbench.log): 2.3 us flat from 8 to 128 workgroups, against 3.1 us (64) / 4.7 us (128) for a
// shared counter with a release fence (L2 write-back) before and an acquire fence (L2 invalidate) after, and against 45 us for
// the very first version (every thread __threadfence() on both sides + an ACQUIRE poll loop).
struct GridSync {
    unsigned* w;
    unsigned base;      // sync[1] at kernel start
    unsigned n;         // barriers passed so far
    unsigned* host_err; // host-mapped copy of the sticky error word (may be null): the host sees a timeout without a D2H copy
    int a2a;            // 1: all-to-all form — every workgroup polls every arrival flag itself (one memory round trip less)
};

// `pf` runs after this wave's stores have drained and before it waits: the place to request what the NEXT stage needs and does not
// depend on the other workgroups (its weight fragments) — the loads then fly while the barrier is being crossed.
template <class PF>
__device__ __forceinline__ bool grid_barrier(GridSync& gs, int tid, volatile int* s_ok, PF&& pf) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's device-scope stores have completed
    pf();
    __syncthreads();
    gs.n += 1;
    const unsigned epoch = gs.base + gs.n;
    const int G = gridDim.x;
    constexpr unsigned LIMIT = 1u << 22;                            // ~0.5 s of polling: never hang the GPU
    if (tid == 0) {
        __hip_atomic_store(&gs.w[256 + blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = 1;
    }
#ifdef THMR_EXPERIMENTS
    if (gs.a2a) {
        // Round 3 experiment (THMR_DEC_BARRIER=1), NOT the default: all-to-all.  The two-hop form below costs flag store -> workgroup 0's
        // poll -> release store -> everybody's poll = two device-scope round trips (~0.85 us each, scripts/micro/xcd_handoff.hip);
        // here every workgroup polls the G arrival flags itself (thread t polls flag t): one round trip on paper, but G workgroups x G
        // polling threads contend on the flag lines — barriers 3-5 us instead of 2-3.2, head 0.711 vs 0.664 ms at one crop.
        __syncthreads();                                            // s_ok initialised
        if (tid < G) {
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(&gs.w[256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > LIMIT) { *s_ok = 0; break; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (*s_ok == 0) {
                __hip_atomic_store(&gs.w[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (gs.host_err) __hip_atomic_store(gs.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (__hip_atomic_load(&gs.w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) *s_ok = 0;   // somebody else gave up
        }
        __syncthreads();
        return *s_ok != 0;
    }
#endif
    if (blockIdx.x == 0) {
        __syncthreads();                                            // s_ok initialised
        if (tid < G) {
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(&gs.w[256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > LIMIT) { *s_ok = 0; break; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (*s_ok == 0) {
                __hip_atomic_store(&gs.w[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (gs.host_err) __hip_atomic_store(gs.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(&gs.w[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // also on failure: let the others leave
        }
    } else if (tid == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(&gs.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2 * LIMIT) { *s_ok = 0; break; }
        }
        if (__hip_atomic_load(&gs.w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) *s_ok = 0;
    }
    __syncthreads();
    return *s_ok != 0;
}

__device__ __forceinline__ bool grid_barrier(GridSync& gs, int tid, volatile int* s_ok) {
    return grid_barrier(gs, tid, s_ok, [] {});
}

// ---- one GEMV-class step: C[m][n] = epi(sum_k A'[m][k] W[n][k]), work item = (16-column tile, 16-row sub-tile) ------------
// A' = A or LayerNorm(A) (PreNorm, pose_transformer.py:33-37).  EPI: 0 none, 1 +bias, 2 gelu(+bias), 4 resid + (+bias).
//
// What bounds a step is bytes per CU: a CU sustains ~30 GB/s of misses (8 TB/s / 256 CUs), so the (column tile x row sub-tile)
// items are dealt over as many workgroups as there are items — 64 for <= 16 crops, 256 (the whole chip) from 49 crops on — and
// an item moves only 64 KB of weights + 64 KB of activations.  (The first version gave each of 64 workgroups all 64 rows:
// 320 KB per CU and 8-18 us per step at 64 crops, profiles/r2d_decoder_timeline.log.)
// Registers: 8 waves per workgroup = 256 per wave: W fragments, A fragments of the wave's K slice and, for LN steps, gamma /
// beta of the slice are all requested before anything is consumed, so an item costs one memory round trip.  LayerNorm reads A
// ONCE: the row statistics are two-pass (mean, then sum (x - mean)^2, like nn.LayerNorm) but computed from the fragments
// already in registers — per-wave partial sums of the wave's K slice, exchanged through LDS and added in a fixed wave order.
struct RowStat {                   // LDS exchange area of the in-register LayerNorm: partial sums [wave][row of the sub-tile]
    float part[NWAVE][16];
};

// normalise the fragments in place: x <- (x - mean) * rstd * gamma + beta  (K = E = 1024: NS steps per wave)
__device__ __forceinline__ void layernorm_frags(f32x4 (&xq)[NS], const f32x4 (&gm)[NS], const f32x4 (&bt)[NS], RowStat* rs,
                                                int wave, int l15, int g) {
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float ps = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (pass == 0) ps += (xq[s][0] + xq[s][1]) + (xq[s][2] + xq[s][3]);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float d = xq[s][t] - mean;
                    ps += d * d;
                }
            }
        }
        ps += __shfl_xor(ps, 16, 64);         // the 4 lane groups hold the 4 x 4-float pieces of each 16-k step
        ps += __shfl_xor(ps, 32, 64);
        if (g == 0) rs->part[wave][l15] = ps;
        __syncthreads();
        float tot = rs->part[0][l15];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) tot += rs->part[w][l15];
        if (pass == 0) mean = tot * (1.0f / E);
        else rstd = 1.0f / sqrtf(tot * (1.0f / E) + LN_EPS);
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) xq[s][t] = (xq[s][t] - mean) * rstd * gm[s][t] + bt[s][t];
}

// The W fragments of a workgroup's FIRST item of the next GEMV stage, requested while the grid barrier in front of that stage is
// crossed (the weights do not depend on it; a step at few crops is one HBM round trip for 64 KB of weights + an L2 round trip for the
// activations, and the barrier takes about as long as the former).  Same addresses as gemv_stage uses; item = -1: nothing requested.
struct WPre {
    f32x4 w[NS];
    int item;
};
__device__ __forceinline__ void prefetch_w(WPre& pre, const float* __restrict__ W, int K, int N, int B, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const int ntile = N >> 4, nsub = (B + 15) >> 4;
    const int kper = K / NWAVE, kbeg = wave * kper, nstep = kper >> 4;
    const int item = blockIdx.x;
    pre.item = item < ntile * nsub ? item : -1;
    if (pre.item < 0) return;
    const float* wp = W + (int64_t)((item % ntile) * 16 + l15) * K + kbeg + g * 4;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nstep) pre.w[s] = *reinterpret_cast<const f32x4*>(wp + s * 16);
}

template <bool LN, int EPI>
__device__ __forceinline__ void gemv_stage(const float* __restrict__ A, int lda, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, const float* __restrict__ W, int K,
                                           const float* __restrict__ bias, const float* resid, float* C, int ldc, int N, int B,
                                           float (*red)[64][4], RowStat* rs, int tid, const WPre& pre) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;   // wave: SGPR
    const int ntile = N >> 4, nsub = (B + 15) >> 4;
    const int kper = K / NWAVE, kbeg = wave * kper, nstep = kper >> 4;    // K = 1024 -> 8 steps of 16 k, K = 512 -> 4
    f32x4 gm[LN ? NS : 1], bt[LN ? NS : 1];
    if constexpr (LN) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            gm[s] = *reinterpret_cast<const f32x4*>(gamma + kbeg + s * 16 + g * 4);
            bt[s] = *reinterpret_cast<const f32x4*>(beta + kbeg + s * 16 + g * 4);
        }
    }
    for (int item = blockIdx.x; item < ntile * nsub; item += gridDim.x) {
        const int ct = item % ntile, mt = item / ntile;              // neighbouring workgroups share the rows, not the weights
        const int n0 = ct * 16, m0 = mt * 16;
        const float* wp = W + (int64_t)(n0 + l15) * K + kbeg + g * 4;
        const float* ap = A + (int64_t)min(m0 + l15, B - 1) * lda + kbeg + g * 4;
        f32x4 wq[NS], xq[NS];
        const bool have_w = item == pre.item;             // requested under the grid barrier in front of this stage (workgroup-uniform)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < nstep) {
                if (have_w) wq[s] = pre.w[s];
                else wq[s] = *reinterpret_cast<const f32x4*>(wp + s * 16);
                xq[s] = ld_dev4(ap + s * 16);
            }
        if constexpr (LN) layernorm_frags(xq, gm, bt, rs, wave, l15, g);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < nstep)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s][t], wq[s][t], acc, 0, 0, 0);
        *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
        __syncthreads();
        if (wave == 0) {                      // the K-slice partial tiles (one per wave) in a fixed order, epilogue, store
            f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][lane][0]);
            const int n = n0 + l15;
            float bv = 0.f;
            if constexpr (EPI != 0) bv = bias[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + g * 4 + r;                // D layout 16x16: row = 4*(lane>>4) + reg, col = lane&15
                if (m < B) {
                    float o = v[r];
                    if constexpr (EPI != 0) o = o + bv;
                    if constexpr (EPI == 2) o = gelu_erf(o);
                    if constexpr (EPI == 4) o = ld_dev(resid + (int64_t)m * ldc + n) + o;
                    st_dev(C + (int64_t)m * ldc + n, o);
                }
            }
        }
        __syncthreads();                      // red is reused by the next item
    }
}

// Many column tiles over the SAME rows (the consumers of the decoder output: 2 read-out tiles + 640 mixer_trans tiles, K = 1024):
// a workgroup keeps the A fragments of ONE 16-row sub-tile in registers while it walks its share of the column tiles, and the W
// fragments of the next tile are requested before the current one is reduced, so the 42 MB weight stream never waits for an LDS
// reduction.  vt < n_ro: read-out tile (W0, 31 valid rows + a zero row, ld 32), else mixer_trans tile vt - n_ro (W1).
__device__ __forceinline__ void gemv_multi(const float* __restrict__ A, const float* __restrict__ W0, const float* __restrict__ b0, float* C0,
                                           const float* __restrict__ W1, const float* __restrict__ b1, float* C1, int n_ro, int n_all, int B,
                                           float (*red)[64][4], int tid, bool dev_c1) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const int kbeg = wave * (E / NWAVE);
    const int nsub = (B + 15) >> 4, G = gridDim.x;
    auto wptr = [&](int vt) -> const float* {
        return vt < n_ro ? W0 + (int64_t)min(vt * 16 + l15, 31) * E + kbeg + g * 4           // row 31 = the zero padding row
                         : W1 + (int64_t)((vt - n_ro) * 16 + l15) * E + kbeg + g * 4;
    };
    // sub-tile mt is served by the workgroups b == mt (mod nsub_g); with more sub-tiles than workgroups-per-tile allow, loop
    const int lanes_of_b = min(nsub, G);                    // distinct sub-tiles in flight
    for (int mt = blockIdx.x % lanes_of_b; mt < nsub; mt += lanes_of_b) {
        const int first = blockIdx.x / lanes_of_b, step = G / lanes_of_b;     // this workgroup's column tiles: first, first + step, ...
        if (first >= step) continue;                        // G not divisible: the remainder workgroups idle in this stage
        const int m0 = mt * 16;
        const float* ap = A + (int64_t)min(m0 + l15, B - 1) * E + kbeg + g * 4;
        f32x4 xq[NS], wq[NS], wn[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) xq[s] = ld_dev4(ap + s * 16);
        int vt = first;
        if (vt < n_all) {
            const float* wp = wptr(vt);
#pragma unroll
            for (int s = 0; s < NS; ++s) wq[s] = *reinterpret_cast<const f32x4*>(wp + s * 16);
        }
#pragma unroll 1
        for (; vt < n_all; vt += step) {
            const int nxt = vt + step;
            if (nxt < n_all) {
                const float* wp = wptr(nxt);
#pragma unroll
                for (int s = 0; s < NS; ++s) wn[s] = *reinterpret_cast<const f32x4*>(wp + s * 16);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s][t], wq[s][t], acc, 0, 0, 0);
            *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
            __syncthreads();
            if (wave == 0) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
                for (int w = 1; w < NWAVE; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][lane][0]);
                const bool ro = vt < n_ro;
                const int n = (ro ? vt : vt - n_ro) * 16 + l15, N = ro ? 31 : 10240, ldc = ro ? 32 : 10240;
                if (n < N) {
                    const float bv = (ro ? b0 : b1)[n];
                    float* C = ro ? C0 : C1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + g * 4 + r;
                        if (m < B) {
                            if (!ro && dev_c1) st_dev(&C[(int64_t)m * ldc + n], v[r] + bv);      // consumed by other workgroups of THIS kernel
                            else C[(int64_t)m * ldc + n] = v[r] + bv;
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) wq[s] = wn[s];
        }
    }
}

}  // namespace

namespace {

// CrossAttention.forward for ONE query token (pose_transformer.py:111-124), one wave per (crop, head):
// dots[j] = (q_h . k_h[j]) * 64^-0.5 (scaled AFTER the dot, :117), softmax over the 192 keys, out_h = sum_j a_j v_h[j].
// A wave instruction reads 4 keys x 256 B (16 lanes x 16 B per key row); the 4-float partial dots are summed over the 16
// lanes of a key.  Key quad i = 0..47 of lane group g is key 4i + g; its score is kept by ONE lane of the group (lane i % 16,
// slot i / 16), so a lane holds 3 scores, the softmax is 3 exponentials per lane + two wave reductions, and the un-normalised
// weights go through a wave-private 768-byte LDS row to the lanes that multiply them with the V rows they load.
__device__ __forceinline__ void cross_attn_stage(const DecParams& p, int layer, int tid, float* pl /* wave-private, >= 192 floats */) {
    const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    // item -> (wave slot, workgroup): consecutive items go to different workgroups, so 512 items occupy all of them
    const int gwave = __builtin_amdgcn_readfirstlane(tid >> 6) * gridDim.x + blockIdx.x;
    for (int item = gwave; item < p.B * NHEAD; item += gridDim.x * NWAVE) {
        const int b = item >> 3, h = item & 7;
        const f32x4 qv = ld_dev4(p.dq + (int64_t)b * INNER + h * 64 + l15 * 4);
        const float* kb = p.kv + (int64_t)b * TOKK * p.ldkv + layer * 2 * INNER + h * 64 + l15 * 4;
        float s3[3] = {0.f, 0.f, 0.f};
        // 4 chunks of 12 wave loads, double-buffered: chunk c + 1 is in flight while chunk c is reduced (rolled by two so that the
        // buffers are static registers; fully unrolled, hipcc materialises all 96 row addresses = 192 registers up front)
        const int64_t cstride = (int64_t)48 * p.ldkv;            // 12 wave loads x 4 keys per chunk
        const float* kp = kb + (int64_t)g * p.ldkv;
        auto load8 = [&](const float* src, f32x4 (&dst)[12]) {
#pragma unroll
            for (int i = 0; i < 12; ++i) dst[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * 4) * p.ldkv);
        };
        auto score8 = [&](const f32x4 (&kk)[12], int c) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                float d = kk[i][0] * qv[0];
                d = fmaf(kk[i][1], qv[1], d);
                d = fmaf(kk[i][2], qv[2], d);
                d = fmaf(kk[i][3], qv[3], d);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                d += __shfl_xor(d, 8, 64);
                // key quad idx = 12 c + i is kept by lane idx % 16 in slot idx / 16
                const int idx = c * 12 + i;
                const bool mine = l15 == (idx & 15);
                const float sv = d * 0.125f;
                if (mine && (idx >> 4) == 0) s3[0] = sv;
                if (mine && (idx >> 4) == 1) s3[1] = sv;
                if (mine && (idx >> 4) == 2) s3[2] = sv;
            }
        };
        {
            f32x4 ka[12], kc[12];
            load8(kp, ka);
#pragma unroll 1
            for (int c = 0; c < 4; c += 2) {
                load8(kp + (int64_t)(c + 1) * cstride, kc);
                score8(ka, c);
                if (c + 2 < 4) load8(kp + (int64_t)(c + 2) * cstride, ka);
                score8(kc, c + 1);
            }
        }
        const float m = wave_max(fmaxf(fmaxf(s3[0], s3[1]), s3[2]));
        float e3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            e3[j] = expf(s3[j] - m);
            pl[g * 48 + j * 16 + l15] = e3[j];           // weight of key quad i = 16 j + l15 of group g
        }
        const float sum = wave_sum((e3[0] + e3[1]) + e3[2]);
        const float* vp = kb + INNER + (int64_t)g * p.ldkv;
        const float* wl = pl + g * 48;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto pv8 = [&](const f32x4 (&vv)[12], int c) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const float w = wl[c * 12 + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(w, vv[i][e], acc[e]);
            }
        };
        {
            f32x4 va[12], vc[12];
            load8(vp, va);
#pragma unroll 1
            for (int c = 0; c < 4; c += 2) {
                load8(vp + (int64_t)(c + 1) * cstride, vc);
                pv8(va, c);
                if (c + 2 < 4) load8(vp + (int64_t)(c + 2) * cstride, va);
                pv8(vc, c + 1);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += __shfl_xor(acc[e], 16, 64);
            acc[e] += __shfl_xor(acc[e], 32, 64);
        }
        if (g == 0) {
            const float inv = 1.0f / sum;
            float* o = p.dca + (int64_t)b * INNER + h * 64 + l15 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) st_dev(o + e, acc[e] * inv);
        }
    }
}

// ---- the MLP-Mixer stack, distributed: ten workgroups per crop, one 16-token tile each ---------------------------------------
// token_classifier.py:92-101.  mixer_stack_kernel (one workgroup per crop) leaves all but B compute units idle and takes 0.30 ms
// whatever the batch: at one crop a quarter of the head, 7 % of the whole call.  Here a crop is shared by 10 / 5 / 2 workgroups
// (p.mixer_cluster; up to 25 / 51 / 128 crops on 256 CUs) that own 1 / 2 / 5 of its ten 16-token tiles from mixer_trans'
// LayerNorm to mixer_norm_layer:
//   * per-token work (LayerNorm1 / 2, channel mixing, mixer_norm_layer) stays inside the owner;
//   * token mixing needs every token of the crop: the owners publish their LayerNorm1 rows (device-scope stores), ONE grid barrier
//     per layer, every workgroup gathers the crop's 160 rows into its LDS and recomputes the small hidden activation u (64 x 64,
//     640 MFMAs) redundantly instead of exchanging it, then finishes z only for its own tokens;
//   * the crop statistics of mixer_trans' LayerNorm(10240) are recomputed by every owner with the thread mapping of
//     mixer_stack_kernel.
// Every value is produced by the functions of mixer_device.h on the same operands in the same order as in mixer_stack_kernel:
// results are bit-identical to it (tests/test_gpu_model.py), so the choice of form never shows in a crop's result.
__device__ __forceinline__ bool mixer_cluster_stage(const DecParams& p, GridSync& gs, int tid, volatile int* s_ok, float* Y, float* U, float* Xl,
                                                    float* redbuf) {
    using namespace mixer;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x;
    const int per_crop = p.mixer_cluster, tpw = 10 / per_crop;   // workgroups per crop (10, 5 or 2), token tiles per workgroup (1, 2 or 5)
    const bool active = wg < per_crop * p.B;                 // workgroup-uniform
    const int b = wg / per_crop, t0 = (wg - b * per_crop) * tpw * 16, nrow = tpw * 16;
    if (active) {
        f32x4 o[5];
        trans_ln<true>(p.mt + (int64_t)b * (T * H), p.mx.tln_w, p.mx.tln_b, redbuf, tid, lane, wave, o);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e0 = (i * NT + tid) * 4, t = e0 >> 6, h = e0 & 63;
            if (t >= t0 && t < t0 + nrow) *reinterpret_cast<f32x4*>(&Xl[(t - t0) * LD + h]) = o[i];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const MixerLayerW& w = p.mx.L[l];
        float* yg = p.mixy[l & 1] + (int64_t)b * (T * H);
        if (active) {                                        // y = LayerNorm1(x) for the own rows -> exchange buffer
            const float gamma = w.ln1w[lane], beta = w.ln1b[lane];
            for (int r = wave; r < nrow; r += NW) st_dev(yg + (t0 + r) * H + lane, ln_row_value(Xl[r * LD + lane], gamma, beta));
        }
        if (!grid_barrier(gs, tid, s_ok)) return false;
        if (active) {
            for (int idx = tid; idx < T * H / 4; idx += NT) {    // the crop's 160 LayerNorm-ed rows -> LDS
                const int t = idx >> 4, h = (idx & 15) * 4;
                *reinterpret_cast<f32x4*>(&Y[t * LD + h]) = ld_dev4(yg + idx * 4);
            }
            __syncthreads();
#pragma unroll 1
            for (int tile = wave; tile < 16; tile += NW) token_mix1_tile(Y, U, w, tile, l15, g);
            __syncthreads();
#pragma unroll 1
            for (int tile = wave; tile < 4 * tpw; tile += NW) {     // s = x + z for the own tokens: (h-tile, own token tile) pairs
                const int tt = tile >> 2;
                token_mix2_tile(U, Xl + tt * 16 * LD, Y + (t0 + tt * 16) * LD, w, (tile & 3) * 16, t0 + tt * 16, l15, g);
            }
            __syncthreads();
#pragma unroll 1
            for (int tt = 0; tt < tpw; ++tt)                        // u is dead: its LDS holds zh
                channel_mix_tile_8waves(Y + (t0 + tt * 16) * LD, Xl + tt * 16 * LD, U, w, wave, lane, l15, g);
        }
    }
    if (active)
        for (int tt = wave; tt < tpw; tt += NW) norm_layer_tile(Xl + tt * 16 * LD, p.mx.out + ((int64_t)b * T + t0 + tt * 16) * H, p.mx, l15, g);
    return true;
}

__global__ __launch_bounds__(NWAVE * 64) void decoder_persistent_kernel(DecParams p) {
    __shared__ __attribute__((aligned(16))) float mixY[mixer::T * mixer::LD];      // distributed mixer tail: the crop's LayerNorm-ed rows / s
    __shared__ __attribute__((aligned(16))) float mixU[mixer::H * mixer::LD];      // token-mixing hidden activation
    __shared__ __attribute__((aligned(16))) float mixX[80 * mixer::LD];            // residual rows of the own token tiles (up to 5)
    __shared__ float mixred[mixer::NW];
    static_assert(mixer::NW == NWAVE, "the distributed mixer tail uses the decoder kernel's 8 waves");
    __shared__ __attribute__((aligned(16))) float red[NWAVE][64][4];         // 8 KB: K-slice partial tiles (also the attention weights)
    __shared__ RowStat rowstat;
    __shared__ unsigned s_base;
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = __hip_atomic_load(&p.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    GridSync gs{p.sync, s_base, 0, p.host_err, p.barrier_a2a};
    bool ok = true;
#ifdef THMR_EXPERIMENTS
    if (p.debug_fail && blockIdx.x == 0 && tid == 0) {      // tests only: exercise the host's recovery path without a real timeout
        __hip_atomic_store(&gs.w[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gs.host_err) __hip_atomic_store(gs.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#endif
    const int B = p.B;
    // optional stage timeline (THMR_DEC_TIMELINE=1): workgroup 0 stamps the 100 MHz wall clock after every step and barrier
    unsigned long long* stamp = (p.timeline && blockIdx.x == 0 && tid == 0) ? reinterpret_cast<unsigned long long*>(p.sync + 16) : nullptr;
    int nstamp = 0;
#define THMR_STAMP() do { if (stamp) stamp[nstamp++] = wall_clock64(); } while (0)
    THMR_STAMP();

    // layer-0 input (pose_transformer.py:350-354 with token == 0): Linear(1 -> 1024)(0) == bias exactly, += pos_embedding
    for (int i = blockIdx.x * (NWAVE * 64) + tid; i < B * E; i += gridDim.x * NWAVE * 64) {
        const int c = i & (E - 1);
        st_dev(p.dx + i, p.tok_bias[c] + p.pos[c]);
    }
    THMR_STAMP();
    WPre pre;
    pre.item = -1;
    ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, p.L[0].wv, E, INNER, B, tid); });
    THMR_STAMP();
    for (int l = 0; ok && l < p.depth; ++l) {
        const DecLayerW& w = p.L[l];
        // self-attention over ONE token: softmax of a single score == 1, so out = to_out(v)   (pose_transformer.py:75-86)
        gemv_stage<true, 0>(p.dx, E, w.n0w, w.n0b, w.wv, E, nullptr, nullptr, p.dv, INNER, INNER, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, w.wo1, INNER, E, B, tid); }))) break;
        THMR_STAMP();
        gemv_stage<false, 4>(p.dv, INNER, nullptr, nullptr, w.wo1, INNER, w.bo1, p.dx, p.dx, E, E, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, w.wq, E, INNER, B, tid); }))) break;
        THMR_STAMP();
        // cross-attention (pose_transformer.py:111-124); the context is NOT normalised (PreNorm only touches x)
        gemv_stage<true, 0>(p.dx, E, w.n1w, w.n1b, w.wq, E, nullptr, nullptr, p.dq, INNER, INNER, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        pre.item = -1;
        if (!(ok = grid_barrier(gs, tid, &s_ok))) break;
        THMR_STAMP();
        cross_attn_stage(p, l, tid, &red[__builtin_amdgcn_readfirstlane(tid >> 6)][0][0]);
        THMR_STAMP();
        if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, w.wo2, INNER, E, B, tid); }))) break;
        THMR_STAMP();
        gemv_stage<false, 4>(p.dca, INNER, nullptr, nullptr, w.wo2, INNER, w.bo2, p.dx, p.dx, E, E, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, w.w1, E, DMLP, B, tid); }))) break;
        THMR_STAMP();
        // feed-forward (pose_transformer.py:40-52)
        gemv_stage<true, 2>(p.dx, E, w.n2w, w.n2b, w.w1, E, w.b1, nullptr, p.dff, DMLP, DMLP, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, w.w2, DMLP, E, B, tid); }))) break;
        THMR_STAMP();
        gemv_stage<false, 4>(p.dff, DMLP, nullptr, nullptr, w.w2, DMLP, w.b2, p.dx, p.dx, E, E, B, red, &rowstat, tid, pre);
        THMR_STAMP();
        if (l + 1 < p.depth) {
            if (!(ok = grid_barrier(gs, tid, &s_ok, [&] { prefetch_w(pre, p.L[l + 1].wv, E, INNER, B, tid); }))) break;
        } else {
            pre.item = -1;
            if (!(ok = grid_barrier(gs, tid, &s_ok))) break;
        }
        THMR_STAMP();
    }
    if (ok) {
        // consumers of the decoder output: the four read-outs as one (31 + zero row, 1024) matrix (token_head.py:99-105) and
        // the classifier's first Linear 1024 -> 160*64 (token_classifier.py:71-73); 2 + 640 column tiles dealt to the workgroups
        gemv_multi(p.dx, p.ro_w, p.ro_b, p.ro, p.mt_w, p.mt_b, p.mt, 2, 2 + 640, B, red, tid, p.mixer_cluster != 0);
    }
    THMR_STAMP();
    if (ok && p.mixer_cluster != 0) {
        ok = grid_barrier(gs, tid, &s_ok);                   // mixer_trans' Linear output complete
        if (ok) ok = mixer_cluster_stage(p, gs, tid, &s_ok, mixY, mixU, mixX, mixred);
    }
    THMR_STAMP();
    // exit protocol: the LAST workgroup to leave publishes the last epoch as the next kernel's generation
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&p.sync[2], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(&p.sync[1], gs.base + gs.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync[2], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

constexpr int kDecMinGrid = 128;    // the final gemv_multi streams 42 MB of mixer_trans weights: 128 workgroups take it 2 % faster than 64 at B <= 16 (profiles/r2y_decoder_min_grid.log)
int launch_decoder_fused(const DecParams& p, hipStream_t s) {
    if (p.B < 1 || p.depth < 1 || p.depth > 6) return -1;
    // one workgroup per (column tile, 16-row sub-tile) of the widest step, up to one per CU
    // (the grid barrier needs every workgroup resident: never more workgroups than the device has CUs; any grid size works,
    // the steps deal their items round-robin)
    const int nsub = (p.B + 15) / 16;
    int grid = NBLK * (nsub < 4 ? nsub : 4);
    static const int min_grid = [] { const char* e = thmr_knob("THMR_DEC_MIN_GRID"); return e ? atoi(e) : kDecMinGrid; }();   // A/B knob
    if (grid < min_grid) grid = min_grid;
    if (p.mixer_cluster != 0 && p.mixer_cluster != 10 && p.mixer_cluster != 5 && p.mixer_cluster != 2) return -1;
    if (grid < p.mixer_cluster * p.B) grid = p.mixer_cluster * p.B;     // the distributed mixer tail: mixer_cluster workgroups per crop
    if (p.max_blocks > 0 && grid > p.max_blocks) grid = p.max_blocks;
    if (grid > 256) grid = 256;                  // workgroup 0 polls one arrival flag per thread pair at most; flags[256]
    if (grid < p.mixer_cluster * p.B) return -1;                        // the caller picks a split that fits (engine.hip head_forward)
    // THMR_DEC_COOP=1 (A/B knob): cooperative launch — the runtime then refuses a grid that cannot be co-resident instead of
    // letting the bounded barrier find out.  The default is a plain launch of a grid sized from the occupancy query
    // (decoder_max_coresident_blocks, engine.hip finalize), which is the same guarantee for everything this process controls.
#ifdef THMR_EXPERIMENTS
    static const bool coop = [] { const char* e = thmr_knob("THMR_DEC_COOP"); return e && e[0] == '1'; }();
    if (coop) {
        DecParams pc = p;
        void* args[] = {&pc};
        return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&decoder_persistent_kernel), dim3(grid), dim3(NWAVE * 64), args, 0, s) == hipSuccess ? 0 : -2;
    }
#endif
    hipLaunchKernelGGL(decoder_persistent_kernel, dim3(grid), dim3(NWAVE * 64), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Workgroups of the persistent decoder kernel that can be resident on the device at once (occupancy query x CUs): the grid
// barrier needs every workgroup of a launch resident, so launch_decoder_fused never launches more than this (DecParams::max_blocks).
int decoder_max_coresident_blocks(int device) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decoder_persistent_kernel, NWAVE * 64, 0) != hipSuccess || per_cu < 1) return -2;
    // Only "at least one per CU" is taken from the query: the API over-reports by one block per CU in some SGPR ranges on this
    // ROCm (MI355X_MICROARCH.md, correctness boundaries), and one workgroup per CU is what the step scheduling assumes anyway
    // (bytes per CU bound the steps; a second workgroup would only share the CU).
    return cus;
}
