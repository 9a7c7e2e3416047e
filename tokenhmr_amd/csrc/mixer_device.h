// Device-side building blocks of the MLP-Mixer stack (token_classifier.py:92-101, heads/modules.py:11-24,55-63), shared by
//   * mixer_stack_kernel (mixer_fused.hip): one workgroup per crop, the whole 160 x 64 tile in its LDS, and
//   * the distributed tail of the persistent decoder kernel (decoder_fused.hip): ten workgroups per crop, one 16-token tile each.
// Both run the SAME functions on the same operands in the same order, so a crop's result is bit-identical whichever form serves
// its batch.  v_mfma_f32_16x16x4_f32 (exact fp32), 8 waves per workgroup.
#pragma once
#include "common.h"

namespace mixer {

constexpr int T = 160, H = 64, TI = 64, HI = 256, LD = 68;     // LD: padded LDS row (floats), multiple of 4 for ds_read_b128
constexpr float EPS = 1e-5f;
constexpr int NW = 8, NT = NW * 64;

__device__ __forceinline__ f32x4 mfma4(const f32x4& a, const f32x4& b, f32x4 acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
    return acc;
}

// mixer_trans: LayerNorm over all 160*64 values of the crop + ReLU (FCBlock, heads/modules.py:17-18).  512 threads; thread tid
// owns elements (i * 512 + tid) * 4 ... + 3, i < 5, and gets their normalised values in o[i].  DEV: the input row was written by
// other workgroups of the running kernel (device-scope loads).  redbuf: NW floats of LDS; contains two workgroup barriers.
template <bool DEV>
__device__ __forceinline__ void trans_ln(const float* xr, const float* __restrict__ tln_w, const float* __restrict__ tln_b, float* redbuf, int tid,
                                         int lane, int wave, f32x4 (&o)[5]) {
    f32x4 v[5];                                     // 10240 / 512 threads = 20 values = 5 float4 per thread
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if constexpr (DEV) v[i] = ld_dev4(xr + (i * NT + tid) * 4);
        else v[i] = *reinterpret_cast<const f32x4*>(xr + (i * NT + tid) * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    sum = wave_sum(sum);
    if (lane == 0) redbuf[wave] = sum;
    __syncthreads();
    float tot = redbuf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += redbuf[w];
    const float mean = tot * (1.0f / (T * H));
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq += d * d;
        }
    sq = wave_sum(sq);
    if (lane == 0) redbuf[wave] = sq;
    __syncthreads();
    float tsq = redbuf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tsq += redbuf[w];
    const float rstd = 1.0f / sqrtf(tsq * (1.0f / (T * H)) + EPS);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int e0 = (i * NT + tid) * 4;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(tln_w + e0), bt = *reinterpret_cast<const f32x4*>(tln_b + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i][e] = fmaxf((v[i][e] - mean) * rstd * gm[e] + bt[e], 0.f);
    }
}

// LayerNorm over the 64 hidden values of ONE token row by one wave, lane = hidden index (heads/modules.py:50,52)
__device__ __forceinline__ float ln_row_value(float v, float gamma, float beta) {
    const float mean = wave_sum(v) * (1.0f / H);
    const float d = v - mean;
    const float var = wave_sum(d * d) * (1.0f / H);
    return d * (1.0f / sqrtf(var + EPS)) * gamma + beta;
}

// token mixing 1, one (h-tile, j-tile) of u[h][j] = gelu(sum_t y[t][h] Wt1[j][t] + bt1[j]); Y = all 160 LayerNorm-ed rows (LDS),
// whose columns are the A operand (no transpose is materialised); tile in [0, 16)
__device__ __forceinline__ void token_mix1_tile(const float* Y, float* U, const MixerLayerW& w, int tile, int l15, int g) {
    const int h0 = (tile >> 2) * 16, j0 = (tile & 3) * 16;
    const float* wr = w.wt1 + (int64_t)(j0 + l15) * T + g * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < T / 16; ++ks) {
        const f32x4 wf = *reinterpret_cast<const f32x4*>(wr + ks * 16);
        f32x4 af;                                              // A[h][k = t]: down a column of the token-major tile
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) af[tt] = Y[(ks * 16 + g * 4 + tt) * LD + h0 + l15];
        acc = mfma4(af, wf, acc);
    }
    const float bj = w.bt1[j0 + l15];
#pragma unroll
    for (int r = 0; r < 4; ++r) U[(h0 + g * 4 + r) * LD + j0 + l15] = gelu_erf(acc[r] + bj);
}

// token mixing 2, one (h-tile, token tile): z[h][t] = sum_j u[h][j] Wt2[t][j] + bt2[t];  s[t][h] = x[t][h] + z[h][t].
// Xt / St: row t0 of the residual tile and of the destination (rows of LD floats); t0 = global token index of the tile
__device__ __forceinline__ void token_mix2_tile(const float* U, const float* Xt, float* St, const MixerLayerW& w, int h0, int t0, int l15, int g) {
    const float* wr = w.wt2 + (int64_t)(t0 + l15) * TI + g * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < TI / 16; ++ks) {
        const f32x4 wf = *reinterpret_cast<const f32x4*>(wr + ks * 16);
        const f32x4 af = *reinterpret_cast<const f32x4*>(&U[(h0 + l15) * LD + ks * 16 + g * 4]);
        acc = mfma4(af, wf, acc);
    }
    const float bt = w.bt2[t0 + l15];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = l15 * LD + h0 + g * 4 + r;               // D: row h0 + 4g + r, column t0 + l15
        St[o] = Xt[o] + (acc[r] + bt);
    }
}

// channel mixing of 16 tokens by ONE wave, wave-local from LayerNorm2 to the residual output:
//   z0 = LayerNorm2(s) in the B-operand layout (a token's 64 values = 16 registers in each of the 4 lanes of its column),
//   zh^T = gelu(Wc1 z0^T + bc1) (256 x 16, kept in registers as the B operand of the second product), out^T = Wc2 zh^T + bc2,
//   x_new = s + out.  St: the 16 rows of s (LDS), Xt: where x_new goes.  Weight fragments run PF k-groups ahead in a register
//   ring; the compiler fences keep hipcc from hoisting all 128 fragment loads (it spilled 470 registers doing that).
__device__ __forceinline__ void channel_mix_tile(const float* St, float* Xt, const MixerLayerW& w, int l15, int g) {
    constexpr int PF = 3;
    f32x4 zb[4];                                               // B[k = hidden][col = token l15]
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        zb[ks] = *reinterpret_cast<const f32x4*>(&St[l15 * LD + ks * 16 + g * 4]);
        sum += (zb[ks][0] + zb[ks][1]) + (zb[ks][2] + zb[ks][3]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / H);
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = zb[ks][e] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / H) + EPS);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(w.ln2w + ks * 16 + g * 4), bt = *reinterpret_cast<const f32x4*>(w.ln2b + ks * 16 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) zb[ks][e] = (zb[ks][e] - mean) * rstd * gm[e] + bt[e];
    }
    f32x4 zh[16];
    {
        f32x4 ring[PF + 1][4];
        const float* wbase = w.wc1 + (int64_t)l15 * H + g * 4;              // A[row = channel][k = hidden]
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ring[i][ks] = *reinterpret_cast<const f32x4*>(wbase + (int64_t)i * 16 * H + ks * 16);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            if (nt + PF < 16) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    ring[(nt + PF) % (PF + 1)][ks] = *reinterpret_cast<const f32x4*>(wbase + (int64_t)(nt + PF) * 16 * H + ks * 16);
            }
            asm volatile("" ::: "memory");
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma4(ring[nt % (PF + 1)][ks], zb[ks], acc);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(w.bc1 + nt * 16 + g * 4);   // rows 4g + r of this tile
#pragma unroll
            for (int r = 0; r < 4; ++r) zh[nt][r] = gelu_erf(acc[r] + bv[r]);
        }
    }
    f32x4 oacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        f32x4 ring[PF + 1][4];                                              // [k-group nt][out-channel tile ct]
        const float* wbase = w.wc2 + (int64_t)l15 * HI + g * 4;             // A[row = out channel][k = hidden channel]
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) ring[i][ct] = *reinterpret_cast<const f32x4*>(wbase + (int64_t)ct * 16 * HI + i * 16);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            if (nt + PF < 16) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    ring[(nt + PF) % (PF + 1)][ct] = *reinterpret_cast<const f32x4*>(wbase + (int64_t)ct * 16 * HI + (nt + PF) * 16);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) oacc[ct] = mfma4(ring[nt % (PF + 1)][ct], zh[nt], oacc[ct]);
        }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(w.bc2 + ct * 16 + g * 4);
        const int o = l15 * LD + ct * 16 + g * 4;              // D: row = channel ct*16 + 4g + r, column = token
        const f32x4 sv = *reinterpret_cast<const f32x4*>(&St[o]);
        f32x4 xo;
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[r] = sv[r] + (oacc[ct][r] + bv[r]);  // out = (x + y) + z
        *reinterpret_cast<f32x4*>(&Xt[o]) = xo;
    }
}

// The same channel mixing of ONE 16-token tile spread over the 8 waves of a workgroup (the distributed form owns a single tile: on
// one wave its 512 dependent MFMAs behind L2-latency-bound weight loads took ~17 us per layer).  Per accumulator the operation
// sequence is exactly channel_mix_tile's, so the result is bit-identical:
//   * every wave recomputes z0 = LayerNorm2(s) (same instructions, same values);
//   * hidden tile nt of zh^T = gelu(Wc1 z0^T + bc1) is computed by wave nt % 8 and parked in LDS in register layout (ZH: 16 tiles x
//     64 lanes x 4 floats = 16 KB) — the value a lane reads back is the one a lane of the single wave would hold in zh[nt];
//   * out-channel tile ct of out^T = Wc2 zh^T is accumulated by wave ct over nt = 0 ... 15 in order.
// Contains two workgroup barriers; all 512 threads must call it.
__device__ __forceinline__ void channel_mix_tile_8waves(const float* St, float* Xt, float* ZH, const MixerLayerW& w, int wave, int lane, int l15, int g) {
    f32x4 zb[4];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        zb[ks] = *reinterpret_cast<const f32x4*>(&St[l15 * LD + ks * 16 + g * 4]);
        sum += (zb[ks][0] + zb[ks][1]) + (zb[ks][2] + zb[ks][3]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / H);
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = zb[ks][e] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / H) + EPS);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(w.ln2w + ks * 16 + g * 4), bt = *reinterpret_cast<const f32x4*>(w.ln2b + ks * 16 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) zb[ks][e] = (zb[ks][e] - mean) * rstd * gm[e] + bt[e];
    }
    {
        const float* wbase = w.wc1 + (int64_t)l15 * H + g * 4;              // A[row = channel][k = hidden]
        f32x4 wf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[i][ks] = *reinterpret_cast<const f32x4*>(wbase + (int64_t)(wave + i * NW) * 16 * H + ks * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nt = wave + i * NW;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma4(wf[i][ks], zb[ks], acc);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(w.bc1 + nt * 16 + g * 4);
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = gelu_erf(acc[r] + bv[r]);
            *reinterpret_cast<f32x4*>(&ZH[(nt * 64 + lane) * 4]) = z;
        }
    }
    f32x4 wq[16];
    if (wave < 4) {                                                         // this wave's Wc2 fragments, requested before the barrier
        const float* wbase = w.wc2 + (int64_t)(wave * 16 + l15) * HI + g * 4;  // A[row = out channel][k = hidden channel]
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) wq[nt] = *reinterpret_cast<const f32x4*>(wbase + nt * 16);
    }
    __syncthreads();
    if (wave < 4) {
        const int ct = wave;
        f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) oacc = mfma4(wq[nt], *reinterpret_cast<const f32x4*>(&ZH[(nt * 64 + lane) * 4]), oacc);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(w.bc2 + ct * 16 + g * 4);
        const int o = l15 * LD + ct * 16 + g * 4;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(&St[o]);
        f32x4 xo;
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[r] = sv[r] + (oacc[r] + bv[r]);      // out = (x + y) + z
        *reinterpret_cast<f32x4*>(&Xt[o]) = xo;
    }
    __syncthreads();
}

// mixer_norm_layer for 16 tokens by one wave: Linear(64,64) + LayerNorm(64) + ReLU per token, transposed product (rows = out
// channel).  Xt: the 16 rows (LDS), orow: the 16 output rows of H floats in global memory
__device__ __forceinline__ void norm_layer_tile(const float* Xt, float* orow, const MixerParams& p, int l15, int g) {
    f32x4 xb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const f32x4*>(&Xt[l15 * LD + ks * 16 + g * 4]);
    f32x4 y[4];
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float* wr = p.wn + (int64_t)(ct * 16 + l15) * H + g * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mfma4(*reinterpret_cast<const f32x4*>(wr + ks * 16), xb[ks], acc);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bn + ct * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[ct][r] = acc[r] + bv[r];
        sum += (y[ct][0] + y[ct][1]) + (y[ct][2] + y[ct][3]);
    }
    // a token's 64 values live in the 4 lanes (g = 0..3) of its column l15: 16 registers each
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / H);
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = y[ct][r] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / H) + EPS);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.nln_w + ct * 16 + g * 4), bt = *reinterpret_cast<const f32x4*>(p.nln_b + ct * 16 + g * 4);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf((y[ct][r] - mean) * rstd * gm[r] + bt[r], 0.f);
        *reinterpret_cast<f32x4*>(orow + l15 * H + ct * 16 + g * 4) = o;
    }
}

}  // namespace mixer
