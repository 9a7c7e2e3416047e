// The MLP-Mixer stack of the token classifier in ONE kernel, one workgroup per crop, the (160 tokens x 64 hidden) tile
// resident in LDS from the first LayerNorm to the last.
//
// Replaces tokenhmr/lib/models/heads/token_classifier.py:92-101: FCBlock mixer_trans' LayerNorm(10240) + ReLU (its Linear runs
// at the end of the persistent decoder kernel), the four MixerLayers (heads/modules.py:55-63: LayerNorm -> token-mixing MLP on
// the transposed tile -> x + y -> LayerNorm -> channel-mixing MLP -> (x + y) + z) and FCBlock mixer_norm_layer
// (Linear(64,64) + LayerNorm + ReLU, heads/modules.py:11-24).  Round 1 ran this as 8 launches per mixer layer (LayerNorm,
// transpose, two GEMMs, transpose, add+LayerNorm, two GEMMs): ~40 launches and ~0.5 ms per 64 crops for 3.5 GFLOP.
//
// gfx950 design (v_mfma_f32_16x16x4_f32, exact fp32; 8 waves, 104 KB of LDS):
//   * token mixing:  u = gelu(y^T W1^T + b1) is a (64 hidden x 64) product over K = 160 tokens whose A operand is read
//     DOWN the columns of the LDS tile (no transpose is ever materialised); z = u W2^T + b2 comes out as (hidden x token)
//     tiles and is added to x while being scattered back token-major.
//   * channel mixing is per token, so a wave owns 16 tokens from LayerNorm-ed input to residual output with NO LDS round
//     trip for the 256-wide hidden activation: it computes the TRANSPOSED product zh^T = Wc1 z0^T, whose accumulator
//     registers (row = channel 4g + r, column = token) are exactly the B operand of out^T = Wc2 zh^T.
//   * weights stream from L2 as 16-byte k-permuted fragments straight into MFMA operands; every reduction has a fixed
//     order, and a crop never sees another crop's data (results do not depend on the batch).
#include "common.h"

namespace {

constexpr int T = 160, H = 64, TI = 64, HI = 256, LD = 68;     // LD: padded LDS row (floats), multiple of 4 for ds_read_b128
constexpr float EPS = 1e-5f;
constexpr int NW = 8, NT = NW * 64;

__device__ __forceinline__ f32x4 mfma4(const f32x4& a, const f32x4& b, f32x4 acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
    return acc;
}

// LayerNorm over the 64 hidden values of token rows, one wave per row, lane = hidden index (heads/modules.py:50,52)
__device__ __forceinline__ void ln_rows(const float* src, float* dst, const float* __restrict__ gw, const float* __restrict__ gb,
                                        int wave, int lane) {
    const float gamma = gw[lane], beta = gb[lane];
    for (int t = wave; t < T; t += NW) {
        const float v = src[t * LD + lane];
        const float mean = wave_sum(v) * (1.0f / H);
        const float d = v - mean;
        const float var = wave_sum(d * d) * (1.0f / H);
        dst[t * LD + lane] = d * (1.0f / sqrtf(var + EPS)) * gamma + beta;
    }
}

__global__ __launch_bounds__(NT) void mixer_stack_kernel(MixerParams p) {
    __shared__ __attribute__((aligned(16))) float X[T * LD];       // residual stream x (token-major)
    __shared__ __attribute__((aligned(16))) float Y[T * LD];       // LayerNorm1(x), later s = x + y
    __shared__ __attribute__((aligned(16))) float U[H * LD];       // token-mixing hidden activation (hidden x 64)
    __shared__ float redbuf[NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;

    // ---- mixer_trans: LayerNorm over all 160*64 values of the crop + ReLU (FCBlock, heads/modules.py:17-18) ----
    {
        const float* xr = p.mt + (int64_t)b * (T * H);
        f32x4 v[5];                                     // 10240 / 512 threads = 20 values = 5 float4 per thread
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + (i * NT + tid) * 4);
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
            const int e0 = (i * NT + tid) * 4, t = e0 >> 6, h = e0 & 63;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.tln_w + e0), bt = *reinterpret_cast<const f32x4*>(p.tln_b + e0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf((v[i][e] - mean) * rstd * gm[e] + bt[e], 0.f);
            *reinterpret_cast<f32x4*>(&X[t * LD + h]) = o;
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const MixerLayerW& w = p.L[l];
        // ---- y = LayerNorm1(x) ----
        ln_rows(X, Y, w.ln1w, w.ln1b, wave, lane);
        __syncthreads();
        // ---- token mixing 1: u[h][j] = gelu(sum_t y[t][h] Wt1[j][t] + bt1[j]); 16 (h-tile, j-tile) tiles, 2 per wave ----
#pragma unroll 1
        for (int tile = wave; tile < 16; tile += NW) {
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
        __syncthreads();
        // ---- token mixing 2: z[h][t] = sum_j u[h][j] Wt2[t][j] + bt2[t];  s[t][h] = x[t][h] + z[h][t]  -> Y ----
#pragma unroll 1
        for (int tile = wave; tile < 40; tile += NW) {
            const int h0 = (tile & 3) * 16, t0 = (tile >> 2) * 16;
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
                const int o = (t0 + l15) * LD + h0 + g * 4 + r;       // D: row h0 + 4g + r, column t0 + l15
                Y[o] = X[o] + (acc[r] + bt);
            }
        }
        __syncthreads();
        // ---- channel mixing, 16 tokens per wave pass, wave-local from LayerNorm2 to the residual output:
        //      z0 = LayerNorm2(s) in the B-operand layout (a token's 64 values = 16 registers in each of the 4 lanes of its
        //      column), zh^T = gelu(Wc1 z0^T + bc1) (256 x 16, kept in registers as the B operand of the second product),
        //      out^T = Wc2 zh^T + bc2, x_new = s + out.  Weight fragments run PF k-groups ahead in a register ring; the
        //      compiler fences keep hipcc from hoisting all 128 fragment loads (it spilled 470 registers doing that). ----
        constexpr int PF = 3;
#pragma unroll 1
        for (int tt0 = wave; tt0 < T / 16; tt0 += NW) {
            const int t0 = tt0 * 16;
            f32x4 zb[4];                                               // B[k = hidden][col = token l15]
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                zb[ks] = *reinterpret_cast<const f32x4*>(&Y[(t0 + l15) * LD + ks * 16 + g * 4]);
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
                const int o = (t0 + l15) * LD + ct * 16 + g * 4;       // D: row = channel ct*16 + 4g + r, column = token
                const f32x4 sv = *reinterpret_cast<const f32x4*>(&Y[o]);
                f32x4 xo;
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[r] = sv[r] + (oacc[ct][r] + bv[r]);  // out = (x + y) + z
                *reinterpret_cast<f32x4*>(&X[o]) = xo;
            }
        }
        __syncthreads();
    }

    // ---- mixer_norm_layer: Linear(64,64) + LayerNorm(64) + ReLU per token, transposed product again: rows = out channel ----
    float* orow = p.out + (int64_t)b * (T * H);
#pragma unroll 1
    for (int tt0 = wave; tt0 < T / 16; tt0 += NW) {
        const int t0 = tt0 * 16;
        f32x4 xb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const f32x4*>(&X[(t0 + l15) * LD + ks * 16 + g * 4]);
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
            *reinterpret_cast<f32x4*>(orow + (t0 + l15) * H + ct * 16 + g * 4) = o;
        }
    }
}

}  // namespace

int launch_mixer_fused(const MixerParams& p, int B, hipStream_t s) {
    if (B < 1) return -1;
    hipLaunchKernelGGL(mixer_stack_kernel, dim3(B), dim3(NT), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
