// HBM-bound row kernels of the TokenHMR path (LayerNorm, im2col, transposes, softmax+argmax, conv gathers).
// All are one-wave-per-row (64 lanes) or one-thread-per-element kernels with 16-byte coalesced accesses.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// nn.LayerNorm (biased variance): vit.py:136,144,252 (eps 1e-6); pose_transformer.py PreNorm :27-37,
// heads/modules.py:17,50,52 (eps 1e-5).  One wave per row, the row stays in registers (NV float4 per lane,
// D = NV*256): one HBM read + one write per element.
template <int NV>
__global__ __launch_bounds__(256) void ln_wave_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y, int rows,
                                                      float eps, int relu) {
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
    float* yr = y + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if (relu) o[e] = fmaxf(o[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(yr + c) = o;
    }
}

// any D (64, 10240, ...): one wave per row, strided scalar accesses, three passes over an L2-resident row
__global__ __launch_bounds__(256) void ln_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int rows,
                                                         int D, float eps, int relu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)row * D;
    float sum = 0.f;
    for (int i = lane; i < D; i += 64) sum += xr[i];
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float d = xr[i] - mean;
        sq += d * d;
    }
    const float var = wave_sum(sq) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + (int64_t)row * D;
    for (int i = lane; i < D; i += 64) {
        float o = (xr[i] - mean) * rstd * gamma[i] + beta[i];
        if (relu) o = fmaxf(o, 0.f);
        yr[i] = o;
    }
}

// MixerLayer: s = x + y ; z = LayerNorm64(s)   (heads/modules.py:59: layernorm2(x + y)); D = 64, lane per element
__global__ __launch_bounds__(256) void add_ln64_kernel(const float* __restrict__ x, const float* __restrict__ yv,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ s_out, float* __restrict__ z_out, int rows,
                                                       float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)row * 64 + lane;
    const float s = x[o] + yv[o];
    const float mean = wave_sum(s) * (1.0f / 64);
    const float d = s - mean;
    const float var = wave_sum(d * d) * (1.0f / 64);
    const float rstd = 1.0f / sqrtf(var + eps);
    s_out[o] = s;
    z_out[o] = d * rstd * gamma[lane] + beta[lane];
}

// ------------------------------------------------------------------------------------------------ patch-embed im2col
// vit.py:341 x[:,:,:,32:-32] ; :168 Conv2d(3,1280,k16,s16,p2) -> rows (b, py*12+px), cols c*256 + ky*16 + kx.
// Zero padding is applied on the SLICED 192-wide window (cols -2,-1 are zeros, not image cols 30,31).
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, float* __restrict__ A, int B) {
    // one thread per float4 of A: 768/4 = 192 float4 per row
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * 192 * 192;
    if (idx >= total) return;
    const int c4 = (int)(idx % 192);
    const int64_t rowi = idx / 192;
    const int tok = (int)(rowi % 192), b = (int)(rowi / 192);
    const int py = tok / 12, px = tok % 12;
    const int k = c4 * 4, c = k >> 8, ky = (k >> 4) & 15, kx0 = k & 15;
    const int iy = py * 16 + ky - 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < 256) {
        const float* src = img + (((int64_t)b * 3 + c) * 256 + iy) * 256 + 32;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = px * 16 + kx0 + e - 2;
            if (ix >= 0 && ix < 192) v[e] = src[ix];
        }
    }
    *reinterpret_cast<f32x4*>(A + rowi * 768 + k) = v;
}

// The same operand written directly as the split3 A of the patch-embed GEMM of the default mode ([M][768 / 8][3][8] bf16: three bf16 pieces per
// pixel, exact to 2^-24): one thread per k-group of 8 consecutive kx of one (row, c, ky) = four aligned 8-byte reads (the window starts at
// image column 30 + 16 px + kx0: even) and 48 contiguous bytes written.  No fp32 copy of the im2col matrix exists in that mode.
__global__ __launch_bounds__(256) void im2col_patch_split3_kernel(const float* __restrict__ img, char* __restrict__ A, int B) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * 192 * 96;
    if (idx >= total) return;
    const int g8 = (int)(idx % 96);
    const int64_t rowi = idx / 96;
    const int tok = (int)(rowi % 192), b = (int)(rowi / 192);
    const int py = tok / 12, px = tok % 12;
    const int k = g8 * 8, c = k >> 8, ky = (k >> 4) & 15, kx0 = k & 15;
    const int iy = py * 16 + ky - 2;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < 256) {
        const float* src = img + (((int64_t)b * 3 + c) * 256 + iy) * 256 + 32;
        const int ix0 = px * 16 + kx0 - 2;                  // even; a pair (ix, ix + 1) is inside [0, 192) or outside as a whole
        f32x2 p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = ix0 + 2 * e;
            p[e] = (ix >= 0 && ix < 192) ? *reinterpret_cast<const f32x2*>(src + ix) : f32x2{0.f, 0.f};
        }
        lo = f32x4{p[0].x, p[0].y, p[1].x, p[1].y};
        hi = f32x4{p[2].x, p[2].y, p[3].x, p[3].y};
    }
    store_split3_oct(A + rowi * (768 * 6), k, lo, hi);
}

// ------------------------------------------------------------------------------------------------ batched transpose
// (Bn, R, C) -> (Bn, C, R); MixerLayer y.transpose(2,1) (heads/modules.py:56-58)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* src = in + (int64_t)b * R * C;
    float* dst = out + (int64_t)b * R * C;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < R && c < C) tile[i][tx] = src[(int64_t)r * C + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dst[(int64_t)c * R + r] = tile[tx][i];
    }
}

// ------------------------------------------------------------------------------------------------ softmax + argmax (2048)
// token_classifier.py:104 cls_logits.softmax(-1) over 2048 classes; also emits the build-defined token index
// argmax_k logits (lowest index on ties).  One wave per row, the row (32 floats/lane) in registers.
__global__ __launch_bounds__(256) void softmax_argmax2048_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                                 int32_t* __restrict__ idx, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* lr = logits + (int64_t)row * 2048;
    f32x4 v[8];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(lr + (i * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = (i * 64 + lane) * 4 + e;
            if (v[i][e] > m) { m = v[i][e]; am = k; }       // ascending k within a lane: first max kept
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] = expf(v[i][e] - m);
            sum += v[i][e];
        }
    sum = wave_sum(sum);
    if (probs) {
        float* pr = probs + (int64_t)row * 2048;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i][e] / sum;
            *reinterpret_cast<f32x4*>(pr + (i * 64 + lane) * 4) = o;
        }
    }
    if (idx && lane == 0) idx[row] = am;
}

// ------------------------------------------------------------------------------------------------ Conv1d(k=3) gather
// VQ decoder, channels-last: builds the GEMM A operand of Conv1d(C -> *, k3, pad = dil, dilation = dil) applied to
// the nearest-resampled (optionally pre-ReLU'd) signal (vanilla_pose_vqvae.py:135-154, resnet.py:55-68):
//   out[b][t][dk*C + c] = f(in[b][src[t + (dk-1)*dil]][c])  if 0 <= t + (dk-1)*dil < Tout else 0
// src = nn.Upsample(size) nearest index table (identity when no resample).
__global__ __launch_bounds__(256) void conv3_gather_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const int32_t* __restrict__ src, int Bn, int Tin, int Tout,
                                                           int C, int dil, int prerelu) {
    const int c4n = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)Bn * Tout * 3 * c4n;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    int64_t rest = idx / c4n;
    const int dk = (int)(rest % 3);
    rest /= 3;
    const int t = (int)(rest % Tout), b = (int)(rest / Tout);
    const int tp = t + (dk - 1) * dil;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tp >= 0 && tp < Tout) {
        const int ts = src ? src[tp] : tp;
        v = *reinterpret_cast<const f32x4*>(in + ((int64_t)b * Tin + ts) * C + c4 * 4);
        if (prerelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
    }
    *reinterpret_cast<f32x4*>(out + ((int64_t)b * Tout + t) * (3 * C) + dk * C + c4 * 4) = v;
}

// weight repacks done once at finalize: Conv1d weight [co][ci][k] -> [co][k*ci_n + ci]
__global__ void conv_repack_kernel(const float* __restrict__ w, float* __restrict__ wp, int co_n, int ci_n, int kk) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)co_n * ci_n * kk;
    if (idx >= total) return;
    const int k = (int)(idx % kk);
    const int ci = (int)((idx / kk) % ci_n);
    const int co = (int)(idx / ((int64_t)kk * ci_n));
    wp[((int64_t)co * kk + k) * ci_n + ci] = w[idx];
}

// General Conv1d gather for the tokenizer ENCODER (vanilla_pose_vqvae.py:66-88): kernel size ks, stride, padding,
// optional nearest-resample table, input channels C zero-padded to Cp (so that K = ks*Cp is a multiple of 32):
//   out[b][t][kk*Cp + c] = in[b][src[tp]][c]   with tp = t*stride - pad + kk, valid if 0 <= tp < Tsrc and c < C
__global__ __launch_bounds__(256) void conv_gather_general_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  const int32_t* __restrict__ src, int Bn, int Tin, int Tsrc,
                                                                  int Tout, int C, int Cp, int ks, int stride, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)Bn * Tout * ks * Cp;
    if (idx >= total) return;
    const int c = (int)(idx % Cp);
    int64_t rest = idx / Cp;
    const int kk = (int)(rest % ks);
    rest /= ks;
    const int t = (int)(rest % Tout), b = (int)(rest / Tout);
    const int tp = t * stride - pad + kk;
    float v = 0.f;
    if (c < C && tp >= 0 && tp < Tsrc) {
        const int ts = src ? src[tp] : tp;
        v = in[((int64_t)b * Tin + ts) * C + c];
    }
    out[idx] = v;
}

// Conv1d weight [co][ci][k] -> [co][k*cp + ci], ci zero-padded to cp
__global__ void conv_repack_pad_kernel(const float* __restrict__ w, float* __restrict__ wp, int co_n, int ci_n, int cp, int kk) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)co_n * kk * cp;
    if (idx >= total) return;
    const int ci = (int)(idx % cp);
    const int k = (int)((idx / cp) % kk);
    const int co = (int)(idx / ((int64_t)cp * kk));
    wp[idx] = ci < ci_n ? w[((int64_t)co * ci_n + ci) * kk + k] : 0.f;
}

// ---- split-K reducers for gemm_ring_kernel (gemm_f32.hip): part[S][M][N] summed over s in a FIXED order ----
// generic: C = epilogue(sum_s part[s]) — float4 per thread
template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmArgs a, const float* __restrict__ part, int S) {
    const int n4 = a.N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) * 4;
    const int64_t mn = (int64_t)a.M * a.N;
    const float* p = part + (int64_t)m * a.N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(p + s * mn);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != EPI_NONE) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gemm_epilogue<EPI>(a, v[e], bias[e], m, n + e);
    *reinterpret_cast<f32x4*>(a.C + (int64_t)m * a.ldc + n) = o;
}

// fused for the ViT residual stream (vit.py:149-150 followed by the next norm, :149/:150/:335):
//   x_new = resid + (sum_s part[s] + bias);  y = LayerNorm(x_new)        one wave per row, D = NV*256
// ST > 0: the split factor is a compile-time constant, so all ST*NV partial loads of a lane are issued before the first add
// (the engine's factor 4: 8.8 -> ~5 us at 192 rows, where the kernel is one dependent-load chain per wave); ST = 0: runtime S.
// SPLIT: y is a split3 operand [rows][D/8][3][8] bf16 (gemm_split.hip) instead of fp32 — the same values as three bf16 pieces each
template <int NV, int ST, bool SPLIT = false>
__global__ __launch_bounds__(256) void splitk_resid_ln_kernel(const float* __restrict__ part, int S, int64_t mn,
                                                              const float* __restrict__ bias, const float* resid, float* xout,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, int rows, float eps) {
    constexpr int D = NV * 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t ro = (int64_t)row * D;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 t = *reinterpret_cast<const f32x4*>(part + ro + c);
        if constexpr (ST > 0) {
            f32x4 ps[ST > 1 ? ST - 1 : 1];
#pragma unroll
            for (int s = 1; s < ST; ++s) ps[s - 1] = *reinterpret_cast<const f32x4*>(part + s * mn + ro + c);
#pragma unroll
            for (int s = 1; s < ST; ++s) t += ps[s - 1];
        } else {
            for (int s = 1; s < S; ++s) t += *reinterpret_cast<const f32x4*>(part + s * mn + ro + c);
        }
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
        const f32x4 r = *reinterpret_cast<const f32x4*>(resid + ro + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = r[e] + (t[e] + b[e]);
        *reinterpret_cast<f32x4*>(xout + ro + c) = t;
        v[i] = t;
        sum += (t[0] + t[1]) + (t[2] + t[3]);
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
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        if constexpr (SPLIT) store_split3_quad(reinterpret_cast<char*>(y) + ro * 6, c, o);
        else *reinterpret_cast<f32x4*>(y + ro + c) = o;
    }
}

}  // namespace

int launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, int relu,
                     hipStream_t s) {
    if (rows <= 0 || D <= 0) return -1;
    dim3 grid((rows + 3) / 4), block(256);
    if (D == 1280)
        hipLaunchKernelGGL(ln_wave_kernel<5>, grid, block, 0, s, x, g, b, y, rows, eps, relu);
    else if (D == 1024)
        hipLaunchKernelGGL(ln_wave_kernel<4>, grid, block, 0, s, x, g, b, y, rows, eps, relu);
    else
        hipLaunchKernelGGL(ln_generic_kernel, grid, block, 0, s, x, g, b, y, rows, D, eps, relu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_splitk_epilogue(const GemmArgs& a, int epi, const float* part, int S, hipStream_t s) {
    if ((a.N & 3) || (a.ldc & 3) || S < 1) return -1;
    const int64_t total = (int64_t)a.M * (a.N >> 2);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define THMR_SK_CASE(E) \
    case E: hipLaunchKernelGGL(splitk_epilogue_kernel<E>, grid, block, 0, s, a, part, S); break;
    switch (epi) {
        THMR_SK_CASE(EPI_NONE)
        THMR_SK_CASE(EPI_BIAS)
        THMR_SK_CASE(EPI_BIAS_GELU)
        THMR_SK_CASE(EPI_BIAS_RELU)
        THMR_SK_CASE(EPI_BIAS_RESID)
        THMR_SK_CASE(EPI_BIAS_QSCALE)
        default: return -1;
    }
#undef THMR_SK_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_splitk_resid_ln(const float* part, int S, int rows, int D, const float* bias, const float* resid, float* xout,
                           const float* gamma, const float* beta, float* y, float eps, hipStream_t s, bool y_is_split3) {
    if (rows <= 0 || S < 1 || D != 1280) return -1;
    if (y_is_split3) {      // the split3 mode's ranges: 2 ways (5 ... 31 crops) or 4 ways (3 and 4 crops)
        if (S == 4)
            hipLaunchKernelGGL((splitk_resid_ln_kernel<5, 4, true>), dim3((rows + 3) / 4), dim3(256), 0, s, part, S, (int64_t)rows * D, bias,
                               resid, xout, gamma, beta, y, rows, eps);
        else if (S == 2)
            hipLaunchKernelGGL((splitk_resid_ln_kernel<5, 2, true>), dim3((rows + 3) / 4), dim3(256), 0, s, part, S, (int64_t)rows * D, bias,
                               resid, xout, gamma, beta, y, rows, eps);
        else
            return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (S == 4)
        hipLaunchKernelGGL((splitk_resid_ln_kernel<5, 4>), dim3((rows + 3) / 4), dim3(256), 0, s, part, S, (int64_t)rows * D, bias,
                           resid, xout, gamma, beta, y, rows, eps);
    else if (S == 2)
        hipLaunchKernelGGL((splitk_resid_ln_kernel<5, 2>), dim3((rows + 3) / 4), dim3(256), 0, s, part, S, (int64_t)rows * D, bias,
                           resid, xout, gamma, beta, y, rows, eps);
    else
        hipLaunchKernelGGL((splitk_resid_ln_kernel<5, 0>), dim3((rows + 3) / 4), dim3(256), 0, s, part, S, (int64_t)rows * D, bias,
                           resid, xout, gamma, beta, y, rows, eps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_add_ln64(const float* x, const float* y, const float* g, const float* b, float* s_out, float* z_out, int rows,
                    float eps, hipStream_t s) {
    hipLaunchKernelGGL(add_ln64_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, g, b, s_out, z_out, rows, eps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_im2col_patch(const float* img, float* A, int B, hipStream_t s) {
    const int64_t total = (int64_t)B * 192 * 192;
    hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, A, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_im2col_patch_split3(const float* img, void* A_split, int B, hipStream_t s) {
    const int64_t total = (int64_t)B * 192 * 96;
    hipLaunchKernelGGL(im2col_patch_split3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, reinterpret_cast<char*>(A_split), B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_transpose(const float* in, float* out, int Bn, int R, int C, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, Bn), dim3(256), 0, s, in, out, R, C);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_softmax_argmax2048(const float* logits, float* probs, int32_t* idx, int rows, hipStream_t s) {
    hipLaunchKernelGGL(softmax_argmax2048_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, probs, idx, rows);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_conv3_gather(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tout, int C, int dil,
                        int prerelu, hipStream_t s) {
    const int64_t total = (int64_t)Bn * Tout * 3 * (C / 4);
    hipLaunchKernelGGL(conv3_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, src, Bn, Tin,
                       Tout, C, dil, prerelu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_conv_repack(const float* w, float* wp, int co, int ci, int kk, hipStream_t s) {
    const int64_t total = (int64_t)co * ci * kk;
    hipLaunchKernelGGL(conv_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wp, co, ci, kk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_conv_gather_general(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tsrc, int Tout, int C,
                               int Cp, int ks, int stride, int pad, hipStream_t s) {
    const int64_t total = (int64_t)Bn * Tout * ks * Cp;
    hipLaunchKernelGGL(conv_gather_general_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, src, Bn, Tin,
                       Tsrc, Tout, C, Cp, ks, stride, pad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_conv_repack_pad(const float* w, float* wp, int co, int ci, int cp, int kk, hipStream_t s) {
    const int64_t total = (int64_t)co * kk * cp;
    hipLaunchKernelGGL(conv_repack_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wp, co, ci, cp, kk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
