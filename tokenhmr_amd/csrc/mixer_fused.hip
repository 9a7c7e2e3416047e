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
#include "mixer_device.h"

namespace {

using namespace mixer;

__global__ __launch_bounds__(NT) void mixer_stack_kernel(MixerParams p) {
    __shared__ __attribute__((aligned(16))) float X[T * LD];       // residual stream x (token-major)
    __shared__ __attribute__((aligned(16))) float Y[T * LD];       // LayerNorm1(x), later s = x + y
    __shared__ __attribute__((aligned(16))) float U[H * LD];       // token-mixing hidden activation (hidden x 64)
    __shared__ float redbuf[NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;

    // ---- mixer_trans: LayerNorm over all 160*64 values of the crop + ReLU (FCBlock, heads/modules.py:17-18) ----
    {
        f32x4 o[5];
        trans_ln<false>(p.mt + (int64_t)b * (T * H), p.tln_w, p.tln_b, redbuf, tid, lane, wave, o);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e0 = (i * NT + tid) * 4, t = e0 >> 6, h = e0 & 63;
            *reinterpret_cast<f32x4*>(&X[t * LD + h]) = o[i];
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const MixerLayerW& w = p.L[l];
        // ---- y = LayerNorm1(x), one wave per row ----
        {
            const float gamma = w.ln1w[lane], beta = w.ln1b[lane];
            for (int t = wave; t < T; t += NW) Y[t * LD + lane] = ln_row_value(X[t * LD + lane], gamma, beta);
        }
        __syncthreads();
        // ---- token mixing 1: 16 (h-tile, j-tile) tiles, 2 per wave ----
#pragma unroll 1
        for (int tile = wave; tile < 16; tile += NW) token_mix1_tile(Y, U, w, tile, l15, g);
        __syncthreads();
        // ---- token mixing 2: s = x + z -> Y; 40 (h-tile, token-tile) tiles, 5 per wave ----
#pragma unroll 1
        for (int tile = wave; tile < 40; tile += NW) {
            const int h0 = (tile & 3) * 16, t0 = (tile >> 2) * 16;
            token_mix2_tile(U, X + t0 * LD, Y + t0 * LD, w, h0, t0, l15, g);
        }
        __syncthreads();
        // ---- channel mixing, 16 tokens per wave pass ----
#pragma unroll 1
        for (int tt0 = wave; tt0 < T / 16; tt0 += NW) channel_mix_tile(Y + tt0 * 16 * LD, X + tt0 * 16 * LD, w, l15, g);
        __syncthreads();
    }

    // ---- mixer_norm_layer ----
    float* orow = p.out + (int64_t)b * (T * H);
#pragma unroll 1
    for (int tt0 = wave; tt0 < T / 16; tt0 += NW) norm_layer_tile(X + tt0 * 16 * LD, orow + tt0 * 16 * H, p, l15, g);
}

}  // namespace

int launch_mixer_fused(const MixerParams& p, int B, hipStream_t s) {
    if (B < 1) return -1;
    hipLaunchKernelGGL(mixer_stack_kernel, dim3(B), dim3(NT), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
