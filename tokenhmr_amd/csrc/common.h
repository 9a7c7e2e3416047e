// Shared declarations for the gfx950 TokenHMR kernels.  CDNA4 only: wave = 64 lanes,
// fp32-input MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, exact fp32 == fmaf chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define THMR_WAVE 64

// ---- epilogue ids of the tiled / skinny GEMMs (also exposed through thmr_op_gemm) ----
enum GemmEpi {
    EPI_NONE = 0,        // C = acc
    EPI_BIAS = 1,        // C = acc + bias[n]
    EPI_BIAS_GELU = 2,   // C = gelu_erf(acc + bias[n])
    EPI_BIAS_RELU = 3,   // C = max(acc + bias[n], 0)
    EPI_BIAS_RESID = 4,  // C = resid[m][n] + (acc + bias[n])
    EPI_BIAS_QSCALE = 5, // C = (acc + bias[n]) * (n < qcols ? qscale : 1)
    EPI_BIAS_POS = 6,    // C = ((acc + bias[n]) + pos[1 + m % 192][n]) + pos[0][n]   (vit.py:327)
    EPI_NUM = 7
};

struct GemmArgs {
    const float* A;      // [M][lda]  K-contiguous
    const float* W;      // [N][ldw]  K-contiguous (torch Linear weight layout)
    const float* bias;   // [N] or null
    const float* resid;  // [M][ldr] (EPI_BIAS_RESID) or pos_embed [193][N] (EPI_BIAS_POS)
    float* C;            // [M][ldc]
    int64_t lda, ldw, ldc, ldr;
    int M, N, K;
    float qscale;
    int qcols;
};

__device__ __forceinline__ float gelu_erf(float x) {
    // torch.nn.GELU() default (approximate='none'): 0.5*x*(1+erf(x/sqrt(2)))
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int EPI>
__device__ __forceinline__ float gemm_epilogue(const GemmArgs& a, float acc, float bias, int m, int n) {
    float v = acc;
    if constexpr (EPI != EPI_NONE) v = acc + bias;
    if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
    if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
    if constexpr (EPI == EPI_BIAS_RESID) v = a.resid[(int64_t)m * a.ldr + n] + v;
    if constexpr (EPI == EPI_BIAS_QSCALE) v = (n < a.qcols) ? v * a.qscale : v;
    if constexpr (EPI == EPI_BIAS_POS) {
        const int t = m % 192;
        v = (v + a.resid[(int64_t)(1 + t) * a.N + n]) + a.resid[n];
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// host-side launch helpers (defined in the .hip files); all return 0 / negative
int launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s);          // gemm_f32.hip
int launch_gemm_skinny(const GemmArgs& a, int epi, hipStream_t s);                // gemm_skinny.hip
int launch_vit_attention(const float* qkv, float* out, int B, hipStream_t s);     // attention.hip
// rowops.hip
int launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, int relu,
                     hipStream_t s);
int launch_add_ln64(const float* x, const float* y, const float* g, const float* b, float* s_out, float* z_out, int rows,
                    float eps, hipStream_t s);
int launch_im2col_patch(const float* img, float* A, int B, hipStream_t s);
int launch_transpose(const float* in, float* out, int Bn, int R, int C, hipStream_t s);
int launch_softmax_argmax2048(const float* logits, float* probs, int32_t* idx, int rows, hipStream_t s);
int launch_conv3_gather(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tout, int C, int dil,
                        int prerelu, hipStream_t s);
int launch_conv_repack(const float* w, float* wp, int co, int ci, int kk, hipStream_t s);
int launch_conv_gather_general(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tsrc, int Tout, int C,
                               int Cp, int ks, int stride, int pad, hipStream_t s);
int launch_conv_repack_pad(const float* w, float* wp, int co, int ci, int cp, int kk, hipStream_t s);
// head.hip
int launch_decoder_init(const float* bias, const float* pos, float* x, int B, int E, hipStream_t s);
int launch_cross_attn(const float* q, const float* kv, int64_t ldkv, int koff, float* out, int B, hipStream_t s);
int launch_assemble(const float* ro, int ldro, const float* bpose, const float* init_pose, const float* init_betas,
                    const float* init_cam, float* pose6d, float* rotmat, float* betas, float* cam, float* cam_t,
                    float* focal, float focal_length, float img_size, int B, hipStream_t s);
int launch_rot6d(const float* x, float* R, int n, hipStream_t s);
int launch_cam_t(const float* cam, float* cam_t, float focal_length, float img_size, int B, hipStream_t s);
int launch_vq_argmin_rows(const float* x, const float* dot, const float* cnorm, int32_t* idx, float* dist, int rows,
                          hipStream_t s);
int launch_code_norm(const float* cb, float* cn, int ncode, hipStream_t s);
// lbs.hip
int launch_lbs_jreg(const float* Jreg, const float* vt, const float* sd, float* Jt, float* Jsd, hipStream_t s);
int launch_lbs(const float* rotmat, const float* betas, const float* cam_t, const float* Jt, const float* Jsd,
               const int32_t* parents, const float* vt, const float* sd, const float* pd, const float* W,
               const float* J19, const int32_t* extra, const int32_t* jmap, float* A, float* pf, float* Jtr, float* verts,
               float* joints, float* kp2d, float focal_over_size, int B, hipStream_t s);
int launch_rodrigues(const float* aa, float* R, int n, hipStream_t s);
// eval.hip
int launch_eval_pose(const float* pred, const float* gt, int nj, int gt_stride, const int32_t* kp, int nkp, int pelvis_ind,
                     int pelvis_mode, float* mpjpe, float* re, float* pelv, int B, hipStream_t s);
int launch_eval_pve(const float* pv, const float* gv, const float* pelv, int nv, float* pve, int B, hipStream_t s);
int launch_regress_joints(const float* J, const float* verts, int nj, int nv, float* out, int B, hipStream_t s);
