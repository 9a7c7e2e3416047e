// Small kernels of the SMPL token head: 1-token decoder glue, cross-attention over the 192 image tokens,
// read-out assembly + 6D->rotmat + camera, and the argmin-L2 row kernel of the VQ quantiser.
#include "common.h"

namespace {

// pose_transformer.py:350-354 with the zero input token of token_head.py:91:
// Linear(1->1024)(0) == bias exactly, then += pos_embedding[:, :1]  ->  x[b][:] = bias + pos  for every crop.
__global__ void decoder_init_kernel(const float* __restrict__ bias, const float* __restrict__ pos, float* __restrict__ x,
                                    int B, int E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * E) {
        const int c = i % E;
        x[i] = bias[c] + pos[c];
    }
}

// CrossAttention.forward, pose_transformer.py:111-124, for ONE query token: per (crop, head)
//   dots[j] = (q_h . k_h[j]) * 64^-0.5 ; softmax over 192 keys ; out_h = sum_j a[j] v_h[j].
// q (B,512); kv rows (b*192 + j) of a (B*192, ldkv) matrix, K at column koff + h*64, V at koff + 512 + h*64
// (to_kv(context).chunk(2), :113; context is NOT normalised, PreNorm only normalises x, :33-37).
__global__ __launch_bounds__(256) void cross_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                         int64_t ldkv, int koff, float* __restrict__ out) {
    __shared__ float qs[64];
    __shared__ float p[192];
    __shared__ float red[4];
    __shared__ float part[4][64];
    const int b = blockIdx.x >> 3, h = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) qs[tid] = q[(int64_t)b * 512 + h * 64 + tid];
    __syncthreads();
    const float* kbase = kv + (int64_t)b * 192 * ldkv + koff + h * 64;
    float d = -INFINITY;
    if (tid < 192) {
        const float* kr = kbase + (int64_t)tid * ldkv;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(qs[c * 4 + e], kk[e], acc);
        }
        d = acc * 0.125f;   // scale applied after the dot (pose_transformer.py:117)
    }
    float m = wave_max(d);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float e = (tid < 192) ? expf(d - m) : 0.f;
    float sum = wave_sum(e);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = ((red[0] + red[1]) + red[2]) + red[3];
    if (tid < 192) p[tid] = e / sum;
    __syncthreads();
    // out[dd] = sum_j p[j] * V[j][dd]: 4 waves x 48 keys, lanes over the 64 dims (coalesced 256 B rows)
    const float* vbase = kbase + 512;
    float acc = 0.f;
    for (int j = wave * 48; j < wave * 48 + 48; ++j) acc = fmaf(p[j], vbase[(int64_t)j * ldkv + lane], acc);
    part[wave][lane] = acc;
    __syncthreads();
    if (tid < 64) out[(int64_t)b * 512 + h * 64 + tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
}

// token_head.py:99-105,123 + tokenhmr.py:165-169: assemble pose6d/betas/cam from the fused read-out GEMM
// (ro columns: grot 0..5 | shape 6..15 | cam 16..18 | hands 19..30) and the VQ-decoded body pose, add the mean
// parameters, convert 6D -> rotation matrices (geometry.py:64-84) and compute the camera translation.
__global__ __launch_bounds__(64) void assemble_kernel(const float* __restrict__ ro, int ldro, const float* __restrict__ bpose,
                                                     const float* __restrict__ init_pose, const float* __restrict__ init_betas,
                                                     const float* __restrict__ init_cam, float* __restrict__ pose6d,
                                                     float* __restrict__ rotmat, float* __restrict__ betas,
                                                     float* __restrict__ cam, float* __restrict__ cam_t,
                                                     float* __restrict__ focal, float focal_length, float img_size) {
    __shared__ float p6[144];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* r = ro + (int64_t)b * ldro;
    for (int i = t; i < 144; i += 64) {
        float v;
        if (i < 6) v = r[i];
        else if (i < 132) v = bpose[(int64_t)b * 126 + (i - 6)];
        else v = r[19 + (i - 132)];
        v += init_pose[i];
        p6[i] = v;
        if (pose6d) pose6d[(int64_t)b * 144 + i] = v;
    }
    __syncthreads();
    if (t < 24) {
        const float a1x = p6[t * 6 + 0], a1y = p6[t * 6 + 1], a1z = p6[t * 6 + 2];
        const float a2x = p6[t * 6 + 3], a2y = p6[t * 6 + 4], a2z = p6[t * 6 + 5];
        const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);   // F.normalize eps
        const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
        const float dp = b1x * a2x + b1y * a2y + b1z * a2z;
        const float ux = a2x - dp * b1x, uy = a2y - dp * b1y, uz = a2z - dp * b1z;
        const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
        const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
        const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
        float* R = rotmat + ((int64_t)b * 24 + t) * 9;
        R[0] = b1x; R[1] = b1y; R[2] = b1z;
        R[3] = b2x; R[4] = b2y; R[5] = b2z;
        R[6] = b3x; R[7] = b3y; R[8] = b3z;
    }
    if (t >= 32 && t < 42) betas[(int64_t)b * 10 + (t - 32)] = r[6 + (t - 32)] + init_betas[t - 32];
    if (t == 63) {
        const float c0 = r[16] + init_cam[0], c1 = r[17] + init_cam[1], c2 = r[18] + init_cam[2];
        cam[b * 3 + 0] = c0; cam[b * 3 + 1] = c1; cam[b * 3 + 2] = c2;
        if (cam_t) {
            cam_t[b * 3 + 0] = c1;
            cam_t[b * 3 + 1] = c2;
            cam_t[b * 3 + 2] = (2.0f * focal_length) / (img_size * c0 + 1e-9f);
        }
        if (focal) { focal[b * 2 + 0] = focal_length; focal[b * 2 + 1] = focal_length; }
    }
}

// tokenhmr.py:165-169: pred_cam_t = [cam1, cam2, 2*f / (IMAGE_SIZE*cam0 + 1e-9)]
__global__ void cam_t_kernel(const float* __restrict__ cam, float* __restrict__ cam_t, float focal_length, float img_size, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float c0 = cam[b * 3 + 0], c1 = cam[b * 3 + 1], c2 = cam[b * 3 + 2];
    cam_t[b * 3 + 0] = c1;
    cam_t[b * 3 + 1] = c2;
    cam_t[b * 3 + 2] = (2.0f * focal_length) / (img_size * c0 + 1e-9f);
}

// standalone rot6d_to_rotmat (geometry.py:64-84): (n,6) -> (n,3,3)
__global__ void rot6d_kernel(const float* __restrict__ x, float* __restrict__ Rm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = x + (int64_t)i * 6;
    const float a1x = p[0], a1y = p[1], a1z = p[2], a2x = p[3], a2y = p[4], a2z = p[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dp = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - dp * b1x, uy = a2y - dp * b1y, uz = a2z - dp * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    float* R = Rm + (int64_t)i * 9;
    R[0] = b1x; R[1] = b1y; R[2] = b1z;
    R[3] = b2x; R[4] = b2y; R[5] = b2z;
    R[6] = b1y * b2z - b1z * b2y; R[7] = b1z * b2x - b1x * b2z; R[8] = b1x * b2y - b1y * b2x;
}

// aa_to_rotmat (tokenhmr/lib/utils/geometry.py:5-44): axis-angle -> quaternion -> rotation matrix, operation by operation
// (angle = ||theta + 1e-8||, axis = theta / angle, half-angle quaternion, re-normalised, nine quadratic forms)
__global__ void aa_to_rotmat_kernel(const float* __restrict__ aa, float* __restrict__ Rm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float tx = aa[i * 3 + 0], ty = aa[i * 3 + 1], tz = aa[i * 3 + 2];
    const float ex = tx + 1e-8f, ey = ty + 1e-8f, ez = tz + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float nx = tx / angle, ny = ty / angle, nz = tz / angle;
    const float half = angle * 0.5f;
    const float qw = cosf(half), sn = sinf(half);
    const float qx = sn * nx, qy = sn * ny, qz = sn * nz;
    const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    const float w = qw / qn, x = qx / qn, y = qy / qn, z = qz / qn;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    float* R = Rm + (int64_t)i * 9;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;    R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;    R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
}

// QuantizeEMAReset.quantize (tokenization/models/quantize_cnn.py:80-86), second half: given dot = x.C^T (MFMA GEMM),
// dist[k] = (sum(x^2) - 2*dot[k]) + sum(C_k^2), argmin with lowest-index tie-break; wavefront min-reduction.
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ x, const float* __restrict__ dot,
                                                        const float* __restrict__ cnorm, int32_t* __restrict__ idx,
                                                        float* __restrict__ dist_out, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)row * 256 + lane * 4);
    const float xn = wave_sum((xv[0] * xv[0] + xv[1] * xv[1]) + (xv[2] * xv[2] + xv[3] * xv[3]));
    const float* dr = dot + (int64_t)row * 2048;
    float best = INFINITY;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k0 = (i * 64 + lane) * 4;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dr + k0);
        const f32x4 cn = *reinterpret_cast<const f32x4*>(cnorm + k0);
        f32x4 ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ds[e] = (xn - 2.0f * dv[e]) + cn[e];
            if (ds[e] < best) { best = ds[e]; bi = k0 + e; }
        }
        if (dist_out) *reinterpret_cast<f32x4*>(dist_out + (int64_t)row * 2048 + k0) = ds;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) idx[row] = bi;
}

// ||C_k||^2 for the 2048 codes (quantize_cnn.py:83 torch.sum(k_w ** 2, dim=0)); one wave per code
__global__ __launch_bounds__(256) void code_norm_kernel(const float* __restrict__ cb, float* __restrict__ cn, int ncode) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ncode) return;
    const int lane = threadIdx.x & 63;
    const f32x4 v = *reinterpret_cast<const f32x4*>(cb + (int64_t)row * 256 + lane * 4);
    const float s = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
    if (lane == 0) cn[row] = s;
}

}  // namespace

int launch_decoder_init(const float* bias, const float* pos, float* x, int B, int E, hipStream_t s) {
    hipLaunchKernelGGL(decoder_init_kernel, dim3((B * E + 255) / 256), dim3(256), 0, s, bias, pos, x, B, E);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_cross_attn(const float* q, const float* kv, int64_t ldkv, int koff, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(cross_attn_kernel, dim3(B * 8), dim3(256), 0, s, q, kv, ldkv, koff, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_assemble(const float* ro, int ldro, const float* bpose, const float* init_pose, const float* init_betas,
                    const float* init_cam, float* pose6d, float* rotmat, float* betas, float* cam, float* cam_t,
                    float* focal, float focal_length, float img_size, int B, hipStream_t s) {
    hipLaunchKernelGGL(assemble_kernel, dim3(B), dim3(64), 0, s, ro, ldro, bpose, init_pose, init_betas, init_cam, pose6d,
                       rotmat, betas, cam, cam_t, focal, focal_length, img_size);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_cam_t(const float* cam, float* cam_t, float focal_length, float img_size, int B, hipStream_t s) {
    hipLaunchKernelGGL(cam_t_kernel, dim3((B + 255) / 256), dim3(256), 0, s, cam, cam_t, focal_length, img_size, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_aa_to_rotmat(const float* aa, float* R, int n, hipStream_t s) {
    hipLaunchKernelGGL(aa_to_rotmat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, aa, R, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_rot6d(const float* x, float* R, int n, hipStream_t s) {
    hipLaunchKernelGGL(rot6d_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, R, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_vq_argmin_rows(const float* x, const float* dot, const float* cnorm, int32_t* idx, float* dist, int rows,
                          hipStream_t s) {
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, dot, cnorm, idx, dist, rows);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_code_norm(const float* cb, float* cn, int ncode, hipStream_t s) {
    hipLaunchKernelGGL(code_norm_kernel, dim3((ncode + 3) / 4), dim3(256), 0, s, cb, cn, ncode);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
