// On-GPU evaluation metrics for the step right after the hot path:
// MPJPE, PA-MPJPE (batched Procrustes with a 3x3 SVD per crop) and PVE, in millimetres.
//
// Replaces tokenhmr/lib/utils/pose_utils.py:61-114 compute_similarity_transform, :116-127
// reconstruction_error, :129-143 eval_pose and the per-batch arithmetic of Evaluator.__call__ (:201-275),
// which in the reference forces a D2H copy of (B,6890,3) vertices per batch; here only 3 floats per crop
// leave the GPU.
#include "common.h"

namespace {

// ---- 3x3 SVD by one-sided Jacobi in fp64 (per crop, one thread) ----
// K = U diag(s) V^T.  Returns R = V Z U^T with Z = diag(1,1,sign(det(U V^T))) applied to the SMALLEST singular
// value (torch.svd orders singular values descending, pose_utils.py:95-103), and trace(R K) = s0 + s1 + z*s2.
__device__ void procrustes_rotation(const double Kin[3][3], double R[3][3], double& trace_rk) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = Kin[i][j];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                off += fabs(gamma);
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-300) break;
    }
    // columns of A are s_i * u_i
    double sv[3], U[3][3];
    for (int j = 0; j < 3; ++j) {
        sv[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
        const double inv = sv[j] > 0 ? 1.0 / sv[j] : 0.0;
        for (int i = 0; i < 3; ++i) U[i][j] = A[i][j] * inv;
    }
    int smallest = 0;
    if (sv[1] < sv[smallest]) smallest = 1;
    if (sv[2] < sv[smallest]) smallest = 2;
    // a zero singular value leaves u undefined: complete it to a right-handed frame so det(U) is defined
    if (sv[smallest] == 0.0) {
        const int a = (smallest + 1) % 3, b = (smallest + 2) % 3;
        U[0][smallest] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][smallest] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][smallest] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
    }
    auto det3 = [](const double M[3][3]) {
        return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
               M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    };
    const double dd = det3(U) * det3(V);
    const double z = dd > 0 ? 1.0 : (dd < 0 ? -1.0 : 0.0);       // torch.sign
    trace_rk = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double zk = (k == smallest) ? z : 1.0;
        trace_rk += zk * sv[k];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] += zk * V[i][k] * U[j][k];   // R = V Z U^T
    }
}

// One 64-thread block per crop.  pred (B,nj,3); gt (B,nj,gt_stride) (keypoints_3d carries a confidence column,
// the reference slices [:, :, :-1], pose_utils.py:225).  pelvis_mode 0: joint `pelvis_ind`; 1: (j1 + j2)/2 (EMDB).
__global__ __launch_bounds__(64) void eval_pose_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      int nj, int gt_stride, const int32_t* __restrict__ kp, int nkp,
                                                      int pelvis_ind, int pelvis_mode, float* __restrict__ mpjpe,
                                                      float* __restrict__ re, float* __restrict__ pelv_out) {
    __shared__ float P[64][3], Gt[64][3];
    __shared__ float pelv[2][3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* pb = pred + (int64_t)b * nj * 3;
    const float* gb = gt + (int64_t)b * nj * gt_stride;
    if (t < 3) {
        if (pelvis_mode == 0) {
            pelv[0][t] = pb[pelvis_ind * 3 + t];
            pelv[1][t] = gb[pelvis_ind * gt_stride + t];
        } else {
            pelv[0][t] = (pb[1 * 3 + t] + pb[2 * 3 + t]) / 2.0f;
            pelv[1][t] = (gb[1 * gt_stride + t] + gb[2 * gt_stride + t]) / 2.0f;
        }
    }
    __syncthreads();
    if (pelv_out && t < 6) pelv_out[b * 6 + t] = pelv[t / 3][t % 3];
    float err = 0.f;
    if (t < nkp) {
        int j = kp[t];
        // an index outside [0, nj) never reads out of bounds: the crop's metrics become NaN instead (the host binding
        // rejects such lists up front, like the reference's IndexError)
        const bool bad = j < 0 || j >= nj;
        if (bad) j = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            P[t][i] = pb[j * 3 + i] - pelv[0][i];
            Gt[t][i] = gb[j * gt_stride + i] - pelv[1][i];
        }
        const float dx = P[t][0] - Gt[t][0], dy = P[t][1] - Gt[t][1], dz = P[t][2] - Gt[t][2];
        err = bad ? __builtin_nanf("") : sqrtf(dx * dx + dy * dy + dz * dz);
        if (bad) P[t][0] = __builtin_nanf("");
    }
    const float s = wave_sum(err);
    if (t == 0) mpjpe[b] = 1000.0f * (s / (float)nkp);
    __syncthreads();
    // Procrustes (pose_utils.py:76-112) by thread 0 in fp64: N <= 64 points
    if (t == 0) {
        double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
        for (int k = 0; k < nkp; ++k)
            for (int i = 0; i < 3; ++i) { mu1[i] += P[k][i]; mu2[i] += Gt[k][i]; }
        for (int i = 0; i < 3; ++i) { mu1[i] /= nkp; mu2[i] /= nkp; }
        double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
        for (int k = 0; k < nkp; ++k) {
            double x1[3], x2[3];
            for (int i = 0; i < 3; ++i) { x1[i] = P[k][i] - mu1[i]; x2[i] = Gt[k][i] - mu2[i]; var1 += x1[i] * x1[i]; }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) K[i][j] += x1[i] * x2[j];      // K = X1 X2^T
        }
        double R[3][3], tr;
        procrustes_rotation(K, R, tr);
        const double scale = tr / var1;
        double tvec[3];
        for (int i = 0; i < 3; ++i) tvec[i] = mu2[i] - scale * (R[i][0] * mu1[0] + R[i][1] * mu1[1] + R[i][2] * mu1[2]);
        double acc = 0.0;
        for (int k = 0; k < nkp; ++k) {
            double d2 = 0.0;
            for (int i = 0; i < 3; ++i) {
                const double h = scale * (R[i][0] * P[k][0] + R[i][1] * P[k][1] + R[i][2] * P[k][2]) + tvec[i];
                const double d = h - Gt[k][i];
                d2 += d * d;
            }
            acc += sqrt(d2);
        }
        re[b] = (float)(1000.0 * acc / nkp);
    }
}

// PVE: mean over vertices of || (pv - pred_pelvis) - (gv - gt_pelvis) || * 1000  (pose_utils.py:239-247)
__global__ __launch_bounds__(256) void eval_pve_kernel(const float* __restrict__ pv, const float* __restrict__ gv,
                                                       const float* __restrict__ pelv, int nv, float* __restrict__ pve) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = pv + (int64_t)b * nv * 3;
    const float* g = gv + (int64_t)b * nv * 3;
    const float px = pelv[b * 6 + 0], py = pelv[b * 6 + 1], pz = pelv[b * 6 + 2];
    const float gx = pelv[b * 6 + 3], gy = pelv[b * 6 + 4], gz = pelv[b * 6 + 5];
    float acc = 0.f;
    for (int v = t; v < nv; v += 256) {
        const float dx = (p[v * 3 + 0] - px) - (g[v * 3 + 0] - gx);
        const float dy = (p[v * 3 + 1] - py) - (g[v * 3 + 1] - gy);
        const float dz = (p[v * 3 + 2] - pz) - (g[v * 3 + 2] - gz);
        acc += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    const float s = wave_sum(acc);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) pve[b] = 1000.0f * ((((red[0] + red[1]) + red[2]) + red[3]) / (float)nv);
}

// joints = J (nj, nv) @ verts (B, nv, 3): the EMDB branch's J_regressor_24_SMPL (pose_utils.py:212,219)
__global__ __launch_bounds__(256) void regress_joints_kernel(const float* __restrict__ J, const float* __restrict__ verts,
                                                             int nj, int nv, float* __restrict__ out) {
    __shared__ float part[4][3];
    const int b = blockIdx.x, j = blockIdx.y, t = threadIdx.x;
    const float* vb = verts + (int64_t)b * nv * 3;
    const float* w = J + (int64_t)j * nv;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int v = t; v < nv; v += 256) {
        const float ww = w[v];
        a0 = fmaf(ww, vb[v * 3 + 0], a0);
        a1 = fmaf(ww, vb[v * 3 + 1], a1);
        a2 = fmaf(ww, vb[v * 3 + 2], a2);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if ((t & 63) == 0) { part[t >> 6][0] = a0; part[t >> 6][1] = a1; part[t >> 6][2] = a2; }
    __syncthreads();
    if (t < 3) out[((int64_t)b * nj + j) * 3 + t] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
}

}  // namespace

int launch_eval_pose(const float* pred, const float* gt, int nj, int gt_stride, const int32_t* kp, int nkp, int pelvis_ind,
                     int pelvis_mode, float* mpjpe, float* re, float* pelv, int B, hipStream_t s) {
    if (nkp < 1 || nkp > 64 || B < 1) return -1;
    hipLaunchKernelGGL(eval_pose_kernel, dim3(B), dim3(64), 0, s, pred, gt, nj, gt_stride, kp, nkp, pelvis_ind, pelvis_mode,
                       mpjpe, re, pelv);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_eval_pve(const float* pv, const float* gv, const float* pelv, int nv, float* pve, int B, hipStream_t s) {
    hipLaunchKernelGGL(eval_pve_kernel, dim3(B), dim3(256), 0, s, pv, gv, pelv, nv, pve);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_regress_joints(const float* J, const float* verts, int nj, int nv, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(regress_joints_kernel, dim3(B, nj), dim3(256), 0, s, J, verts, nj, nv, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
