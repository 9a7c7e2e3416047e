// SMPL forward as used by TokenHMR: linear blend skinning over 6890 vertices from rotation matrices + betas,
// 44 output joints and their weak-perspective projection.
//
// Replaces tokenhmr/lib/models/smpl_wrapper.py:27-41 (SMPL.forward: joint_map remap + J19 regressor) over the
// un-vendored smplx==0.1.28 `SMPLLayer.forward(pose2rot=False)` -> `lbs.lbs` (restated from its published
// algorithm), and tokenhmr/lib/utils/geometry.py:86-124 perspective_projection as called
// at tokenhmr/lib/models/tokenhmr.py:183-187.
//
// HBM-bound stage (83 KB written per crop; 19.8 MB of constants).  Layout decisions:
//   * J = J_regressor . v_shaped is linear in betas, so J_template (24x3) and J_shapedirs (24x3x10) are
//     precomputed once in fp64 at load time: no per-crop reduction over 6890 vertices before the chain.
//   * skin kernel: a block owns 32 vertices (a 207 x 96 slice of posedirs, staged ONCE in LDS) and loops over
//     32 crops, so the 17 MB posedirs stream is read from HBM once per 32 crops instead of once per crop;
//     pose features, betas and the 24 bone matrices of those crops sit in LDS too (broadcast reads).
//   * every global access is coalesced: consecutive lanes = consecutive vertices (12 B each) on stores.
#include "common.h"

namespace {

constexpr int NV = 6890, NJ = 24, NB = 10, NP = 207;
constexpr int VCH = 32;      // vertices per block
constexpr int CG = 32;       // crops per block (4 per thread)

// ---- one-time: J_template[j][i], J_shapedirs[j][i][l] in fp64 -> fp32 ----
__global__ __launch_bounds__(256) void lbs_jreg_kernel(const float* __restrict__ Jreg, const float* __restrict__ vt,
                                                       const float* __restrict__ sd, float* __restrict__ Jt,
                                                       float* __restrict__ Jsd) {
    __shared__ double red[256];
    const int j = blockIdx.x, q = blockIdx.y;   // q in [0,33): 0..2 template coords, 3.. = 3 + i*10 + l
    double acc = 0.0;
    for (int v = threadIdx.x; v < NV; v += 256) {
        const double w = Jreg[(int64_t)j * NV + v];
        const double val = (q < 3) ? (double)vt[v * 3 + q] : (double)sd[(int64_t)v * 30 + (q - 3)];
        acc += w * val;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (q < 3) Jt[j * 3 + q] = (float)red[0];
        else Jsd[j * 30 + (q - 3)] = (float)red[0];
    }
}

// ---- per crop: joints, kinematic chain, bone matrices A, pose feature ----
__global__ __launch_bounds__(128) void lbs_prep_kernel(const float* __restrict__ rotmat, const float* __restrict__ betas,
                                                       const float* __restrict__ Jt, const float* __restrict__ Jsd,
                                                       const int32_t* __restrict__ parents, float* __restrict__ A,
                                                       float* __restrict__ pf, float* __restrict__ Jtr) {
    __shared__ float J[NJ][3];
    __shared__ float G[NJ][12];
    __shared__ float R[NJ][9];
    __shared__ float bs[NB];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < NJ * 9; i += 128) R[i / 9][i % 9] = rotmat[(int64_t)b * NJ * 9 + i];
    if (t < NB) bs[t] = betas[(int64_t)b * NB + t];
    __syncthreads();
    if (t < NJ * 3) {
        float v = 0.f;
#pragma unroll
        for (int l = 0; l < NB; ++l) v = fmaf(bs[l], Jsd[t * NB + l], v);
        J[t / 3][t % 3] = Jt[t] + v;
    }
    // pose_feature = (R[1:] - I).view(207)   (smplx lbs.py: pose_feature)
    for (int i = t; i < NP; i += 128) {
        const int j = 1 + i / 9, e = i % 9;
        pf[(int64_t)b * NP + i] = R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    __syncthreads();
    // kinematic chain (smplx batch_rigid_transform): G_0 = T_0, G_i = G_parent(i) . T_i,  T_i = [R_i | J_i - J_parent]
    const int r = t / 4, c = t % 4;
    if (t < 12) G[0][t] = (c < 3) ? R[0][r * 3 + c] : J[0][r];
    __syncthreads();
    for (int i = 1; i < NJ; ++i) {
        const int p = parents[i];
        if (t < 12) {
            float v;
            if (c < 3) {
                v = G[p][r * 4 + 0] * R[i][0 * 3 + c] + G[p][r * 4 + 1] * R[i][1 * 3 + c] + G[p][r * 4 + 2] * R[i][2 * 3 + c];
            } else {
                const float rx = J[i][0] - J[p][0], ry = J[i][1] - J[p][1], rz = J[i][2] - J[p][2];
                v = G[p][r * 4 + 0] * rx + G[p][r * 4 + 1] * ry + G[p][r * 4 + 2] * rz + G[p][r * 4 + 3];
            }
            G[i][t] = v;
        }
        __syncthreads();
    }
    // A_i = G_i with the rest-pose joint removed: A[:, :3, 3] = G[:, :3, 3] - G[:, :3, :3] . J_i
    for (int i = t; i < NJ * 12; i += 128) {
        const int j = i / 12, e = i % 12, rr = e / 4, cc = e % 4;
        float v = G[j][e];
        if (cc == 3) v = v - (G[j][rr * 4 + 0] * J[j][0] + G[j][rr * 4 + 1] * J[j][1] + G[j][rr * 4 + 2] * J[j][2]);
        A[(int64_t)b * NJ * 12 + i] = v;
    }
    if (t < NJ * 3) Jtr[(int64_t)b * NJ * 3 + t] = G[t / 3][(t % 3) * 4 + 3];
}

// ---- skinning: blend shapes + pose correctives + weighted bone transform, 32 vertices x 32 crops per block ----
__global__ __launch_bounds__(256, 1) void lbs_skin_kernel(const float* __restrict__ vt, const float* __restrict__ sd,
                                                          const float* __restrict__ pd, const float* __restrict__ W,
                                                          const float* __restrict__ A, const float* __restrict__ pf,
                                                          const float* __restrict__ betas, float* __restrict__ verts, int B) {
    __shared__ __attribute__((aligned(16))) float pdS[NP * VCH * 3];      // 79,488 B
    __shared__ __attribute__((aligned(16))) float pfS[CG * NP];            // 26,496 B
    __shared__ __attribute__((aligned(16))) float AS[CG * NJ * 12];        // 36,864 B
    __shared__ float bS[CG * NB];
    const int tid = threadIdx.x;
    const int v0 = blockIdx.x * VCH, c0 = blockIdx.y * CG;
    const int nc = min(CG, B - c0);
    const int ncols = min(VCH, NV - v0) * 3;

    for (int i = tid; i < NP * VCH * 3; i += 256) {
        const int k = i / (VCH * 3), col = i % (VCH * 3);
        pdS[i] = (col < ncols) ? pd[(int64_t)k * (NV * 3) + v0 * 3 + col] : 0.f;
    }
    for (int i = tid; i < CG * NP; i += 256) pfS[i] = (i < nc * NP) ? pf[(int64_t)c0 * NP + i] : 0.f;
    for (int i = tid; i < CG * NJ * 12; i += 256) AS[i] = (i < nc * NJ * 12) ? A[(int64_t)c0 * NJ * 12 + i] : 0.f;
    for (int i = tid; i < CG * NB; i += 256) bS[i] = (i < nc * NB) ? betas[(int64_t)c0 * NB + i] : 0.f;
    __syncthreads();

    const int vl = tid & 31, slot = tid >> 5;      // crops slot, slot+8, slot+16, slot+24
    const int v = v0 + vl;
    const bool vok = v < NV;
    const int vv = vok ? v : NV - 1;

    // pose correctives: off[c][i] = sum_k pf[c][k] * posedirs[k][3v+i]
    float off[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) off[c][0] = off[c][1] = off[c][2] = 0.f;
    const float* pcol = pdS + vl * 3;
    for (int k = 0; k < NP; ++k) {
        const float p0 = pcol[k * VCH * 3 + 0], p1 = pcol[k * VCH * 3 + 1], p2 = pcol[k * VCH * 3 + 2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float f = pfS[(slot + 8 * c) * NP + k];
            off[c][0] = fmaf(f, p0, off[c][0]);
            off[c][1] = fmaf(f, p1, off[c][1]);
            off[c][2] = fmaf(f, p2, off[c][2]);
        }
    }
    // shape blend: v_shaped = v_template + shapedirs . betas
    float sdv[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) sdv[i] = sd[(int64_t)vv * 30 + i];
    const float t0 = vt[vv * 3 + 0], t1 = vt[vv * 3 + 1], t2 = vt[vv * 3 + 2];
    float wv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wv[j] = W[(int64_t)vv * NJ + j];

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cl = slot + 8 * c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const float bt = bS[cl * NB + l];
            s0 = fmaf(bt, sdv[0 * NB + l], s0);
            s1 = fmaf(bt, sdv[1 * NB + l], s1);
            s2 = fmaf(bt, sdv[2 * NB + l], s2);
        }
        const float x = (t0 + s0) + off[c][0], y = (t1 + s1) + off[c][1], z = (t2 + s2) + off[c][2];
        // T = sum_j W[v][j] * A[c][j]  (3x4)
        f32x4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
        const f32x4* Ac = reinterpret_cast<const f32x4*>(AS + cl * NJ * 12);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float w = wv[j];
            T0 += w * Ac[j * 3 + 0];
            T1 += w * Ac[j * 3 + 1];
            T2 += w * Ac[j * 3 + 2];
        }
        const float ox = T0[0] * x + T0[1] * y + T0[2] * z + T0[3];
        const float oy = T1[0] * x + T1[1] * y + T1[2] * z + T1[3];
        const float oz = T2[0] * x + T2[1] * y + T2[2] * z + T2[3];
        if (vok && cl < nc) {
            float* o = verts + ((int64_t)(c0 + cl) * NV + v) * 3;
            o[0] = ox; o[1] = oy; o[2] = oz;
        }
    }
}

// ---- joints: 24 chain joints + 21 vertex picks -> joint_map(25) ++ J19 regressor(19) = 44, + projection ----
__global__ __launch_bounds__(256) void lbs_joints_kernel(const float* __restrict__ verts, const float* __restrict__ Jtr,
                                                         const float* __restrict__ J19, const int32_t* __restrict__ extra,
                                                         const int32_t* __restrict__ jmap, const float* __restrict__ cam_t,
                                                         float* __restrict__ joints, float* __restrict__ kp2d,
                                                         float focal_over_size) {
    __shared__ float part[4][57];
    __shared__ float jo[44][3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vb = verts + (int64_t)b * NV * 3;
    float acc[19][3];
#pragma unroll
    for (int j = 0; j < 19; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    for (int v = tid; v < NV; v += 256) {
        const float x = vb[v * 3 + 0], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
#pragma unroll
        for (int j = 0; j < 19; ++j) {
            const float w = J19[(int64_t)j * NV + v];
            acc[j][0] = fmaf(w, x, acc[j][0]);
            acc[j][1] = fmaf(w, y, acc[j][1]);
            acc[j][2] = fmaf(w, z, acc[j][2]);
        }
    }
#pragma unroll
    for (int j = 0; j < 19; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float s = wave_sum(acc[j][i]);
            if (lane == 0) part[wave][j * 3 + i] = s;
        }
    __syncthreads();
    if (tid < 57) jo[25 + tid / 3][tid % 3] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    if (tid >= 64 && tid < 64 + 75) {
        const int t = tid - 64, j = t / 3, i = t % 3;
        const int src = jmap[j];
        jo[j][i] = (src < NJ) ? Jtr[((int64_t)b * NJ + src) * 3 + i] : vb[extra[src - NJ] * 3 + i];
    }
    __syncthreads();
    if (tid < 132 && joints) joints[(int64_t)b * 132 + tid] = jo[tid / 3][tid % 3];
    if (tid < 44 && kp2d && cam_t) {
        const float px = jo[tid][0] + cam_t[b * 3 + 0], py = jo[tid][1] + cam_t[b * 3 + 1], pz = jo[tid][2] + cam_t[b * 3 + 2];
        kp2d[((int64_t)b * 44 + tid) * 2 + 0] = (px / pz) * focal_over_size;
        kp2d[((int64_t)b * 44 + tid) * 2 + 1] = (py / pz) * focal_over_size;
    }
}

// smplx.lbs.batch_rodrigues (smplx==0.1.28, pose2rot=True path used for GT meshes, image_dataset.py:254-270):
//   angle = ||r + 1e-8||, dir = r / angle, R = I + sin(angle) K + (1 - cos(angle)) K^2,  K = [dir]_x
__global__ void rodrigues_kernel(const float* __restrict__ aa, float* __restrict__ R, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = aa[i * 3 + 0], y = aa[i * 3 + 1], z = aa[i * 3 + 2];
    const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = x / angle, ry = y / angle, rz = z / angle;
    const float s = sinf(angle), c = cosf(angle), oc = 1.0f - c;
    // K^2 = dir dir^T - I (|dir| = 1 up to the epsilon), written out as smplx's bmm(K, K)
    const float k2[9] = {-(rz * rz) - ry * ry, rx * ry, rx * rz,
                         rx * ry, -(rz * rz) - rx * rx, ry * rz,
                         rx * rz, ry * rz, -(ry * ry) - rx * rx};
    const float k1[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float* o = R + (int64_t)i * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) o[e] = ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f) + s * k1[e] + oc * k2[e];
}

}  // namespace

int launch_rodrigues(const float* aa, float* R, int n, hipStream_t s) {
    hipLaunchKernelGGL(rodrigues_kernel, dim3((n + 255) / 256), dim3(256), 0, s, aa, R, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_lbs_jreg(const float* Jreg, const float* vt, const float* sd, float* Jt, float* Jsd, hipStream_t s) {
    hipLaunchKernelGGL(lbs_jreg_kernel, dim3(NJ, 33), dim3(256), 0, s, Jreg, vt, sd, Jt, Jsd);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_lbs(const float* rotmat, const float* betas, const float* cam_t, const float* Jt, const float* Jsd,
               const int32_t* parents, const float* vt, const float* sd, const float* pd, const float* W,
               const float* J19, const int32_t* extra, const int32_t* jmap, float* A, float* pf, float* Jtr, float* verts,
               float* joints, float* kp2d, float focal_over_size, int B, hipStream_t s) {
    hipLaunchKernelGGL(lbs_prep_kernel, dim3(B), dim3(128), 0, s, rotmat, betas, Jt, Jsd, parents, A, pf, Jtr);
    hipLaunchKernelGGL(lbs_skin_kernel, dim3((NV + VCH - 1) / VCH, (B + CG - 1) / CG), dim3(256), 0, s, vt, sd, pd, W, A, pf,
                       betas, verts, B);
    hipLaunchKernelGGL(lbs_joints_kernel, dim3(B), dim3(256), 0, s, verts, Jtr, J19, extra, jmap, cam_t, joints, kp2d,
                       focal_over_size);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
