// SMPL forward as used by TokenHMR: linear blend skinning over 6890 vertices from rotation matrices + betas,
// 44 output joints and their weak-perspective projection.
//
// Replaces tokenhmr/lib/models/smpl_wrapper.py:27-41 (SMPL.forward: joint_map remap + J19 regressor) over the
// un-vendored smplx==0.1.28 `SMPLLayer.forward(pose2rot=False)` -> `lbs.lbs` (restated from its published
// algorithm), and tokenhmr/lib/utils/geometry.py:86-124 perspective_projection as called
// at tokenhmr/lib/models/tokenhmr.py:183-187.
//
// SURVEY classes this stage as HBM-bound (83 KB written per crop; 19.8 MB of constants per batch).  Measured it is NOT: 25 MB in 55 us at 64
// crops = 0.057 of the HBM peak — three dependent launches (9 + 13 + 29 us) whose time is per-workgroup set-up, the 72 broadcast LDS reads of
// bone matrices per thread and crop, and the serial joint finish (five restructurings measured and not kept, HISTORY.md 10.4,
// 11.8).  Layout decisions:
//   * J = J_regressor . v_shaped is linear in betas, so J_template (24x3) and J_shapedirs (24x3x10) are
//     precomputed once in fp64 at load time: no per-crop reduction over 6890 vertices before the chain.
//   * blend shapes + pose correctives are ONE matrix product: v_posed (B x 20670) = [betas | pose_feature] (B x 217, zero-
//     padded to 224) . dirs^T + v_template, with dirs^T (20670 x 224, K-contiguous) built once at load from shapedirs and
//     posedirs.  It runs on the MFMA GEMM (gemm_f32.hip) with the template as the bias epilogue, so the 18.5 MB dirs
//     stream is read once per 64-row tile of crops at matrix-core speed (the first version did these 6890*621 FMAs per
//     crop on the VALU out of LDS and took 107 us at B = 64).
//   * skin + joints kernel: one thread per vertex, 8 crops per workgroup pass: 24 weights (registers) x 24 bone matrices (scalar
//     loads) -> 3x4 transform; the crop group's last workgroup finishes its joints (three launches per call: prep, blend GEMM, skin + joints).
//   * every global access is coalesced: consecutive lanes = consecutive vertices (12 B each) on loads and stores.
#include "common.h"

namespace {

constexpr int NV = 6890, NJ = 24, NB = 10, NP = 207;
constexpr int KX = 224;      // 10 betas + 207 pose features, zero-padded to a multiple of the GEMM's 32-deep K tile

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
                                                       float* __restrict__ xf, float* __restrict__ Jtr, unsigned* __restrict__ cnt) {
    __shared__ float J[NJ][3];
    __shared__ float G[NJ][12];
    __shared__ float R[NJ][9];
    __shared__ float bs[NB];
    const int b = blockIdx.x, t = threadIdx.x;
    // arrival counter of this crop's 27 skin workgroups (lbs_skin_joints_kernel): zeroed HERE, by the launch that always precedes
    // them on the stream, so an aborted launch cannot leave a count behind for the next call (round 2: the last arriver re-zeroed it)
    if (t == 0) cnt[b] = 0u;
    for (int i = t; i < NJ * 9; i += 128) R[i / 9][i % 9] = rotmat[(int64_t)b * NJ * 9 + i];
    if (t < NB) bs[t] = betas[(int64_t)b * NB + t];
    __syncthreads();
    if (t < NJ * 3) {
        float v = 0.f;
#pragma unroll
        for (int l = 0; l < NB; ++l) v = fmaf(bs[l], Jsd[t * NB + l], v);
        J[t / 3][t % 3] = Jt[t] + v;
    }
    // blend-shape GEMM operand row: [betas (10) | pose_feature = (R[1:] - I).view(207) (smplx lbs.py) | 0 x 7]
    for (int i = t; i < KX; i += 128) {
        float v = 0.f;
        if (i < NB) v = bs[i];
        else if (i < NB + NP) {
            const int q = i - NB, j = 1 + q / 9, e = q % 9;
            v = R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
        xf[(int64_t)b * KX + i] = v;
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

// ---- one-time: dirs^T[n][k], n = 3*vertex + coordinate: k < 10 shapedirs, 10 <= k < 217 posedirs, rest 0 ----
__global__ __launch_bounds__(256) void lbs_build_dirs_kernel(const float* __restrict__ sd, const float* __restrict__ pd,
                                                             float* __restrict__ dirsT) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)NV * 3 * KX) return;
    const int n = (int)(idx / KX), k = (int)(idx % KX);
    float v = 0.f;
    if (k < NB) v = sd[(int64_t)n * NB + k];                         // shapedirs (6890,3,10) == [n][10]
    else if (k < NB + NP) v = pd[(int64_t)(k - NB) * (NV * 3) + n];   // posedirs (207, 20670)
    dirsT[idx] = v;
}

// ---- skinning + joints in ONE kernel.
//   skin:   T = sum_j W[v][j] * A[b][j] (3x4), out = T . [v_posed; 1] — one thread per vertex, a workgroup owns 256 vertices and
//           walks CG crops, so the per-vertex constants (24 skinning weights, the J19 regressor entries) stay in registers for
//           the whole pass.  The crop's 24 bone matrices are staged in LDS (double-buffered, requested one crop ahead) and read
//           as broadcast ds_read_b128.
//   J19:    (smpl_wrapper.py:38-39 vertices2joints) thread (joint j, vertex group g) of 19 x 12 adds its 22 vertices' products
//           for all three coordinates — the regressor entries come from registers, the skinned vertex is ONE broadcast
//           ds_read_b128 — then 57 threads add the 12 group sums in a fixed order.
//   joints: the LAST of a crop group's 27 workgroups to finish (ONE device-scope arrival per pass) adds the 27 partial sums in
//           block order for each crop of the group, picks the 21 extra vertices (vertex_joint_selector), applies joint_map
//           (smpl_wrapper.py:19-20,32), update_hips (:33-36), appends the 19 regressed joints and projects (geometry.py:86-124).
//           What crosses workgroups (partials, the 21 picked vertices) moves with device-scope stores / loads.
// Round 3 (PMC at 512 crops, profiles/r3a_pmc_lbs_b512.json): the round-2 kernel was LDS-bound, not HBM- or latency-bound —
// every thread re-read the 72 float4 of the bone matrices from LDS per crop and the regression did two ds_read_b32 per FMA:
// ~3700 LDS cycles per workgroup and crop = 83 of its 142 us; it also drained its stores and took one device-scope atomic round
// trip PER CROP.  Here: 22 ds_read_b128 per thread for the regression instead of 128 ds_read_b32, one arrival per pass; the 72
// broadcast reads of the bone matrices stay (their 73 KB of LDS return traffic per wave and crop is what is left of the bound). ----
constexpr int SKB = (NV + 255) / 256;     // skin workgroups per crop = 27
constexpr int CG_MAX = 8;                 // most crops per workgroup pass (chosen per call: enough workgroups first)
constexpr int RG = 12, RV = 22;           // regression: 12 vertex groups of 22 (the last one: 14) x 19 joints = 228 threads
__global__ __launch_bounds__(256) void lbs_skin_joints_kernel(const float* __restrict__ vposed, const float* __restrict__ W,
                                                              const float* __restrict__ A, const float* __restrict__ J19,
                                                              const float* __restrict__ Jtr, const int32_t* __restrict__ extra,
                                                              const int32_t* __restrict__ jmap, const int32_t* __restrict__ update_hips,
                                                              const float* __restrict__ cam_t, float* __restrict__ verts,
                                                              float* jpart, float* xv, unsigned* cnt, float* __restrict__ joints,
                                                              float* __restrict__ kp2d, float focal_over_size, int B, int CG) {
    __shared__ f32x4 AS[2][NJ * 3];          // bone matrices of the current / next crop
    __shared__ __attribute__((aligned(16))) float outs[256 * 4];     // the crop's skinned vertices of this workgroup (x, y, z, -)
    __shared__ float part[RG][57];
    __shared__ float jo[44][3];
    __shared__ int s_last;
    const int tid = threadIdx.x, v = blockIdx.x * 256 + tid;
    const bool vok = v < NV;
    const int vv = vok ? v : NV - 1;
    f32x4 wv[NJ / 4];
#pragma unroll
    for (int q = 0; q < NJ / 4; ++q) wv[q] = reinterpret_cast<const f32x4*>(W + (int64_t)vv * NJ)[q];
    // regression thread (rj, rg): joint rj, vertices rg*22 .. rg*22+21 of this workgroup's 256; its regressor entries, once
    const int rj = tid % 19, rg = tid / 19;
    float jw[RV];
#pragma unroll
    for (int u = 0; u < RV; ++u) {
        const int lv = rg * RV + u, gv = blockIdx.x * 256 + lv;
        jw[u] = (rg < RG && lv < 256 && gv < NV) ? J19[(int64_t)rj * NV + gv] : 0.f;
    }
    unsigned slots = 0;                   // which of the 21 extra-joint slots pick this thread's vertex (bit k; an id may repeat)
#pragma unroll
    for (int k = 0; k < 21; ++k) slots |= (extra[k] == v && vok) ? 1u << k : 0u;      // 21 scalar loads in one batch, no branches
    const int b0 = blockIdx.y * CG;
    // the posed vertex of crop c + 1 is requested before crop c is skinned
    float xn = 0.f, yn = 0.f, zn = 0.f;
    f32x4 an = {0.f, 0.f, 0.f, 0.f};
    if (b0 < B) {
        const float* p = vposed + ((int64_t)b0 * NV + vv) * 3;
        xn = p[0]; yn = p[1]; zn = p[2];
        if (tid < NJ * 3) AS[0][tid] = reinterpret_cast<const f32x4*>(A + (int64_t)b0 * NJ * 12)[tid];
    }
    for (int c = 0; c < CG; ++c) {
        const int b = b0 + c;             // wave-uniform
        if (b >= B) break;
        __syncthreads();                  // AS[c & 1] is complete; outs / part of the previous crop are no longer read
        const float x = xn, y = yn, z = zn;
        const bool more = c + 1 < CG && b + 1 < B;
        if (more) {
            const float* p = vposed + ((int64_t)(b + 1) * NV + vv) * 3;
            xn = p[0]; yn = p[1]; zn = p[2];
            if (tid < NJ * 3) an = reinterpret_cast<const f32x4*>(A + (int64_t)(b + 1) * NJ * 12)[tid];
        }
        // (Round 3 also tried the bone matrices as SCALAR loads — 288 floats into SGPRs, one SGPR operand per FMA, no LDS return
        // traffic: each new crop misses the scalar cache and the 12 dependent s_load round trips per crop made the kernel slower,
        // 161 vs 142 us at 512 crops, profiles/r3b_lbs_scalar_loads.log.)
        const f32x4* ASc = AS[c & 1];
        f32x4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float w = wv[j >> 2][j & 3];
            T0 += w * ASc[j * 3 + 0];
            T1 += w * ASc[j * 3 + 1];
            T2 += w * ASc[j * 3 + 2];
        }
        const float ox = T0[0] * x + T0[1] * y + T0[2] * z + T0[3];
        const float oy = T1[0] * x + T1[1] * y + T1[2] * z + T1[3];
        const float oz = T2[0] * x + T2[1] * y + T2[2] * z + T2[3];
        if (vok) {
            float* o = verts + ((int64_t)b * NV + v) * 3;
            o[0] = ox; o[1] = oy; o[2] = oz;
        }
        for (unsigned m = slots; m; m &= m - 1) {
            float* xo = xv + ((int64_t)b * 21 + __builtin_ctz(m)) * 3;
            st_dev(xo + 0, ox); st_dev(xo + 1, oy); st_dev(xo + 2, oz);
        }
        *reinterpret_cast<f32x4*>(outs + tid * 4) = f32x4{vok ? ox : 0.f, vok ? oy : 0.f, vok ? oz : 0.f, 0.f};
        __syncthreads();
        if (tid < 19 * RG) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int u = 0; u < RV; ++u) {
                const int lv = min(rg * RV + u, 255);                 // past the end: weight 0
                const f32x4 o = *reinterpret_cast<const f32x4*>(outs + lv * 4);
                sx = fmaf(jw[u], o[0], sx); sy = fmaf(jw[u], o[1], sy); sz = fmaf(jw[u], o[2], sz);
            }
            part[rg][rj * 3 + 0] = sx; part[rg][rj * 3 + 1] = sy; part[rg][rj * 3 + 2] = sz;
        }
        __syncthreads();
        if (tid < 57) {
            float sacc = part[0][tid];
#pragma unroll
            for (int g = 1; g < RG; ++g) sacc += part[g][tid];
            st_dev(jpart + ((int64_t)b * SKB + blockIdx.x) * 57 + tid, sacc);
        }
        if (more && tid < NJ * 3) AS[(c + 1) & 1][tid] = an;          // next crop's bone matrices (landed during the regression)
    }
    // ONE arrival per pass: this workgroup's device-scope stores for all its crops have completed before it counts itself in
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(SKB - 1);
    __syncthreads();
    if (!s_last) return;
    for (int c = 0; c < CG; ++c) {        // all 27 workgroups of this crop group have arrived: finish its crops
        const int b = b0 + c;
        if (b >= B) break;
        if (tid < 57) {                   // J19 regressor: the partial sums of workgroups 0 .. 26 in order
            float sacc = 0.f;
            for (int k = 0; k < SKB; ++k) sacc += ld_dev(jpart + ((int64_t)b * SKB + k) * 57 + tid);
            jo[25 + tid / 3][tid % 3] = sacc;
        }
        if (tid >= 64 && tid < 64 + 75) {
            const int t = tid - 64, j = t / 3, i = t % 3;
            const int src = jmap[j];
            jo[j][i] = (src < NJ) ? Jtr[((int64_t)b * NJ + src) * 3 + i] : ld_dev(xv + ((int64_t)b * 21 + (src - NJ)) * 3 + i);
        }
        __syncthreads();
        // SMPL(update_hips=True), smpl_wrapper.py:33-36, on the 25 mapped joints (before the extra joints are appended):
        //   j[9,12] = (j[9,12] + 0.25*(j[9,12] - j[12,9])) + 0.5*(j[8] - 0.5*(j[9,12] + j[12,9]))
        if (*update_hips && tid < 3) {
            const float a = jo[9][tid], cc = jo[12][tid], m = jo[8][tid];
            jo[9][tid] = (a + 0.25f * (a - cc)) + 0.5f * (m - 0.5f * (a + cc));
            jo[12][tid] = (cc + 0.25f * (cc - a)) + 0.5f * (m - 0.5f * (cc + a));
        }
        __syncthreads();
        if (tid < 132 && joints) joints[(int64_t)b * 132 + tid] = jo[tid / 3][tid % 3];
        if (tid < 44 && kp2d && cam_t) {
            const float px = jo[tid][0] + cam_t[b * 3 + 0], py = jo[tid][1] + cam_t[b * 3 + 1], pz = jo[tid][2] + cam_t[b * 3 + 2];
            kp2d[((int64_t)b * 44 + tid) * 2 + 0] = (px / pz) * focal_over_size;
            kp2d[((int64_t)b * 44 + tid) * 2 + 1] = (py / pz) * focal_over_size;
        }
        __syncthreads();                  // jo is rewritten by the next crop
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

int launch_lbs_build_dirs(const float* sd, const float* pd, float* dirsT, hipStream_t s) {
    const int64_t total = (int64_t)NV * 3 * KX;
    hipLaunchKernelGGL(lbs_build_dirs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sd, pd, dirsT);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// scratch: A (B,24,12), xf (B, 224 + 27*57) [operand rows, then J19 partial sums], Jtr (B,24,3), vposed (B,20670),
//          xv (B,21,3) picked extra-joint vertices, cnt (B) arrival counters (zeroed by lbs_prep_kernel of the same call)
int launch_lbs(const float* rotmat, const float* betas, const float* cam_t, const float* Jt, const float* Jsd,
               const int32_t* parents, const float* vt, const float* dirsT, const float* W, const float* J19,
               const int32_t* extra, const int32_t* jmap, const int32_t* update_hips, float* A, float* xf, float* Jtr,
               float* vposed, float* verts, float* joints, float* kp2d, float focal_over_size, int B, float* xv,
               unsigned* cnt, hipStream_t s) {
    hipLaunchKernelGGL(lbs_prep_kernel, dim3(B), dim3(128), 0, s, rotmat, betas, Jt, Jsd, parents, A, xf, Jtr, cnt);
    GemmArgs g{};
    g.A = xf; g.lda = KX; g.W = dirsT; g.ldw = KX; g.bias = vt; g.resid = nullptr; g.ldr = 0;
    g.C = vposed; g.ldc = NV * 3; g.M = B; g.N = NV * 3; g.K = KX; g.qscale = 1.f; g.qcols = 0;
    if (int r = launch_gemm(g, EPI_BIAS, -1, s)) return r;
    // the J19 partial sums live behind the (B,224) operand rows in the xf scratch: B * 27 * 57 floats
    float* jpart = xf + (size_t)B * KX;
    // crops per workgroup pass: reuse of the per-vertex constants only pays once the grid already fills the chip several times
    // (64 crops: 2 -> 864 workgroups; 256 crops and up: 8)
    const int cg = B >= 32 * CG_MAX ? CG_MAX : (B >= 32 ? B / 32 : 1);
    hipLaunchKernelGGL(lbs_skin_joints_kernel, dim3(SKB, (B + cg - 1) / cg), dim3(256), 0, s, vposed, W, A, J19, Jtr, extra, jmap,
                       update_hips, cam_t, verts, jpart, xv, cnt, joints, kp2d, focal_over_size, B, cg);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
