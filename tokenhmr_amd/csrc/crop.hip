// Crop preprocessing on the GPU: decoded uint8 frame + per-crop affine -> normalised (n,3,P,P) fp32
// crops, i.e. `batch['img']` of the hot path.
//
// Replaces, per crop, on the CPU side of the reference:
//   skimage.filters.gaussian (anti-alias blur of the WHOLE frame in float64)   lib/datasets/vitdet_dataset.py:62-68, utils.py:583-587
//   cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT)                               lib/datasets/utils.py:351-356
//   [:, :, ::-1], HWC->CHW float32, (x - mean) / std                            vitdet_dataset.py:75-80, utils.py:599-617
//
// Arithmetic follows the third-party code bit for bit so that results equal the CPU path's, not merely approximate it:
//   * warp: OpenCV's fixed-point source coordinates (AB_BITS 10, INTER_BITS 5): X = (rint((M1*y+M2)*1024) + 16 +
//     rint(M0*x*1024)) >> 5; uint8 frames use the integer weights 32*a*b and (sum + 2^14) >> 15; blurred (float64) frames use
//     the float table a*b/1024 with the 4-term sum accumulated in double, left to right; zero border.
//   * blur: scipy.ndimage.correlate1d's symmetric-kernel loop in fp64 — centre*w0 + sum_jj (x[l+jj] + x[l-jj])*w[jj], rows
//     first, then columns, edge-replicated ('nearest') — with NO fused multiply-adds (this file is compiled contract-off).
//     Only the part of the frame a crop can sample (+ the kernel radius) is blurred, which is identical inside that region.
//   * normalisation in float32: (x - float(mean)) / float(std), IEEE division (numpy 1.23 semantics, requirements.txt:1).
//
// gfx950 notes: this is HBM/latency-bound byte work (a 256x256 crop reads <= 4 source pixels per output pixel); one thread
// per output pixel with the three channels together so the 3-byte BGR texels are fetched once, x-fastest indexing for
// coalesced fp32 stores; the two blur passes are one thread per (row, column, channel) element.  fp64 VALU is plentiful
// on CDNA4, so the blur is done in the reference's own precision instead of being approximated in fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/tokenhmr_hip.h"

#pragma clang fp contract(off)

namespace {

struct CropDev {
    double Mi[6];                 // inverse affine (dst -> src), as cv::warpAffine computes it
    int32_t blur, lw;             // blur on/off, kernel radius
    int32_t rx0, ry0, rw, rh;     // blurred region in frame coordinates (clipped to the frame)
    int32_t tx0, tw;              // column range of the row-pass output (region +- lw, clipped)
    int32_t w_off, pad;           // offset of this crop's centred weights in the weight array
    int64_t tmp_off, blur_off;    // offsets (doubles) into the scratch buffer
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// rows pass: tmp[y][x][c] over the region rows and the widened column range
__global__ __launch_bounds__(256) void crop_vpass_kernel(const uint8_t* __restrict__ frame, int H, int W, int64_t stride,
                                                         const CropDev* __restrict__ cds, const double* __restrict__ wts,
                                                         double* __restrict__ scratch) {
    const CropDev c = cds[blockIdx.y];
    if (!c.blur) return;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)c.rh * c.tw * 3) return;
    const int ch = (int)(idx % 3), x = (int)((idx / 3) % c.tw), y = (int)(idx / (3 * (int64_t)c.tw));
    const int gx = c.tx0 + x, gy = c.ry0 + y;
    const double* fw = wts + c.w_off + c.lw;
    const uint8_t* col = frame + (int64_t)gx * 3 + ch;
    double t = (double)col[(int64_t)gy * stride] * fw[0];
    for (int jj = -c.lw; jj < 0; ++jj) {
        const double a = (double)col[(int64_t)clampi(gy + jj, 0, H - 1) * stride];
        const double b = (double)col[(int64_t)clampi(gy - jj, 0, H - 1) * stride];
        t = t + (a + b) * fw[jj];
    }
    scratch[c.tmp_off + idx] = t;
}

// columns pass: blur[y][x][c] over the region
__global__ __launch_bounds__(256) void crop_hpass_kernel(int W, const CropDev* __restrict__ cds, const double* __restrict__ wts,
                                                         double* __restrict__ scratch) {
    const CropDev c = cds[blockIdx.y];
    if (!c.blur) return;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)c.rh * c.rw * 3) return;
    const int ch = (int)(idx % 3), x = (int)((idx / 3) % c.rw), y = (int)(idx / (3 * (int64_t)c.rw));
    const int gx = c.rx0 + x;
    const double* fw = wts + c.w_off + c.lw;
    const double* row = scratch + c.tmp_off + ((int64_t)y * c.tw) * 3 + ch;
    double t = row[(int64_t)(gx - c.tx0) * 3] * fw[0];
    for (int jj = -c.lw; jj < 0; ++jj) {
        const double a = row[(int64_t)(clampi(gx + jj, 0, W - 1) - c.tx0) * 3];
        const double b = row[(int64_t)(clampi(gx - jj, 0, W - 1) - c.tx0) * 3];
        t = t + (a + b) * fw[jj];
    }
    scratch[c.blur_off + idx] = t;
}

__global__ __launch_bounds__(256) void crop_warp_kernel(const uint8_t* __restrict__ frame, int H, int W, int64_t stride,
                                                        const CropDev* __restrict__ cds, const double* __restrict__ scratch,
                                                        int P, int swap_rb, float m0, float m1, float m2, float s0, float s1,
                                                        float s2, float* __restrict__ out) {
    const CropDev c = cds[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * P) return;
    const int x = idx % P, y = idx / P;
    // WarpAffineInvoker (imgwarp.cpp): AB_BITS = 10, round_delta = 16, 5 fractional bits kept
    const long long X0 = __double2ll_rn((c.Mi[1] * (double)y + c.Mi[2]) * 1024.0) + 16;
    const long long Y0 = __double2ll_rn((c.Mi[4] * (double)y + c.Mi[5]) * 1024.0) + 16;
    const long long ad = __double2ll_rn(c.Mi[0] * (double)x * 1024.0);
    const long long bd = __double2ll_rn(c.Mi[3] * (double)x * 1024.0);
    const long long X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
    long long sxl = X >> 5, syl = Y >> 5;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);      // saturate_cast<short>
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
    const bool in00 = sx >= 0 && sx < W && sy >= 0 && sy < H, in01 = sx + 1 >= 0 && sx + 1 < W && sy >= 0 && sy < H;
    const bool in10 = sx >= 0 && sx < W && sy + 1 >= 0 && sy + 1 < H, in11 = sx + 1 >= 0 && sx + 1 < W && sy + 1 >= 0 && sy + 1 < H;
    float v[3];
    if (!c.blur) {
        const int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
        const uint8_t* p = frame + (int64_t)sy * stride + (int64_t)sx * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int a = in00 ? p[ch] : 0, b = in01 ? p[3 + ch] : 0, d = in10 ? p[stride + ch] : 0, e = in11 ? p[stride + 3 + ch] : 0;
            v[ch] = (float)((a * w00 + b * w01 + d * w10 + e * w11 + (1 << 14)) >> 15);
        }
    } else {
        const float t = 1.0f / 32.0f;
        const float fx1 = (float)fx * t, fy1 = (float)fy * t, fx0 = 1.0f - fx1, fy0 = 1.0f - fy1;
        const double w00 = (double)(fy0 * fx0), w01 = (double)(fy0 * fx1), w10 = (double)(fy1 * fx0), w11 = (double)(fy1 * fx1);
        // the blurred region covers every in-frame texel this crop samples (host-side bounding box of sx / sy)
        const double* p = scratch + c.blur_off + ((int64_t)(sy - c.ry0) * c.rw + (sx - c.rx0)) * 3;
        const int64_t rs = (int64_t)c.rw * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const double a = in00 ? p[ch] : 0.0, b = in01 ? p[3 + ch] : 0.0, d = in10 ? p[rs + ch] : 0.0, e = in11 ? p[rs + 3 + ch] : 0.0;
            v[ch] = (float)(((a * w00 + b * w01) + d * w10) + e * w11);
        }
    }
    float* o = out + (int64_t)blockIdx.y * 3 * P * P + idx;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};      // indexed by OUTPUT channel
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int oc = swap_rb ? 2 - ch : ch;
        o[(int64_t)oc * P * P] = (v[ch] - mean[oc]) / sd[oc];
    }
}

thread_local std::string g_crop_err;
int cfail(int code, const std::string& m) { g_crop_err = m; return code; }

// numpy's pairwise sum (loops_utils.h.src) for n <= 128 — the normalisation of scipy's _gaussian_kernel1d uses ndarray.sum()
double np_sum(const std::vector<double>& a) {
    const size_t n = a.size();
    if (n < 8) {
        double r = 0.;
        for (size_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

}  // namespace

struct thmr_cropper {
    int device = 0;
    double* scratch = nullptr;
    size_t scratch_doubles = 0;
    CropDev* cds = nullptr;
    double* wts = nullptr;
    size_t cds_cap = 0, wts_cap = 0;
    std::string err;
};

extern "C" {

const char* thmr_cropper_last_error(const thmr_cropper* c) { return c ? c->err.c_str() : g_crop_err.c_str(); }

int thmr_cropper_create(int32_t device, thmr_cropper** out) {
    if (!out) return cfail(THMR_ERR_INVALID, "null out");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return cfail(THMR_ERR_HIP, "no such HIP device (the crop kernels have no CPU fallback)");
    }
    thmr_cropper* c = new thmr_cropper();
    c->device = device;
    *out = c;
    return 0;
}

void thmr_cropper_destroy(thmr_cropper* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->cds) (void)hipFree(c->cds);
    if (c->wts) (void)hipFree(c->wts);
    delete c;
}

int thmr_cropper_run(thmr_cropper* c, const uint8_t* frame_dev, int32_t H, int32_t W, int64_t row_stride,
                     const thmr_crop_desc* crops, int32_t n, int32_t patch, int32_t swap_rb, const float* mean, const float* std_,
                     float* out_dev, void* stream) {
    if (!c) return cfail(THMR_ERR_INVALID, "null cropper");
    auto bad = [&](const std::string& m) { c->err = m; g_crop_err = m; return THMR_ERR_INVALID; };
    if (!frame_dev || !crops || !out_dev || !mean || !std_) return bad("null buffer");
    if (H <= 0 || W <= 0 || n <= 0 || patch <= 0 || patch > 4096 || row_stride < (int64_t)W * 3) return bad("bad frame / patch geometry");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipSetDevice(c->device) != hipSuccess) return bad("hipSetDevice failed");

    std::vector<CropDev> cds(n);
    std::vector<double> wts;
    size_t need = 0;
    int64_t max_v = 0, max_h = 0;
    for (int i = 0; i < n; ++i) {
        CropDev& d = cds[i];
        double M[6];
        for (int k = 0; k < 6; ++k) M[k] = crops[i].M[k];
        // cv::warpAffine: invert the forward 2x3 matrix in double
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
        for (int k = 0; k < 6; ++k) {
            if (!std::isfinite(M[k])) return bad("crop " + std::to_string(i) + ": singular or non-finite affine");
            d.Mi[k] = M[k];
        }
        d.blur = 0; d.lw = 0; d.rx0 = d.ry0 = d.rw = d.rh = d.tx0 = d.tw = 0; d.w_off = 0; d.pad = 0; d.tmp_off = d.blur_off = 0;
        const double sigma = crops[i].sigma;
        if (!(sigma >= 0) || !std::isfinite(sigma) || !(crops[i].truncate > 0)) return bad("crop " + std::to_string(i) + ": bad sigma / truncate");
        if (sigma > 1e-15) {          // scipy.ndimage.gaussian_filter skips axes with sigma <= 1e-15
            // bounding box of the texels the warp can touch: the map is affine, so the extremes are at the patch corners
            long long lo_x = INT64_MAX, hi_x = INT64_MIN, lo_y = INT64_MAX, hi_y = INT64_MIN;
            for (int cy = 0; cy < 2; ++cy)
                for (int cx = 0; cx < 2; ++cx) {
                    const double x = cx ? patch - 1 : 0, y = cy ? patch - 1 : 0;
                    const long long X = ((long long)std::llrint((M[1] * y + M[2]) * 1024.0) + 16 + (long long)std::llrint(M[0] * x * 1024.0)) >> 10;
                    const long long Y = ((long long)std::llrint((M[4] * y + M[5]) * 1024.0) + 16 + (long long)std::llrint(M[3] * x * 1024.0)) >> 10;
                    lo_x = std::min(lo_x, X); hi_x = std::max(hi_x, X); lo_y = std::min(lo_y, Y); hi_y = std::max(hi_y, Y);
                }
            // +-1: the per-pixel sum of two separately rounded terms can differ by one fixed-point step from the corner value
            const long long x0 = std::max<long long>(lo_x - 1, 0), x1 = std::min<long long>(hi_x + 2, W - 1);
            const long long y0 = std::max<long long>(lo_y - 1, 0), y1 = std::min<long long>(hi_y + 2, H - 1);
            if (x0 <= x1 && y0 <= y1) {
                d.blur = 1;
                d.lw = (int)(crops[i].truncate * sigma + 0.5);            // scipy: int(truncate * sd + 0.5)
                if (d.lw > 4096) return bad("crop " + std::to_string(i) + ": blur radius too large");
                d.rx0 = (int)x0; d.ry0 = (int)y0; d.rw = (int)(x1 - x0 + 1); d.rh = (int)(y1 - y0 + 1);
                d.tx0 = (int)std::max<long long>(x0 - d.lw, 0);
                d.tw = (int)(std::min<long long>(x1 + d.lw, W - 1) - d.tx0 + 1);
                // scipy _gaussian_kernel1d: exp(-0.5 / sigma^2 * x^2), normalised by the numpy sum
                std::vector<double> phi(2 * d.lw + 1);
                const double s2 = sigma * sigma;
                for (int k = -d.lw; k <= d.lw; ++k) phi[k + d.lw] = std::exp(-0.5 / s2 * (double)(k * k));
                const double tot = np_sum(phi);
                d.w_off = (int)wts.size();
                for (double p : phi) wts.push_back(p / tot);
                d.tmp_off = (int64_t)need; need += (size_t)d.rh * d.tw * 3;
                d.blur_off = (int64_t)need; need += (size_t)d.rh * d.rw * 3;
                max_v = std::max<int64_t>(max_v, (int64_t)d.rh * d.tw * 3);
                max_h = std::max<int64_t>(max_h, (int64_t)d.rh * d.rw * 3);
            }
        }
    }
    auto hip_bad = [&](const char* what, hipError_t e) { c->err = std::string(what) + ": " + hipGetErrorString(e); g_crop_err = c->err; return THMR_ERR_HIP; };
    hipError_t e;
    // grow-only device buffers (re-allocation synchronises the stream first: earlier launches may still read the old ones)
    if ((size_t)n > c->cds_cap || wts.size() > c->wts_cap || need > c->scratch_doubles) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_bad("hipStreamSynchronize", e);
        if ((size_t)n > c->cds_cap) {
            if (c->cds) (void)hipFree(c->cds);
            c->cds = nullptr; c->cds_cap = 0;
            if ((e = hipMalloc(&c->cds, sizeof(CropDev) * (size_t)n * 2)) != hipSuccess) return hip_bad("hipMalloc(crop descriptors)", e);
            c->cds_cap = (size_t)n * 2;
        }
        if (wts.size() > c->wts_cap) {
            if (c->wts) (void)hipFree(c->wts);
            c->wts = nullptr; c->wts_cap = 0;
            if ((e = hipMalloc(&c->wts, sizeof(double) * wts.size() * 2)) != hipSuccess) return hip_bad("hipMalloc(weights)", e);
            c->wts_cap = wts.size() * 2;
        }
        if (need > c->scratch_doubles) {
            if (c->scratch) (void)hipFree(c->scratch);
            c->scratch = nullptr; c->scratch_doubles = 0;
            if ((e = hipMalloc(&c->scratch, sizeof(double) * need)) != hipSuccess) return hip_bad("hipMalloc(blur scratch)", e);
            c->scratch_doubles = need;
        }
    }
    // pageable-host -> device copies return after staging, so the vectors may die at the end of this call
    if ((e = hipMemcpyAsync(c->cds, cds.data(), sizeof(CropDev) * n, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_bad("hipMemcpyAsync", e);
    if (!wts.empty() && (e = hipMemcpyAsync(c->wts, wts.data(), sizeof(double) * wts.size(), hipMemcpyHostToDevice, st)) != hipSuccess)
        return hip_bad("hipMemcpyAsync", e);
    if (max_v > 0) {
        hipLaunchKernelGGL(crop_vpass_kernel, dim3((unsigned)((max_v + 255) / 256), n), dim3(256), 0, st, frame_dev, H, W, row_stride,
                           c->cds, c->wts, c->scratch);
        hipLaunchKernelGGL(crop_hpass_kernel, dim3((unsigned)((max_h + 255) / 256), n), dim3(256), 0, st, W, c->cds, c->wts, c->scratch);
    }
    hipLaunchKernelGGL(crop_warp_kernel, dim3((patch * patch + 255) / 256, n), dim3(256), 0, st, frame_dev, H, W, row_stride, c->cds,
                       c->scratch, patch, swap_rb, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], out_dev);
    if ((e = hipGetLastError()) != hipSuccess) return hip_bad("crop kernel launch", e);
    return 0;
}

}  // extern "C"
