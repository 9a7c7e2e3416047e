// Micro-benchmark 6: what limits the small-batch ring GEMM (gemm_ring_kernel, M = 192 * B rows, B <= 6)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o build_ab/ring_ablation scripts/micro/ring_ablation.hip
// At one crop a K tile takes ~0.7 us per block against 0.43 us of matrix work (16 MFMAs x 64 cycles per wave), and neither a deeper
// ring nor a second co-resident block changed that (profiles/r1_small_gemm_variants.log).  Timing-only ablations of the PRODUCT
// kernel source (results are garbage except for ABL 0 and 8):
//   1 no copies in the K loop   2 no per-tile wait + barrier   4 no LDS fragment reads   3 / 7 combinations
//   8 K sweep rotated by the column-tile index (all blocks no longer fetch the same 128-byte column of every row at the same time:
//     rows are K * 4 bytes apart, a power-of-two-ish stride) — correct results, different summation order
#include "../../tokenhmr_amd/csrc/gemm_f32.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

template <int EPI, bool PARTIAL, int ABL>
static void launch(const GemmArgs& a, int ksplit, float* part) {
    const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 63) / 64;
    const int groups = tiles_n * ksplit;
    dim3 grid(8 * tiles_m * ((groups + 7) / 8)), block(256);
    hipLaunchKernelGGL((gemm_ring_kernel<4, EPI, PARTIAL, ABL>), grid, block, 0, 0, a, tiles_m, groups, ksplit, part);
}

template <int EPI, bool PARTIAL>
static void launch_abl(int abl, const GemmArgs& a, int ksplit, float* part) {
    switch (abl) {
        case 0: launch<EPI, PARTIAL, 0>(a, ksplit, part); break;
        case 1: launch<EPI, PARTIAL, 1>(a, ksplit, part); break;
        case 2: launch<EPI, PARTIAL, 2>(a, ksplit, part); break;
        case 3: launch<EPI, PARTIAL, 3>(a, ksplit, part); break;
        case 4: launch<EPI, PARTIAL, 4>(a, ksplit, part); break;
        case 7: launch<EPI, PARTIAL, 7>(a, ksplit, part); break;
        case 8: launch<EPI, PARTIAL, 8>(a, ksplit, part); break;
        default: break;
    }
}

struct Shape {
    const char* name;
    int N, K, ksplit, epi;
};

int main() {
    const Shape shapes[] = {{"qkv", 3840, 1280, 1, EPI_BIAS}, {"proj", 1280, 1280, 4, -1}, {"fc1", 5120, 1280, 1, EPI_BIAS_GELU}, {"fc2", 1280, 5120, 4, -1},
                            {"fc1/k2", 5120, 1280, 2, -1}, {"qkv/k2", 3840, 1280, 2, -1}};
    const int abls[] = {0, 1, 2, 4, 3, 7, 8};
    const int maxM = 192 * 6;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hA((size_t)maxM * 5120), hW((size_t)5120 * 5120 / 4 * 1);   // W: up to 5120 x 1280 or 1280 x 5120 = 6.55 M floats
    hW.resize((size_t)5120 * 1280);
    for (auto& x : hA) x = nd(rng);
    for (auto& x : hW) x = nd(rng) * 0.02f;
    float *A, *W, *bias, *C, *C2, *part;
    (void)hipMalloc(&A, hA.size() * 4);
    (void)hipMalloc(&W, hW.size() * 4);
    (void)hipMalloc(&bias, 5120 * 4);
    (void)hipMalloc(&C, (size_t)maxM * 5120 * 4);
    (void)hipMalloc(&C2, (size_t)maxM * 5120 * 4);
    (void)hipMalloc(&part, (size_t)4 * maxM * 5120 * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(bias, 0, 5120 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int B : {1, 2, 4, 6}) {
        const int M = 192 * B;
        for (const Shape& sh : shapes) {
            GemmArgs a{};
            a.A = A; a.W = W; a.bias = bias; a.C = C;
            a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.ldr = sh.N;
            a.M = M; a.N = sh.N; a.K = sh.K;
            auto run = [&](int abl) {
                if (sh.epi == EPI_BIAS) launch_abl<EPI_BIAS, false>(abl, a, 1, nullptr);
                else if (sh.epi == EPI_BIAS_GELU) launch_abl<EPI_BIAS_GELU, false>(abl, a, 1, nullptr);
                else launch_abl<EPI_NONE, true>(abl, a, sh.ksplit, part);
            };
            std::vector<std::vector<float>> ts(16);
            for (int abl : abls) for (int i = 0; i < 3; ++i) run(abl);
            (void)hipDeviceSynchronize();
            for (int rep = 0; rep < 9; ++rep)
                for (int abl : abls) {
                    (void)hipEventRecord(e0);
                    for (int i = 0; i < 4; ++i) run(abl);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    ts[abl].push_back(ms * 250.f);
                }
            const int tiles = ((M + 63) / 64) * (sh.N / 64) * sh.ksplit;
            const double floor_us = 2.0 * M * sh.N * sh.K / (157.3e12 * std::min(tiles, 256) / 256.0) * 1e6;
            printf("M %4d %-7s blocks %4d  MFMA floor %5.1f us |", M, sh.name, tiles, floor_us);
            for (int abl : abls) {
                std::sort(ts[abl].begin(), ts[abl].end());
                printf(" abl%d %5.1f", abl, ts[abl][ts[abl].size() / 2]);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    // correctness of the rotated sweep (abl 8) against the product order
    {
        GemmArgs a{};
        a.A = A; a.W = W; a.bias = bias; a.C = C;
        a.lda = 1280; a.ldw = 1280; a.ldc = 5120; a.ldr = 5120;
        a.M = 384; a.N = 5120; a.K = 1280;
        launch<EPI_BIAS, false, 0>(a, 1, nullptr);
        a.C = C2;
        launch<EPI_BIAS, false, 8>(a, 1, nullptr);
        (void)hipDeviceSynchronize();
        std::vector<float> h0((size_t)384 * 5120), h1(h0.size());
        (void)hipMemcpy(h0.data(), C, h0.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h1.data(), C2, h1.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < h0.size(); ++i) { md = std::max(md, (double)fabsf(h0[i] - h1[i])); mx = std::max(mx, (double)fabsf(h0[i])); }
        printf("rotated K sweep vs product order: max |diff| %.3g (max |c| %.3g)\n", md, mx);
    }
    return 0;
}
