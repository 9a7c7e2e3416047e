// Micro-benchmark 7: are the two co-resident blocks of a CU in lockstep in the big-tile GEMM, and does de-phasing them pay?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o build_ab/gemm_dephase scripts/micro/gemm_dephase.hip
// The product 128x160 LDS-DMA kernel on the four ViT shapes at 64 crops (M = 12288); the block on the odd threadgroup slot of
// each CU starts D us late in the first round (512 blocks) only.  Results stay bit-identical; D = 0 is the product behaviour.
#include "../../tokenhmr_amd/csrc/gemm_f32.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

template <int EPI, int DPH>
static void run(const GemmArgs& a) {
    constexpr int BM = 128, BN = 160;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int tail = tiles % kSlots;
    const bool isolate = tiles > kSlots && tail > 0 && tail <= kSlots / 2;
    const int main_tiles = isolate ? tiles - tail : tiles;
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1, 5, true, EPI, 0, 0, 0, DPH>), dim3(main_tiles), dim3(256), 0, 0, a, tiles_m, tiles_n, main_tiles, 0);
    if (isolate)
        hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1, 5, true, EPI, 0, 0, 0, DPH>), dim3(tail), dim3(256), kTailLds, 0, a, tiles_m, tiles_n, tail, main_tiles);
}

struct Shape {
    const char* name;
    int N, K, epi;
};

int main() {
    const int M = 12288;
    const Shape shapes[] = {{"qkv", 3840, 1280, EPI_BIAS_QSCALE}, {"proj", 1280, 1280, EPI_BIAS_RESID}, {"fc1", 5120, 1280, EPI_BIAS_GELU}, {"fc2", 1280, 5120, EPI_BIAS_RESID}};
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hA((size_t)M * 5120), hW((size_t)5120 * 1280);
    for (auto& x : hA) x = nd(rng);
    for (auto& x : hW) x = nd(rng) * 0.02f;
    float *A, *W, *bias, *C, *R;
    (void)hipMalloc(&A, hA.size() * 4);
    (void)hipMalloc(&W, hW.size() * 4);
    (void)hipMalloc(&bias, 5120 * 4);
    (void)hipMalloc(&C, (size_t)M * 5120 * 4);
    (void)hipMalloc(&R, (size_t)M * 5120 * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(bias, 0, 5120 * 4);
    (void)hipMemset(R, 0, (size_t)M * 5120 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int delays[] = {0, 5, 10, 20, 40, 60, 85};
    for (const Shape& sh : shapes) {
        GemmArgs a{};
        a.A = A; a.W = W; a.bias = bias; a.C = C; a.resid = R;
        a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.ldr = sh.N;
        a.M = M; a.N = sh.N; a.K = sh.K; a.qscale = 0.1118f; a.qcols = 1280;
        auto go = [&](bool dph) {
            switch (sh.epi) {
                case EPI_BIAS_QSCALE: dph ? run<EPI_BIAS_QSCALE, 1>(a) : run<EPI_BIAS_QSCALE, 0>(a); break;
                case EPI_BIAS_GELU: dph ? run<EPI_BIAS_GELU, 1>(a) : run<EPI_BIAS_GELU, 0>(a); break;
                default: dph ? run<EPI_BIAS_RESID, 1>(a) : run<EPI_BIAS_RESID, 0>(a); break;
            }
        };
        for (int i = 0; i < 5; ++i) go(false);
        (void)hipDeviceSynchronize();
        std::vector<std::vector<float>> ts(8);
        for (int rep = 0; rep < 7; ++rep) {
            for (int di = -1; di < 7; ++di) {                 // -1: the product instantiation (no HW_ID read at all)
                int ticks = di < 0 ? 0 : delays[di] * 100;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dephase_ticks), &ticks, sizeof(int));
                (void)hipEventRecord(e0);
                for (int i = 0; i < 4; ++i) go(di >= 0);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                ts[di + 1].push_back(ms * 250.f);
            }
        }
        const double fl = 2.0 * M * sh.N * sh.K;
        printf("%-5s", sh.name);
        for (int i = 0; i < 8; ++i) {
            std::sort(ts[i].begin(), ts[i].end());
            const float t = ts[i][ts[i].size() / 2];
            if (i == 0) printf(" product %.1f us (%.1f TF) |", t, fl / t / 1e6);
            else printf(" D=%d: %.1f (%.1f)", delays[i - 1], t, fl / t / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
