// Micro-benchmark 3: where does the hardware dispatcher put the workgroups of a grid that is 1.5x the resident capacity?
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/dispatch_map scripts/micro/dispatch_map.hip
// Every block (256 threads, sized so that two fit on a CU like the product GEMM: 72 KB of LDS) records its XCC / SE / CU ids
// (s_getreg HW_ID, XCC_ID), start and end time, then spins on MFMAs for a fixed number of iterations.  The host prints, for
// grids of 256 / 512 / 768 / 1024 blocks, how many blocks each CU received in the FIRST wave and in the TRAILING wave.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Rec {
    unsigned hw, xcc;
    unsigned long long t0, t1;
};

__global__ __launch_bounds__(256) void k(Rec* rec, float* out, int iters, int jitter) {
    __shared__ float pad[18432];      // 72 KB -> two blocks per CU
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, bits 0..31
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    unsigned long long t0 = wall_clock64();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-9f, b = 0.5f + pad[threadIdx.x & 63] * 1e-9f;
    const int n = iters + (jitter ? (int)((blockIdx.x * 2654435761u) >> 24) * jitter : 0);     // optional per-block duration spread
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{hw, xcc, t0, (unsigned long long)wall_clock64()};
}

static void run(int blocks, int iters, int jitter) {
    Rec* d;
    float* out;
    (void)hipMalloc(&d, sizeof(Rec) * blocks);
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, out, iters, jitter);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, out, iters, jitter);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> r(blocks);
    (void)hipMemcpy(r.data(), d, sizeof(Rec) * blocks, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tend = 0;
    for (auto& x : r) { tmin = std::min(tmin, x.t0); tend = std::max(tend, x.t1); }
    // a block belongs to the trailing wave if it started after the earliest END of any block
    unsigned long long first_end = ~0ull;
    for (auto& x : r) first_end = std::min(first_end, x.t1);
    std::map<unsigned, int> first, trail;
    for (auto& x : r) {
        // CU key: xcc id (4 bits) | se_id, sh_id, cu_id = HW_ID bits 8..15
        const unsigned key = ((x.xcc & 0xF) << 8) | ((x.hw >> 8) & 0xFF);
        (x.t0 >= first_end ? trail : first)[key]++;
    }
    auto hist = [](const std::map<unsigned, int>& m) {
        int h[5] = {0, 0, 0, 0, 0};
        for (auto& kv : m) h[std::min(kv.second, 4)]++;
        return std::vector<int>(h, h + 5);
    };
    auto hf = hist(first), ht = hist(trail);
    int nfirst = 0, ntrail = 0;
    for (auto& kv : first) nfirst += kv.second;
    for (auto& kv : trail) ntrail += kv.second;
    printf("blocks %5d jitter %d: %7.3f ms | first wave %4d blocks on %3zu CUs (CUs with 1/2/3/4+ blocks: %d/%d/%d/%d) | trailing %4d blocks on %3zu CUs "
           "(1/2/3/4+: %d/%d/%d/%d) | span %.3f ms (100 MHz clock)\n",
           blocks, jitter, ms, nfirst, first.size(), hf[1], hf[2], hf[3], hf[4], ntrail, trail.size(), ht[1], ht[2], ht[3], ht[4],
           (tend - tmin) / 1e5);
    (void)hipFree(d);
    (void)hipFree(out);
}

int main() {
    const int iters = 6000;
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {256, 512, 768, 1024, 1280}) run(blocks, iters, 0);
        for (int blocks : {768, 1280}) run(blocks, iters, 8);      // blocks of slightly different length (drift, as real tiles have)
    }
    return 0;
}
