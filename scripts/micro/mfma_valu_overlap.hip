// Micro-benchmark: can packed fp32 VALU FMAs (v_pk_fma_f32) run in the shadow of fp32 MFMAs on gfx950, and what does the
// chip sustain when both pipes are busy?  Register-only operands, no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/mfma_valu_overlap scripts/micro/mfma_valu_overlap.hip
// Each wave runs ITER iterations of: NM independent v_mfma_f32_32x32x2_f32 (64 cycles each) interleaved with NV
// v_pk_fma_f32 per MFMA on independent accumulators.  Prints time, MFMA TFLOP/s, VALU TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x, (float)i};
    const float a = a0 + threadIdx.x * 1e-9f, b = b0;
    const f32x2 va = f32x2{a, a * 0.5f}, vb = f32x2{b * 1e-3f, b * 2e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            if constexpr (MF) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = (m * NV + j) % 16;
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(va), "v"(vb));   // hipcc scalarises the builtin here
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool MF>
void run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double waves = (double)blocks * 4;
    const double mf = MF ? waves * iters * 5.0 * 4096.0 : 0.0;                 // 32*32*2*2 flop per MFMA
    const double vf = waves * iters * 5.0 * NV * 256.0;                        // 64 lanes * 2 * 2 flop per v_pk_fma_f32
    printf("blocks %5d  mfma %d  pk_fma/mfma %2d : %8.3f ms   MFMA %6.1f TF   VALU %6.1f TF   sum %6.1f TF\n", blocks, (int)MF, NV, ms,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

// second experiment (argument "occ"): MFMA-only kernel at 2 / 3 / 4 / 8 waves per SIMD in alternating order (the alternation
// separates an occupancy effect from clock drift over the run)
static void occupancy(float* out) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {512, 1024, 768, 2048, 256, 512}) run<0, true>(out, blocks, iters);
}

int main(int argc, char** argv) {
    if (argc > 1) {
        float* o;
        (void)hipMalloc(&o, 4096 * 256 * sizeof(float));
        occupancy(o);
        return 0;
    }
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 20000;
    for (int blocks : {512, 1024}) {      // 2 or 4 blocks per CU = 2 or 4 waves per SIMD
        run<0, true>(out, blocks, iters);
        run<2, true>(out, blocks, iters);
        run<4, true>(out, blocks, iters);
        run<8, true>(out, blocks, iters);
        run<12, true>(out, blocks, iters);
        run<15, true>(out, blocks, iters);
        run<16, true>(out, blocks, iters);
        run<4, false>(out, blocks, iters);
        run<16, false>(out, blocks, iters);
    }
    return 0;
}
