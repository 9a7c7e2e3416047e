// Micro-benchmark 9 (round 3): what rate does the bf16 matrix pipe SUSTAIN on this part, and what do the LDS reads of a GEMM loop cost it?
// Context: the split3 GEMM (csrc/gemm_split.hip) keeps the MFMA pipe 74 % busy in cycles, but its counters show the shader clock at
// ~1.53 GHz during the kernel (GRBM_GUI_ACTIVE / duration; the exact-fp32 kernel runs at ~2.16 GHz).  Is that a power ceiling of the
// matrix pipe itself, or the price of the data movement around it?
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/mfma_bf16_sustained scripts/micro/mfma_bf16_sustained.hip && build_ab/mfma_bf16_sustained
// Every variant: 256 workgroups (one per CU) of W waves, each wave issues v_mfma_f32_32x32x16_bf16 round-robin over 4 accumulators for
// ~DUR ms; operands never change (registers).  Variants:
//   pure      : MFMAs only
//   lds R     : R conflict-free ds_read_b128 per 24 MFMAs in their shadow (the split3 kernel has 12), results fed to the operands
//   lds+dma   : plus 9 global_load_lds_dwordx4 (1 KiB each) per 48 MFMAs from an L2-resident buffer (the split3 kernel's copy rate)
// Prints TFLOP/s (bf16) and the implied average clock = TF / (256 CUs x 4 SIMDs x 1024 flop/clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// AGPR: the accumulators are forced into the AGPR half of the register file (inline-asm MFMA with "+a" operands) — does the matrix pipe
// then share less with the VGPR writes of the LDS reads?
template <int R, bool DMA, bool AGPR = false, bool RND = false>
__global__ __launch_bounds__(512) void k(float* out, const char* src, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[144 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 144 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf16x8 fa[12], fb[2];
    // RND: operands with random bits in every position (what real activations / weights look like to the multiplier array) instead of
    // the smooth ramps — is the sustained rate a power limit that depends on the data?
    for (int i = 0; i < 12; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (lane * 2654435761u) ^ ((i * 8 + e + 1) * 40503u) ^ (blockIdx.x * 97u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            fa[i][e] = RND ? (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)) : (__bf16)(0.001f * (lane + i + e));
        }
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (lane * 40503u) ^ ((i * 8 + e + 7) * 2654435761u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            fb[i][e] = RND ? (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)) : (__bf16)(0.002f * (lane + i - e));
        }
    // conflict-free: 16 B per lane, lanes consecutive
    const char* lp = lds + wave * 16384 + lane * 16;
    const uint32_t voff = lane * 16;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + 128 * 1024 + wave * 1024));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(fa[m % 12]), "v"(fb[m & 1]));
            else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m % 12], fb[m & 1], acc[m & 3], 0, 0, 0);
            if ((m % 24) * R / 24 != ((m % 24) + 1) * R / 24 && R > 0) {
                const int r = ((m % 24) * R / 24) % 12;
                fa[r] = *reinterpret_cast<const bf16x8*>(lp + r * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DMA && m % 5 == 0 && m / 5 < 9) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src + (size_t)((it * 9 + m / 5) & 1023) * 1024), "s"(lds_base));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (AGPR) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the compiler cannot see the MFMAs inside the asm: no hazard handling
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// Round 4: the same flops as v_mfma_f32_16x16x32_bf16 (half the accumulator traffic per flop, twice the operand reads): does the part
// sustain a different rate at its power limit?  96 MFMAs of 16,384 flops per iteration over 8 accumulators, random operand bits.
template <bool MOVE>      // MOVE: the split3 kernel's data movement beside the MFMAs: 24 ds_read_b128 + 9 LDS-DMA copies per 96 MFMAs
__global__ __launch_bounds__(512) void k16(float* out, const char* src, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[144 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MOVE) {
        for (int i = threadIdx.x; i < 144 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = (i * 2654435761u) ^ (i >> 3);
        __syncthreads();
    }
    const char* lp = lds + wave * 16384 + lane * 16;
    const uint32_t voff = lane * 16;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + 128 * 1024 + wave * 1024));
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[12], fb[2];
    for (int i = 0; i < 12; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (lane * 2654435761u) ^ ((i * 8 + e + 1) * 40503u) ^ (blockIdx.x * 97u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            fa[i][e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
        }
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (lane * 40503u) ^ ((i * 8 + e + 7) * 2654435761u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            fb[i][e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 96; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m % 12], fb[m & 1], acc[m & 7], 0, 0, 0);
            if (MOVE && (m & 3) == 0) {                 // 24 reads per 96 MFMAs
                const int r = (m >> 2) % 12;
                fa[r] = *reinterpret_cast<const bf16x8*>(lp + r * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MOVE && m % 10 == 5 && m / 10 < 9) {    // 9 copies of 1 KiB per 96 MFMAs
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src + (size_t)((it * 9 + m / 10) & 1023) * 1024), "s"(lds_base));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MOVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int e = 0; e < 4; ++e) s += acc[a][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <bool MOVE>
void run16(const char* name, int waves, float* out, const char* src, double ms_target) {
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k16<MOVE>, dim3(256), dim3(waves * 64), 0, 0, out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) { iters = (int)(iters * ms_target / ms); continue; }
        const double flop = 256.0 * waves * (double)iters * 96 * 16384.0;
        const double tf = flop / (ms * 1e-3) / 1e12;
        printf("%-14s waves/CU %d  %8.2f ms  %8.1f TFLOP/s bf16   implied clock %.2f GHz (if the pipes never idled)\n", name, waves, ms, tf,
               tf * 1e12 / (256.0 * 4 * 1024) / 1e9);
    }
}

template <int R, bool DMA, bool AGPR = false, bool RND = false>
void run(const char* name, int waves, float* out, const char* src, double ms_target) {
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<R, DMA, AGPR, RND>), dim3(256), dim3(waves * 64), 0, 0, out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) { iters = (int)(iters * ms_target / ms); continue; }
        const double flop = 256.0 * waves * (double)iters * 48 * 32768.0;
        const double tf = flop / (ms * 1e-3) / 1e12;
        printf("%-14s waves/CU %d  %8.2f ms  %8.1f TFLOP/s bf16   implied clock %.2f GHz (if the pipes never idled)\n", name, waves, ms, tf,
               tf * 1e12 / (256.0 * 4 * 1024) / 1e9);
    }
}

int main() {
    float* out; char* src;
    hipMalloc(&out, 4096); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    for (int w : {4, 8}) {
        run<0, false>("pure", w, out, src, 60.0);
        run<6, false>("lds 6/24", w, out, src, 60.0);
        run<12, false>("lds 12/24", w, out, src, 60.0);
        run<12, true>("lds 12 + dma", w, out, src, 60.0);
    }
    run<0, false, true>("pure agpr", 8, out, src, 60.0);
    run<12, false, true>("lds 12 agpr", 8, out, src, 60.0);
    run<12, true, true>("lds12+dma agpr", 8, out, src, 60.0);
    run<0, false, false, true>("pure random", 8, out, src, 60.0);
    run<12, true, false, true>("lds12+dma rnd", 8, out, src, 60.0);
    run<0, false, false, true>("pure rnd 300", 8, out, src, 300.0);
    run<0, false>("pure 300 ms", 8, out, src, 300.0);
    run16<false>("16x16x32 rnd", 8, out, src, 300.0);
    run<0, false, false, true>("32x32x16 rnd", 8, out, src, 300.0);
    run16<true>("16x16 rnd+move", 8, out, src, 300.0);
    run<12, true, false, true>("32x32 rnd+move", 8, out, src, 300.0);
    run16<true>("16x16 rnd+move", 8, out, src, 300.0);
    run<12, true, false, true>("32x32 rnd+move", 8, out, src, 300.0);
    return 0;
}
