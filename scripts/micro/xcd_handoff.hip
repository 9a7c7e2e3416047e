// Micro-benchmark 8 (round 3): what does it cost to hand MEGABYTES from one stage to the next INSIDE a persistent kernel on this
// 8-XCD part — the open question behind a persistent whole-ViT-layer kernel for B <= 6 (HISTORY.md §9).
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/xcd_handoff scripts/micro/xcd_handoff.hip && build_ab/xcd_handoff
// The decoder's persistent kernel moves a few KB per step with device-scope dword atomics; a ViT stage at one crop hands over
// 192 x 1280 floats (0.98 MB: x / h / attention output) up to 192 x 5120 (3.9 MB: the MLP hidden), and the consumer is a GEMM whose
// 64-row A panel is re-read by every one of its 60-80 column tiles.  Shape of the test (G persistent workgroups of 256 threads, one
// per CU, ITERS iterations): produce = every workgroup writes its 1/G share of a ROWS x 1280 buffer; grid barrier (flags, bounded);
// consume = every workgroup reads one whole 64-row panel of it (what a column tile of the next GEMM does) and checks every value.
// Variants of the data path:
//   0  kernel boundary: produce and consume are two launches, plain stores / loads          (what the product does today)
//   1  plain 16 B stores + agent-scope release fence (L2 write-back) | barrier | acquire fence (L2 invalidate) + plain loads
//   2  agent-scope dword atomic stores / loads (sc1), the barrier only orders                (the decoder kernel's protocol)
//   3  16 B sc1 stores + 16 B sc1 loads (inline asm), the barrier only orders
//   4  16 B sc1 stores + LDS-DMA (global_load_lds_dwordx4 sc1) of the panel, 1 KiB per wave instruction, then ds_read
// Prints us per iteration (produce + barrier + consume), the produce-only and consume-only times, and the number of wrong values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define AGENT __HIP_MEMORY_SCOPE_AGENT
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int COLS = 1280, NT = 256;

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

// bounded flag barrier (same structure as decoder_fused.hip): returns false on timeout
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned epoch, int tid, volatile int* s_ok, bool fence) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int G = gridDim.x;
    constexpr unsigned LIMIT = 1u << 21;
    if (tid == 0) {
        if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&sync[256 + blockIdx.x], epoch, __ATOMIC_RELAXED, AGENT);
        *s_ok = 1;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid < G) {
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(&sync[256 + tid], __ATOMIC_RELAXED, AGENT) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > LIMIT) { *s_ok = 0; break; }
            }
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELAXED, AGENT);
    } else if (tid == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2 * LIMIT) { *s_ok = 0; break; }
        }
    }
    if (tid == 0 && fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return *s_ok != 0;
}

__device__ __forceinline__ float expect(int it, int idx) { return (float)((it * 131 + idx) & 0xFFFF); }

__device__ __forceinline__ void store16_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load16_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void dma16_sc1(const float* base, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(voff), "s"(base), "s"(lds_base) : "memory");
}

template <int V>
__device__ __forceinline__ void produce(float* buf, int rows, int it, int tid) {
    const int total4 = rows * COLS / 4;
    for (int i = blockIdx.x * NT + tid; i < total4; i += gridDim.x * NT) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expect(it, i * 4 + e);
        if (V == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) __hip_atomic_store(buf + i * 4 + e, v[e], __ATOMIC_RELAXED, AGENT);
        } else if (V >= 3) store16_sc1(buf + i * 4, v);
        else *reinterpret_cast<f32x4*>(buf + i * 4) = v;
    }
}

// every workgroup reads the 64-row panel (blockIdx % (rows/64)) completely; returns the number of wrong values seen by this thread.
// Eight 16-byte loads per thread are in flight at a time (a GEMM's ring keeps 3-4 K tiles = 12-16 KB per wave in flight); the LDS-DMA
// variant double-buffers 16 KiB chunks and tracks them with vmcnt like the ring kernel does.
template <int V>
__device__ __forceinline__ unsigned consume(const float* buf, int rows, int it, int tid, float* lds) {
    const int panel = blockIdx.x % (rows / 64);
    const float* p = buf + (size_t)panel * 64 * COLS;
    const int base_idx = panel * 64 * COLS;
    unsigned bad = 0;
    if (V == 4) {
        // 64 rows x 1280 floats = 320 KiB: 20 chunks of 16 KiB through a 2 x 16 KiB LDS ring; a wave instruction moves 1 KiB
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const uint32_t l0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
        auto issue = [&](int c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int piece = c * 16 + q * 4 + wave;                              // 1 KiB pieces of the panel, in order
                dma16_sc1(p + (size_t)piece * 256, (uint32_t)lane * 16u, l0 + (uint32_t)((c & 1) * 16 + q * 4 + wave) * 1024u);
            }
        };
        issue(0);
        for (int c = 0; c < 20; ++c) {
            if (c + 1 < 20) {
                issue(c + 1);
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");      // chunk c has landed (in-order return), c + 1 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            const float* L = lds + (c & 1) * 4096;
            for (int i = tid; i < 4096; i += NT) bad += L[i] != expect(it, base_idx + c * 4096 + i);
            __syncthreads();                                                          // the buffer is rewritten by chunk c + 2
        }
        return bad;
    }
    constexpr int U = 8;
    for (int i0 = tid; i0 < 64 * COLS / 4; i0 += NT * U) {
        f32x4 v[U];
        if (V == 3) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = min(i0 + k * NT, 64 * COLS / 4 - 1);
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(p + i * 4) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
        } else {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = min(i0 + k * NT, 64 * COLS / 4 - 1);
                if (V == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[k][e] = __hip_atomic_load(p + i * 4 + e, __ATOMIC_RELAXED, AGENT);
                } else v[k] = *reinterpret_cast<const f32x4*>(p + i * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = min(i0 + k * NT, 64 * COLS / 4 - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) bad += v[k][e] != expect(it, base_idx + i * 4 + e);
        }
    }
    return bad;
}

template <int V>
__global__ __launch_bounds__(NT) void persistent(unsigned* sync, float* buf, int rows, int iters, unsigned* errors, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 4096];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    unsigned bad = 0, epoch = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, AGENT);
    unsigned long long tp = 0, tc = 0;
    bool ok = true;
    for (int it = 1; ok && it <= iters; ++it) {
        unsigned long long t0 = wall();
        produce<V>(buf, rows, it, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t1 = wall();
        ok = grid_barrier(sync, ++epoch, tid, &s_ok, V == 1);
        if (!ok) break;
        unsigned long long t2 = wall();
        bad += consume<V>(buf, rows, it, tid, lds);
        unsigned long long t3 = wall();
        ok = grid_barrier(sync, ++epoch, tid, &s_ok, false);      // nobody overwrites before everybody has read
        tp += t1 - t0;
        tc += t3 - t2;
    }
    if (!ok && tid == 0) atomicAdd(errors + 1, 1u);
    if (bad) atomicAdd(errors, bad);
    if (blockIdx.x == 0 && tid == 0) { cyc[0] = tp; cyc[1] = tc; __hip_atomic_store(&sync[1], epoch, __ATOMIC_RELAXED, AGENT); }
}

__global__ __launch_bounds__(NT) void k_produce(float* buf, int rows, int it) { produce<0>(buf, rows, it, threadIdx.x); }
__global__ __launch_bounds__(NT) void k_consume(const float* buf, int rows, int it, unsigned* errors) {
    const unsigned bad = consume<0>(buf, rows, it, threadIdx.x, nullptr);
    if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    unsigned *sync, *err;
    unsigned long long* cyc;
    float* buf;
    hipMalloc(&sync, 64 * 1024);
    hipMalloc(&err, 8);
    hipMalloc(&cyc, 16);
    hipMalloc(&buf, (size_t)1152 * 5120 * 4);
    hipMemset(sync, 0, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int G = 240;                                        // <= 256 CUs, one workgroup per CU, like the 240-tile GEMM grids at one crop
    printf("persistent grid %d x %d threads, %d iterations, buffer ROWS x 1280 fp32, every workgroup consumes one 64-row panel (320 KiB)\n", G, NT, iters);
    for (int rows : {192, 768}) {                             // 0.98 MB (x / h at one crop) and 3.9 MB (the MLP hidden at one crop, as 768 x 1280)
        for (int v = 0; v <= 4; ++v) {
            hipMemset(err, 0, 8);
            hipMemset(cyc, 0, 16);
            hipMemset(buf, 0, (size_t)rows * COLS * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            switch (v) {
                case 0:
                    for (int it = 1; it <= iters; ++it) {
                        hipLaunchKernelGGL(k_produce, dim3(G), dim3(NT), 0, 0, buf, rows, it);
                        hipLaunchKernelGGL(k_consume, dim3(G), dim3(NT), 0, 0, buf, rows, it, err);
                    }
                    break;
                case 1: hipLaunchKernelGGL(persistent<1>, dim3(G), dim3(NT), 0, 0, sync, buf, rows, iters, err, cyc); break;
                case 2: hipLaunchKernelGGL(persistent<2>, dim3(G), dim3(NT), 0, 0, sync, buf, rows, iters, err, cyc); break;
                case 3: hipLaunchKernelGGL(persistent<3>, dim3(G), dim3(NT), 0, 0, sync, buf, rows, iters, err, cyc); break;
                default: hipLaunchKernelGGL(persistent<4>, dim3(G), dim3(NT), 0, 0, sync, buf, rows, iters, err, cyc); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned h_err[2];
            unsigned long long h_cyc[2];
            hipMemcpy(h_err, err, 8, hipMemcpyDeviceToHost);
            hipMemcpy(h_cyc, cyc, 16, hipMemcpyDeviceToHost);
            const char* names[] = {"two launches, plain", "fence wb/inv, plain", "sc1 dword atomics", "sc1 16 B st/ld", "sc1 16 B st + LDS-DMA sc1"};
            printf("rows %4d (%.2f MB)  v%d %-28s %8.2f us/iter   produce %6.2f us  consume %6.2f us (workgroup 0, 100 MHz clock)   wrong %u  timeouts %u\n",
                   rows, rows * COLS * 4 / 1e6, v, names[v], ms * 1e3 / iters, h_cyc[0] / (double)iters / 100.0, h_cyc[1] / (double)iters / 100.0,
                   h_err[0], h_err[1]);
            fflush(stdout);
        }
    }
    (void)clk_khz;
    return 0;
}
