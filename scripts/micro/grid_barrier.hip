// Micro-benchmark 4: what does a grid-wide barrier cost on this 8-XCD part, and which piece of it?
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/grid_barrier scripts/micro/grid_barrier.hip && build_ab/grid_barrier
// NB workgroups of NT threads run ITERS barriers back to back; between barriers every workgroup writes one 4 KB slice of a
// shared buffer and, after the barrier, checks the slice of its neighbour (so a barrier that does not make data visible across
// XCDs is caught).  Variants:
//   0  counter, device-scope release fence (L2 write-back) + relaxed poll + acquire fence (L2 invalidate), one thread / workgroup
//   1  counter, NO cache maintenance: data moves with device-scope (sc1) atomic stores / loads, the barrier only orders
//   2  flags: every workgroup writes its own flag, workgroup 0 polls all flags in parallel and publishes the epoch (no RMW)
//   3  like 2 with the data protocol of 1 (no cache maintenance)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define AGENT __HIP_MEMORY_SCOPE_AGENT

template <int VARIANT>
__global__ __launch_bounds__(1024) void k(unsigned* sync, float* buf, int iters, unsigned* errors) {
    __shared__ int s_dummy;
    const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    constexpr bool FENCE = VARIANT == 0 || VARIANT == 2;
    constexpr bool FLAGS = VARIANT >= 2;
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        // produce: slice b of the buffer <- it
        float* mine = buf + b * 1024;
        if (FENCE) mine[tid] = (float)it;
        else __hip_atomic_store(mine + tid, (float)it, __ATOMIC_RELAXED, AGENT);
        // ---- barrier ----
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (!FLAGS) {
            if (tid == 0) {
                if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, AGENT);
                const unsigned target = (unsigned)it * nb;
                while ((int)(__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
                if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (tid == 0) {
                if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(&sync[64 + b * 32], (unsigned)it, __ATOMIC_RELAXED, AGENT);      // own 128-byte line
            }
            if (b == 0) {
                if (tid < nb) while (__hip_atomic_load(&sync[64 + tid * 32], __ATOMIC_RELAXED, AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&sync[0], (unsigned)it, __ATOMIC_RELAXED, AGENT);
            }
            if (tid == 0) {
                while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
                if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        // consume: the neighbour's slice must hold `it`
        const float* other = buf + ((b + 1) % nb) * 1024;
        const float v = FENCE ? other[tid] : __hip_atomic_load(other + tid, __ATOMIC_RELAXED, AGENT);
        if (v != (float)it) ++bad;
        // second barrier-free hazard: the neighbour must not overwrite before we read -> one more barrier would be needed in
        // real code; here the producer of iteration it+1 writes it+1 only after ITS consume of `it`, and a stale read of `it+1`
        // would be flagged as != it only if it races — accept (count separately)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
    (void)s_dummy;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    unsigned *sync, *err;
    float* buf;
    hipMalloc(&sync, 64 * 1024);
    hipMalloc(&err, 4);
    hipMalloc(&buf, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int nb : {8, 16, 32, 64, 128}) {
        for (int v = 0; v < 4; ++v) {
            float best = 1e9f;
            unsigned herr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(sync, 0, 64 * 1024);
                hipMemset(err, 0, 4);
                hipMemset(buf, 0, 256 * 1024 * 4);
                hipEventRecord(e0, 0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(1024), 0, 0, sync, buf, iters, err);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(1024), 0, 0, sync, buf, iters, err);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(1024), 0, 0, sync, buf, iters, err);
                if (v == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(1024), 0, 0, sync, buf, iters, err);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
                unsigned h = 0;
                hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
                herr += h;
            }
            printf("blocks %3d variant %d: %.2f us per barrier (+ produce/consume), data errors %u\n", nb, v, best * 1e3f / iters, herr);
        }
    }
    return 0;
}
