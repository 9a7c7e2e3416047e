// Micro-benchmark 10 (round 4): what does the STORE PATTERN of the split3 producers cost?
// Context: every producer of a split3 operand ([R][K/8][3][8] bf16: per 8 k three adjacent 16-byte chunks h, m, l) writes a 128-byte line in
// pieces — LayerNorm and attention as 8-byte halves of every third chunk, fc1's epilogue as 16-byte chunks at a 48-byte stride per
// instruction — and they take 1.3-1.6x the time their byte counts predict (LayerNorm 2.4 vs 1.5 ms per step for 10 vs 8 bytes per
// element, attention 137 vs 102 us per launch).  Is it the partial-line store pattern?
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/store_patterns scripts/micro/store_patterns.hip && build_ab/store_patterns
// A wave owns one row of 7680 bytes (a 1280-wide split3 row) per iteration; all patterns write the same 94 MB.
//   full    : each store instruction writes 1024 contiguous bytes (lane l: 16 B at 16 l) — what a plane-wise layout [R][3][K] would allow
//   chunk48 : lane l owns 48 contiguous bytes (k-group l), three instructions write its chunks 0 / 1 / 2 (16 B at a 48-byte lane stride): fc1 / new attention
//   half8   : lanes 2 j, 2 j + 1 own the two 8-byte halves of the three chunks of k-group j: three 8-byte stores (LayerNorm, old attention)
//   f32     : the fp32 row (5120 bytes, 16 B per lane contiguous), for scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(char* out, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4 v = {(unsigned)row, (unsigned)lane, 3u, 4u};
    if (MODE == 0) {
        char* r = out + (size_t)row * 7680;
#pragma unroll
        for (int i = 0; i < 7; ++i) *reinterpret_cast<u32x4*>(r + i * 1024 + lane * 16) = v;
        if (lane < 32) *reinterpret_cast<u32x4*>(r + 7 * 1024 + lane * 16) = v;
    } else if (MODE == 1) {
        char* r = out + (size_t)row * 7680;
#pragma unroll
        for (int i = 0; i < 3; ++i) {            // k-groups lane, lane + 64, lane + 128 (< 160)
            const int g = i * 64 + lane;
            if (g < 160) {
                u32x4* o = reinterpret_cast<u32x4*>(r + g * 48);
                o[0] = v; o[1] = v; o[2] = v;
            }
        }
    } else if (MODE == 2) {
        char* r = out + (size_t)row * 7680;
#pragma unroll
        for (int i = 0; i < 5; ++i) {            // 4 consecutive elements per lane and pass: group (i 64 + lane) / 2, half lane & 1
            const int c = (i * 64 + lane) * 4;
            char* o = r + (c >> 3) * 48 + (lane & 1) * 8;
            const u32x2 h = {v.x, v.y};
            *reinterpret_cast<u32x2*>(o) = h;
            *reinterpret_cast<u32x2*>(o + 16) = h;
            *reinterpret_cast<u32x2*>(o + 32) = h;
        }
    } else {
        char* r = out + (size_t)row * 5120;
#pragma unroll
        for (int i = 0; i < 5; ++i) *reinterpret_cast<u32x4*>(r + i * 1024 + lane * 16) = v;
    }
}

template <int MODE>
void run(const char* name, char* buf, int rows, size_t bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3((rows + 3) / 4), dim3(256), 0, 0, buf, rows);
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3((rows + 3) / 4), dim3(256), 0, 0, buf, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("{\"pattern\": \"%s\", \"rows\": %d, \"bytes\": %zu, \"us\": %.2f, \"GB_per_s\": %.1f}\n", name, rows, bytes, ms / n * 1e3, bytes / (ms / n * 1e-3) * 1e-9);
}

int main() {
    const int rows = 12288;
    char* buf;
    hipMalloc(&buf, (size_t)rows * 7680);
    run<0>("full-line (1024 contiguous B per instruction)", buf, rows, (size_t)rows * 7680);
    run<1>("chunk48 (16 B per lane at a 48 B stride, x3)", buf, rows, (size_t)rows * 7680);
    run<2>("half8 (8 B halves of every third chunk, x3)", buf, rows, (size_t)rows * 7680);
    run<3>("fp32 row (5120 B, contiguous)", buf, rows, (size_t)rows * 5120);
    run<0>("full-line again", buf, rows, (size_t)rows * 7680);
    return 0;
}
