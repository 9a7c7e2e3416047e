// Micro-benchmark 11 (round 4): is v_cvt_pk_bf16_f32 bit-identical to the integer round-to-nearest-even of csrc/common.h (bf16_rne)?
// Every high-half pattern (all signs / exponents / the top 7 mantissa bits) x 8 low halves around the rounding boundary = 524,288 fp32
// inputs incl. zeros, denormals, infinities and NaNs; mismatches are counted per class.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o build_ab/bf16_cvt_classes scripts/micro/bf16_cvt_classes.hip && build_ab/bf16_cvt_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ void k(const uint32_t* in, uint32_t* sw, uint32_t* hw, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(in[i]);
    sw[i] = bf16_rne(x);
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    hw[i] = r & 0xffffu;
}

int main() {
    const uint32_t lows[8] = {0x0000, 0x0001, 0x7fff, 0x8000, 0x8001, 0xffff, 0x4000, 0xc000};
    std::vector<uint32_t> h;
    for (uint32_t hi = 0; hi < 65536; ++hi)
        for (uint32_t lo : lows) h.push_back((hi << 16) | lo);
    const int n = (int)h.size();
    uint32_t *din, *dsw, *dhw;
    hipMalloc(&din, n * 4); hipMalloc(&dsw, n * 4); hipMalloc(&dhw, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, din, dsw, dhw, n);
    std::vector<uint32_t> sw(n), hw(n);
    hipMemcpy(sw.data(), dsw, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hw.data(), dhw, n * 4, hipMemcpyDeviceToHost);
    long mis[5] = {0, 0, 0, 0, 0}, cnt[5] = {0, 0, 0, 0, 0};       // zero, denormal, normal, inf, nan
    uint32_t ex[5][3] = {};
    for (int i = 0; i < n; ++i) {
        const uint32_t u = h[i], e = (u >> 23) & 0xff, m = u & 0x7fffff;
        const int c = e == 0 ? (m == 0 ? 0 : 1) : e == 255 ? (m == 0 ? 3 : 4) : 2;
        ++cnt[c];
        if (sw[i] != hw[i]) { if (!mis[c]) { ex[c][0] = u; ex[c][1] = sw[i]; ex[c][2] = hw[i]; } ++mis[c]; }
    }
    const char* names[5] = {"zero", "denormal", "normal", "inf", "nan"};
    for (int c = 0; c < 5; ++c)
        printf("{\"class\": \"%s\", \"inputs\": %ld, \"mismatches\": %ld, \"first\": {\"in\": \"0x%08x\", \"integer_rne\": \"0x%04x\", \"v_cvt_pk_bf16_f32\": \"0x%04x\"}}\n",
               names[c], cnt[c], mis[c], ex[c][0], ex[c][1], ex[c][2]);
    return 0;
}
