// Micro-benchmark 5: where does the ViT attention kernel spend the 29 us it takes beyond its 77 us MFMA floor at B = 64?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o build_ab/attn_timeline scripts/micro/attn_timeline.hip
//   build_ab/attn_timeline [out.csv]
// Includes the PRODUCT kernel source and instantiates its diagnostic variants:
//   (a) product kernel, median of 30 single launches (HIP events);
//   (b) timeline: wave 0 of every workgroup stamps the 100 MHz clock at each phase boundary and records the CU it ran on;
//       printed: mean / p90 duration of every phase for first-round and second-round workgroups, the spread of start times, and how
//       many CUs held two first-round workgroups whose TG_ID parities differ (the de-phasing experiment's assumption);
//   (c) de-phasing: the workgroup on the odd threadgroup slot of its CU starts X us late in the first round only (the second round
//       then stays de-phased by itself), X = 2 ... 14 us; every variant is checked bit for bit against the product kernel.
#include "../../tokenhmr_amd/csrc/attention.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

template <int DBG>
static float time_variant(const float* qkv, float* out, int B, AttnDbg dbg, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < reps + 3; ++i) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((vit_attention_kernel<3, 4, DBG>), dim3(B * NH), dim3(256), 0, 0, qkv, out, dbg);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const int B = 64;
    const size_t nq = (size_t)B * NTOK * QKV_LD, no = (size_t)B * NTOK * DIM;
    std::vector<float> hq(nq);
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t i = 0; i < nq; ++i) hq[i] = nd(rng) * ((i % QKV_LD) < DIM ? 0.1118f : 1.f);
    float *qkv, *out, *ref;
    unsigned long long* tl;
    CK(hipMalloc(&qkv, nq * 4));
    CK(hipMalloc(&out, no * 4));
    CK(hipMalloc(&ref, no * 4));
    CK(hipMalloc(&tl, (size_t)B * NH * 16 * 8));
    CK(hipMemcpy(qkv, hq.data(), nq * 4, hipMemcpyHostToDevice));
    const AttnDbg nodbg{nullptr, 0, 0};
    hipLaunchKernelGGL((vit_attention_kernel<3, 4, 0>), dim3(B * NH), dim3(256), 0, 0, qkv, ref, nodbg);
    CK(hipDeviceSynchronize());
    std::vector<float> href(no), hout(no);
    CK(hipMemcpy(href.data(), ref, no * 4, hipMemcpyDeviceToHost));
    auto same = [&]() {
        (void)hipMemcpy(hout.data(), out, no * 4, hipMemcpyDeviceToHost);
        return memcmp(hout.data(), href.data(), no * 4) == 0;
    };

    // (a) product
    for (int rep = 0; rep < 2; ++rep) printf("product kernel: %.1f us per launch (median of 30)\n", time_variant<0>(qkv, out, B, nodbg, 30));

    // (b) timeline
    AttnDbg d1{tl, 0, 0};
    printf("timeline build: %.1f us per launch (stamps cost)\n", time_variant<1>(qkv, out, B, d1, 10));
    printf("timeline output bit-identical: %d\n", (int)same());
    std::vector<unsigned long long> h((size_t)B * NH * 16);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    const int n = B * NH;
    unsigned long long tmin = ~0ull, tmax = 0, first_end = ~0ull;
    for (int i = 0; i < n; ++i) {
        tmin = std::min(tmin, h[i * 16]);
        tmax = std::max(tmax, h[i * 16 + 12]);
        first_end = std::min(first_end, h[i * 16 + 12]);
    }
    printf("kernel span by stamps: %.2f us; first workgroup ends at %.2f us\n", (tmax - tmin) / 100.0, (first_end - tmin) / 100.0);
    const char* names[12] = {"dephase-wait", "Q+K issue / K1 land wait", "S(keys 0-95)", "wait K2 + V1 issue", "S(keys 96-191)", "wait + V2 issue",
                             "softmax", "wait V1", "PV(keys 0-95)", "wait V2", "PV(keys 96-191)", "normalise + store issue"};
    for (int round = 0; round < 2; ++round) {
        std::vector<std::vector<double>> d(12);
        std::vector<double> st, tot;
        for (int i = 0; i < n; ++i) {
            const bool second = h[i * 16] >= first_end;
            if ((int)second != round) continue;
            for (int p = 0; p < 12; ++p) d[p].push_back((h[i * 16 + p + 1] - h[i * 16 + p]) / 100.0);
            st.push_back((h[i * 16] - tmin) / 100.0);
            tot.push_back((h[i * 16 + 12] - h[i * 16]) / 100.0);
        }
        if (st.empty()) continue;
        auto stat = [](std::vector<double> v) {
            std::sort(v.begin(), v.end());
            double s = 0;
            for (double x : v) s += x;
            return std::vector<double>{s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back()};
        };
        auto s0 = stat(st), s1 = stat(tot);
        printf("round %d: %zu workgroups; start mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us; lifetime mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us\n", round,
               st.size(), s0[0], s0[1], s0[2], s0[3], s0[4], s1[0], s1[1], s1[2], s1[3], s1[4]);
        for (int p = 0; p < 12; ++p) {
            auto s = stat(d[p]);
            printf("   %-28s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", names[p], s[0], s[1], s[2], s[3], s[4]);
        }
    }
    // co-residency: CU key = xcc | se/sh/cu bits of HW_ID; TG_ID = bits 16..19
    {
        std::map<unsigned, std::vector<int>> cu;
        for (int i = 0; i < n; ++i) {
            if (h[i * 16] >= first_end) continue;
            const unsigned hw = (unsigned)h[i * 16 + 13], xcc = (unsigned)h[i * 16 + 14];
            cu[((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF)].push_back(i);
        }
        int pairs = 0, differ = 0, consecutive = 0;
        std::map<int, int> delta;
        for (auto& kv : cu) {
            if (kv.second.size() != 2) continue;
            ++pairs;
            const int a = kv.second[0], b = kv.second[1];
            const unsigned ta = ((unsigned)h[a * 16 + 13] >> 16) & 0xF, tb = ((unsigned)h[b * 16 + 13] >> 16) & 0xF;
            differ += (ta & 1) != (tb & 1);
            delta[std::abs(a - b)]++;
        }
        printf("first round: %zu CUs, %d with exactly two workgroups, %d of those with different TG_ID parity\n", cu.size(), pairs, differ);
        printf("   block-index distance of co-resident pairs:");
        for (auto& kv : delta) printf(" %d:%d", kv.first, kv.second);
        printf("\n");
        (void)consecutive;
    }
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "block,hw_id,xcc,t0..t12 (10 ns ticks from kernel start)\n");
            for (int i = 0; i < n; ++i) {
                fprintf(f, "%d,%llu,%llu", i, h[i * 16 + 13], h[i * 16 + 14]);
                for (int p = 0; p < 13; ++p) fprintf(f, ",%llu", h[i * 16 + p] - tmin);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }

    // (d) block order and the persistent form, interleaved: old order (heads of a crop spread over the XCDs) / XCD-aware order / persistent
    for (int rep = 0; rep < 3; ++rep) {
        const float t_old = time_variant<4>(qkv, out, B, nodbg, 12);
        const bool ok_old = same();
        const float t_new = time_variant<0>(qkv, out, B, nodbg, 12);
        const bool ok_new = same();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        std::vector<float> ts;
        (void)hipMemset(out, 0, no * 4);
        for (int i = 0; i < 15; ++i) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((vit_attention_persistent_kernel<0>), dim3(512), dim3(256), 0, 0, qkv, out, B * NH, nodbg);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        const bool ok_p = same();
        printf("round-1 block order %.1f us (bit-identical %d) | XCD-aware order %.1f us (%d) | persistent %.1f us (%d)\n", t_old, (int)ok_old, t_new, (int)ok_new,
               ts[ts.size() / 2], (int)ok_p);
    }
    // (e) persistent kernel with the odd-slot workgroup of every CU starting late
    for (int rep = 0; rep < 2; ++rep)
        for (int dly : {0, 4, 8, 10, 12, 16, 20}) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            std::vector<float> ts;
            const AttnDbg dd{nullptr, dly * 100, 0};
            for (int i = 0; i < 15; ++i) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL((vit_attention_persistent_kernel<0>), dim3(512), dim3(256), 0, 0, qkv, out, B * NH, dd);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (i >= 3) ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("persistent, odd slot starts %2d us late: %.1f us per launch (bit-identical %d)\n", dly, ts[ts.size() / 2], (int)same());
        }
    // (f) persistent kernel with s_setprio(1) around its MFMA phases (the two workgroups of a CU are in different phases)
    for (int rep = 0; rep < 3; ++rep) {
        float t[2];
        for (int v = 0; v < 2; ++v) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            std::vector<float> ts;
            for (int i = 0; i < 15; ++i) {
                (void)hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL((vit_attention_persistent_kernel<0>), dim3(512), dim3(256), 0, 0, qkv, out, B * NH, nodbg);
                else hipLaunchKernelGGL((vit_attention_persistent_kernel<8>), dim3(512), dim3(256), 0, 0, qkv, out, B * NH, nodbg);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (i >= 3) ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            t[v] = ts[ts.size() / 2];
        }
        printf("persistent: plain %.1f us | s_setprio(1) in the MFMA phases %.1f us (bit-identical %d)\n", t[0], t[1], (int)same());
    }
    {   // timeline of the persistent kernel's FIRST items
        AttnDbg d1p{tl, 0, 0};
        hipLaunchKernelGGL((vit_attention_persistent_kernel<1>), dim3(512), dim3(256), 0, 0, qkv, out, B * NH, d1p);
        (void)hipDeviceSynchronize();
        printf("persistent timeline build bit-identical: %d\n", (int)same());
        std::vector<unsigned long long> hp((size_t)512 * 16);
        (void)hipMemcpy(hp.data(), tl, hp.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 512; ++i) t0 = std::min(t0, hp[i * 16]);
        double acc[13] = {0};
        for (int i = 0; i < 512; ++i)
            for (int p2 = 0; p2 < 12; ++p2) acc[p2] += (hp[i * 16 + p2 + 1] - hp[i * 16 + p2]) / 100.0 / 512;
        printf("persistent, first item of each workgroup, mean phase durations (us):");
        for (int p2 = 1; p2 < 12; ++p2) printf(" %s=%.2f", names[p2], acc[p2]);
        printf("\n");
    }

    // (c) de-phasing by the CU's threadgroup slot, first round only; interleaved with the product kernel
    const int delays_us[] = {0, 2, 4, 6, 8, 10, 12, 14};
    for (int rep = 0; rep < 2; ++rep) {
        for (int dly : delays_us) {
            AttnDbg d2{nullptr, dly * 100, 512};
            const float t = time_variant<2>(qkv, out, B, d2, 12);
            const bool ok = same();
            printf("dephase (TG parity, first 512 workgroups) %2d us: %.1f us per launch  bit-identical %d\n", dly, t, (int)ok);
        }
        printf("product kernel: %.1f us per launch\n", time_variant<0>(qkv, out, B, nodbg, 12));
    }
    // the same delay applied in BOTH rounds (every odd-slot workgroup)
    for (int dly : {4, 8}) {
        AttnDbg d2{nullptr, dly * 100, 1 << 30};
        printf("dephase (TG parity, all workgroups) %2d us: %.1f us per launch\n", dly, time_variant<2>(qkv, out, B, d2, 12));
    }
    return 0;
}
