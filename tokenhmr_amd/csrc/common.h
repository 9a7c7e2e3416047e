// Shared declarations for the gfx950 TokenHMR kernels.  CDNA4 only: wave = 64 lanes,
// fp32-input MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, exact fp32 == fmaf chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- A/B knobs and measured-slower variants: the EXPERIMENTS build only ----
// The shipped library (libtokenhmr_hip.so) reads no environment variable and carries no kernel that lost its A/B: every THMR_* knob
// below resolves to its default at compile time, and the variants they select are compiled out.  -DTHMR_EXPERIMENTS
// (lib/libtokenhmr_hip_exp.so, built beside it by __graft_entry__.build(); tokenhmr_amd._cabi.load(exp=True) or THMR_LIB=exp) keeps
// the knobs, the debug hooks (forced decoder timeout, barrier variants, timelines) and those kernels for the tests and scripts that
// measure them.  The documented user-facing switch, THMR_VIT_GEMM, is read by the Python facade (tokenhmr_amd/model.py), not here.
#include <cstdlib>
#ifdef THMR_EXPERIMENTS
inline const char* thmr_knob(const char* name) { return getenv(name); }
#else
inline const char* thmr_knob(const char*) { return nullptr; }
#endif

#define THMR_WAVE 64

// ---- epilogue ids of the tiled / skinny GEMMs (also exposed through thmr_op_gemm) ----
enum GemmEpi {
    EPI_NONE = 0,        // C = acc
    EPI_BIAS = 1,        // C = acc + bias[n]
    EPI_BIAS_GELU = 2,   // C = gelu_erf(acc + bias[n])
    EPI_BIAS_RELU = 3,   // C = max(acc + bias[n], 0)
    EPI_BIAS_RESID = 4,  // C = resid[m][n] + (acc + bias[n])
    EPI_BIAS_QSCALE = 5, // C = (acc + bias[n]) * (n < qcols ? qscale : 1)
    EPI_BIAS_POS = 6,    // C = ((acc + bias[n]) + pos[1 + m % 192][n]) + pos[0][n]   (vit.py:327)
    EPI_NUM = 7
};

struct GemmArgs {
    const float* A;      // [M][lda]  K-contiguous
    const float* W;      // [N][ldw]  K-contiguous (torch Linear weight layout)
    const float* bias;   // [N] or null
    const float* resid;  // [M][ldr] (EPI_BIAS_RESID) or pos_embed [193][N] (EPI_BIAS_POS)
    float* C;            // [M][ldc]
    int64_t lda, ldw, ldc, ldr;
    int M, N, K;
    float qscale;
    int qcols;
    // optional: ALSO write the result as the im2col operand of the Conv1d(k = 3, padding = dilation) that consumes it, so the
    // VQ decoder needs no gather launches (vanilla_pose_vqvae.py:135-154: nearest resample -> Conv1d; resnet.py:55-68).
    //   rows of C are (crop b, position ts), ts < cs_tin; the consumer works on cs_tout resampled positions tp, tp = cs_inv[ts]
    //   (-1: this position is dropped by the resample; cs_inv == nullptr: identity); element (b, tp, n) is tap dk of gathered
    //   row t = tp - (dk - 1) * cs_dil:  cs_out[(b * cs_tout + t) * 3N + dk * N + n] = f(v), f = ReLU if cs_relu (pre-activation
    //   ResConv1DBlock); the taps that fall outside [0, cs_tout) are written as zeros by the row that owns them.
    //   C may be nullptr when only the gathered copy is needed.
    float* cs_out;
    const int32_t* cs_inv;
    int cs_tin, cs_tout, cs_dil, cs_relu;
    // split-K of the big-tile kernel (gemm_f32_kernel, launch_gemm_splitk): ksplit > 1 = the launch holds ksplit copies of the tile
    // grid, copy sp reduces K slice [sp*K/ksplit, (sp+1)*K/ksplit) and stores its RAW partial tile to C + sp*M*ldc (C = the
    // partial-sum buffer part[ksplit][M][N], no epilogue); 0 / 1 = off.
    int ksplit;
    // split3 GEMM only (gemm_split.hip): c_split != nullptr = the epilogue's result goes out as a split3 operand [M][N/8][3][8] bf16
    // (row stride 6 * ldcs bytes; N % 8 == 0) INSTEAD of fp32 C — the next GEMM's A operand without a conversion pass
    void* c_split;
    int64_t ldcs;
    // split3 GEMM only: the ROW-BLOCKED form of a split3 operand, [rows / 32][K / 8][3][32][8] bf16 — the three 16-byte chunks of a k-group
    // as 32-row panels of 512 contiguous bytes: chunk (r, kg, piece) at (r / 32) K 192 + (kg 3 + piece) 512 + (r % 32) 16 bytes.
    //   a_blk:  A is in this form (lda = its K in fp32-equivalents; rows padded to a multiple of 32).  A K tile of a 32-row block is 6 KB
    //           contiguous in memory AND in the LDS image, so the LDS-DMA copies are linear on both sides and the fragment reads
    //           conflict-free without a swizzle.
    //   cs_blk: c_split is written in this form.  The swapped-role epilogue of the persistent kernel holds one output ROW per lane: in the
    //           row-major form its 16-byte stores hit 64 different lines per instruction (1536 line visits per wave tile against 128 for an
    //           fp32 tile: 7.5 of the 9.9 us fc1's epilogue costs per tile, profiles/r4e_split3_gemm_b64.jsonl); here 32 lanes write 512
    //           contiguous bytes.  Used for the one operand that only GEMM kernels touch: fc1's GELU output = fc2's A (vit.py:84-87).
    int a_blk, cs_blk;
    // split3 GEMM, A/B only (experiments build: the engine sets it from THMR_SPLIT3_NARROW8=1 / THMR_SPLIT3_TAIL8=1 at thmr_create): bit 0 = the
    // 128 x 128 tile on eight waves of 64 x 32 instead of four of 64 x 64, bit 1 = the half-tile tail likewise (round 6: measured slower /
    // equal, gemm_split.hip); bit 2 = the 128 x 128 tile with round 5's TWO-stage K ring instead of three stages (THMR_SPLIT3_RING3=0);
    // bit 3 = small 128 x 256 grids keep their copies spread over the K tile (THMR_SPLIT3_FRONT=0).  Same bits either way.
    int tile_opts;
};

// byte offset of chunk (row m, k-group n8, piece pc) of a split3 operand with row length ld (fp32-equivalents), row-major or row-blocked
__device__ __forceinline__ int64_t split3_chunk_off(int64_t ld, int m, int n8, int pc, bool blk) {
    return blk ? (int64_t)(m >> 5) * ld * 192 + (int64_t)(n8 * 3 + pc) * 512 + (m & 31) * 16 : (int64_t)m * ld * 6 + (int64_t)n8 * 48 + pc * 16;
}

// Branch-free fp32 erf, < 1.5 ulp over the whole line (tests/test_gpu_ops.py::test_gelu_epilogue_ulp): two minimax
// pieces evaluated for every lane and selected, so the GEMM epilogue has no divergent paths.  |x| <= 0.921875 uses an odd
// polynomial x + x*P(x^2); above it erf = 1 - exp(-Q(|x|)) with one hardware exp2 (erf(x) == 1.0f in fp32 from |x| = 3.92,
// so |x| is clamped at 4, which also keeps inf finite).  ocml's erff costs ~34 VALU + divergent branches per element, this
// costs ~22 straight-line VALU; fc1's epilogue applies it to 80 elements per lane per tile.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fminf(fabsf(a), 4.0f);
    const float s = t * t;
    float r = fmaf(0x1.222900p-16f, t, -0x1.91d2ccp-12f);
    const float u = fmaf(0x1.fd1336p-09f, t, -0x1.8d6300p-06f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, 0x1.b55cb0p-4f);
    r = fmaf(r, t, 0x1.450aa0p-1f);
    r = fmaf(r, t, 0x1.079d0cp-3f);
    r = fmaf(r, t, t);
    const float big = 1.0f - __builtin_amdgcn_exp2f(r * -1.44269504088896340736f);
    float q = -0x1.3a1a82p-11f;
    q = fmaf(q, s, 0x1.473f48p-08f);
    q = fmaf(q, s, -0x1.b68bd2p-06f);
    q = fmaf(q, s, 0x1.ce1a46p-04f);
    q = fmaf(q, s, -0x1.8126e0p-02f);
    q = fmaf(q, s, 0x1.06eba6p-03f);
    const float small = fmaf(q, t, t);
    return copysignf(t > 0.921875f ? big : small, a);
}

// Single-piece erf for the GELU epilogue: erf(|a|) = 1 - 2^(|a| Q(|a|)) with |a| Q(|a|) ~ log2(erfc(|a|)) on [0, 4], Q of
// degree 7 (weighted minimax fit, |erf error| 1.6e-8 in exact arithmetic, ~9e-8 = under one ulp of 1.0 evaluated in fp32).
// Unlike erf_fast it is NOT relatively accurate near 0 — GELU does not need that: 0.5 x (1 + erf) only sees erf's ABSOLUTE
// error, and measured against fp64 GELU this form is 2.5x closer than torch's own fp32 GELU (4.3e-7 vs 1.1e-6 max abs error on
// [-9, 9]; tests/test_gpu_ops.py::test_gelu_epilogue_ulp).  13 VALU per element instead of 22: fc1's epilogue is pure VALU
// time on the matrix pipe (profiles/r1_mfma_valu_microbench.log).
__device__ __forceinline__ float erf_gelu(float a) {
    const float t = fminf(fabsf(a), 4.0f);
    float q = -0x1.7c7e70p-15f;
    q = fmaf(q, t, 0x1.d325a2p-12f);
    q = fmaf(q, t, -0x1.867268p-10f);
    q = fmaf(q, t, -0x1.962214p-11f);
    q = fmaf(q, t, 0x1.cee88ep-6f);
    q = fmaf(q, t, -0x1.301722p-3f);
    q = fmaf(q, t, -0x1.d63aacp-1f);
    q = fmaf(q, t, -0x1.a0be9ep+0f);
    return copysignf(1.0f - __builtin_amdgcn_exp2f(q * t), a);
}

#ifndef THMR_GELU_IMPL
#define THMR_GELU_IMPL 3      // 0 = ocml erff (A/B only), 1 = erf_fast per element, 2 = erf_fast on element pairs, 3 = erf_gelu on pairs
#endif

__device__ __forceinline__ float gelu_erf(float x) {
    // torch.nn.GELU() default (approximate='none'): 0.5*x*(1+erf(x/sqrt(2)))
#if THMR_GELU_IMPL == 0
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#elif THMR_GELU_IMPL == 3
    return 0.5f * x * (1.0f + erf_gelu(x * 0.70710678118654752440f));
#else
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
#endif
}

// The same arithmetic on two elements at once: every fma/mul/add is a packed-fp32 VALU op (v_pk_fma_f32 & co. run two fp32
// lanes per instruction on gfx950), bit-identical per element to gelu_erf above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#if THMR_GELU_IMPL < 2
    return f32x2{gelu_erf(x.x), gelu_erf(x.y)};
#elif THMR_GELU_IMPL == 3
    const f32x2 a = x * splat2(0.70710678118654752440f);
    f32x2 t = __builtin_elementwise_abs(a);
    t = f32x2{fminf(t.x, 4.0f), fminf(t.y, 4.0f)};
    f32x2 q = splat2(-0x1.7c7e70p-15f);
    q = __builtin_elementwise_fma(q, t, splat2(0x1.d325a2p-12f));
    q = __builtin_elementwise_fma(q, t, splat2(-0x1.867268p-10f));
    q = __builtin_elementwise_fma(q, t, splat2(-0x1.962214p-11f));
    q = __builtin_elementwise_fma(q, t, splat2(0x1.cee88ep-6f));
    q = __builtin_elementwise_fma(q, t, splat2(-0x1.301722p-3f));
    q = __builtin_elementwise_fma(q, t, splat2(-0x1.d63aacp-1f));
    q = __builtin_elementwise_fma(q, t, splat2(-0x1.a0be9ep+0f));
    const f32x2 r = q * t;
    const f32x2 m = splat2(1.0f) - f32x2{__builtin_amdgcn_exp2f(r.x), __builtin_amdgcn_exp2f(r.y)};
    const f32x2 e = f32x2{copysignf(m.x, a.x), copysignf(m.y, a.y)};
    return (splat2(0.5f) * x) * (splat2(1.0f) + e);
#else
    const f32x2 a = x * splat2(0.70710678118654752440f);
    f32x2 t = __builtin_elementwise_abs(a);
    t = f32x2{fminf(t.x, 4.0f), fminf(t.y, 4.0f)};
    const f32x2 s = t * t;
    f32x2 r = __builtin_elementwise_fma(splat2(0x1.222900p-16f), t, splat2(-0x1.91d2ccp-12f));
    const f32x2 u = __builtin_elementwise_fma(splat2(0x1.fd1336p-09f), t, splat2(-0x1.8d6300p-06f));
    r = __builtin_elementwise_fma(r, s, u);
    r = __builtin_elementwise_fma(r, t, splat2(0x1.b55cb0p-4f));
    r = __builtin_elementwise_fma(r, t, splat2(0x1.450aa0p-1f));
    r = __builtin_elementwise_fma(r, t, splat2(0x1.079d0cp-3f));
    r = __builtin_elementwise_fma(r, t, t);
    r = r * splat2(-1.44269504088896340736f);
    const f32x2 big = splat2(1.0f) - f32x2{__builtin_amdgcn_exp2f(r.x), __builtin_amdgcn_exp2f(r.y)};
    f32x2 q = splat2(-0x1.3a1a82p-11f);
    q = __builtin_elementwise_fma(q, s, splat2(0x1.473f48p-08f));
    q = __builtin_elementwise_fma(q, s, splat2(-0x1.b68bd2p-06f));
    q = __builtin_elementwise_fma(q, s, splat2(0x1.ce1a46p-04f));
    q = __builtin_elementwise_fma(q, s, splat2(-0x1.8126e0p-02f));
    q = __builtin_elementwise_fma(q, s, splat2(0x1.06eba6p-03f));
    const f32x2 small = __builtin_elementwise_fma(q, t, t);
    const f32x2 e = f32x2{copysignf(t.x > 0.921875f ? big.x : small.x, a.x), copysignf(t.y > 0.921875f ? big.y : small.y, a.y)};
    return (splat2(0.5f) * x) * (splat2(1.0f) + e);
#endif
}

template <int EPI>
__device__ __forceinline__ float gemm_epilogue(const GemmArgs& a, float acc, float bias, int m, int n) {
    float v = acc;
    if constexpr (EPI != EPI_NONE) v = acc + bias;
    if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
    if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
    if constexpr (EPI == EPI_BIAS_RESID) v = a.resid[(int64_t)m * a.ldr + n] + v;
    if constexpr (EPI == EPI_BIAS_QSCALE) v = (n < a.qcols) ? v * a.qscale : v;
    if constexpr (EPI == EPI_BIAS_POS) {
        const int t = m % 192;
        v = (v + a.resid[(int64_t)(1 + t) * a.N + n]) + a.resid[n];
    }
    return v;
}

// ---- "split3": an fp32 value as three bf16 pieces h + m + l (gemm_split.hip) ----
// Round-to-nearest-even bf16 of x (any NaN -> 0x7fc0) in integer arithmetic: the form of the stand-alone converter (split3_kernel: weights
// at load time, ops.split3) and the one the numpy restatement in the tests states.  The producers inside the hot path use split3_pair
// below (hardware conversion), which differs from it on NaN payloads only.
// Round 4: the pieces are kept in the HIGH half of a register — (u + 0x7fff + lsb) & 0xffff0000 IS the fp32 value of the rounded piece,
// so the exact residual needs no shift back, one NaN test serves the three pieces, and two values pack with ONE v_perm_b32: ~15 VALU
// per value instead of ~24.  In the persistent fc1 kernel the epilogue's VALU is what is left exposed per tile (PMC: 64.6 M vector
// instructions per fc1 launch against 26.5 M for qkv, profiles/r4p_pmc_split3_persistent.json), at the ~1.5 GHz the part grants it.
__device__ __forceinline__ uint32_t bf16_rne(float x) {        // the 16-bit pattern (low half); kept for single values
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// the three pieces of x, each in the HIGH half of its word (low half zero); bit-identical to bf16_rne applied to x, x - h, x - h - m
__device__ __forceinline__ void split3_hi(float x, uint32_t& H, uint32_t& M, uint32_t& L) {
    const uint32_t u = __float_as_uint(x);
    const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
    H = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    const float r1 = x - __uint_as_float(H);                  // exact: the residual of a rounding fits the format
    const uint32_t u1 = __float_as_uint(r1);
    M = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(M);                 // exact
    const uint32_t u2 = __float_as_uint(r2);
    L = (u2 + 0x7fffu + ((u2 >> 16) & 1u)) & 0xffff0000u;
    if (nan) H = M = L = 0x7fc00000u;                         // what bf16_rne gives for x and for the (NaN) residuals
}
__device__ __forceinline__ void split3_of(float x, uint32_t& h, uint32_t& m, uint32_t& l) {      // 16-bit patterns in the low half
    uint32_t H, M, L;
    split3_hi(x, H, M, L);
    h = H >> 16; m = M >> 16; l = L >> 16;
}
// two values -> three words of the operand format: piece(x0) in the low half, piece(x1) in the high half.  v_cvt_pk_bf16_f32 rounds the
// pair and packs it in ONE instruction, a shift / a mask give the fp32 value of each piece back, the residuals are one packed subtract:
// 9 VALU per pair against ~30 for the integer form above.  Bit-identical to it on every zero, denormal, normal and infinity
// (scripts/micro/bf16_cvt_classes.hip, profiles/r4e_bf16_cvt_classes.jsonl: 524,288 patterns around every rounding boundary); a NaN
// stays a NaN in all three pieces but keeps payload bits where the integer form writes 0x7fc0.  (Round 4, second pass: with whole-line
// stores in the row-blocked operand the fc1 epilogue IS bound by its instruction count — 2100 VALU per wave and tile, two waves per SIMD
// at ~1.45 GHz = 11.6 us of the ~99 us a tile takes; the first attempt at this, profiles/r3z_hw_bf16_cvt_ab.log, was store-bound.)
// (The conversion goes through the compiler — fptrunc <2 x float> to <2 x bfloat> — not through inline asm: gfx950 needs a wait state
// between a packed-fp32 VALU result and its use by the next VALU instruction, which hipcc inserts (s_nop 0) only around instructions it
// can see.  The asm form read stale registers right behind the GELU's v_pk_mul_f32: profiles/r4m_pytest_ops_inline_asm_hazard.log.
// Contraction is off in here: r = x - h must subtract from the ROUNDED x the pieces' consumers see, not from the product that made it.)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& H, uint32_t& M, uint32_t& L) {
#pragma clang fp contract(off)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    H = cvt_pk_bf16(x0, x1);
    f32x2_t r = f32x2_t{x0, x1} - f32x2_t{__uint_as_float(H << 16), __uint_as_float(H & 0xffff0000u)};      // exact
    M = cvt_pk_bf16(r.x, r.y);
    r = r - f32x2_t{__uint_as_float(M << 16), __uint_as_float(M & 0xffff0000u)};                            // exact
    L = cvt_pk_bf16(r.x, r.y);
}
// 4 consecutive columns c ... c + 3 (c % 4 == 0) of one row of a split3 operand [rows][D/8][3][8]: the 8-byte half (c & 4) of the three
// chunks of k-group c / 8.  `row` points at the row's first byte.
__device__ __forceinline__ void store_split3_quad(char* row, int c, f32x4 v) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint32_t H[2], M[2], L[2];
    split3_pair(v[0], v[1], H[0], M[0], L[0]);
    split3_pair(v[2], v[3], H[1], M[1], L[1]);
    char* o = row + (c >> 3) * 48 + (c & 4) * 2;
    *reinterpret_cast<u32x2*>(o) = u32x2{H[0], H[1]};
    *reinterpret_cast<u32x2*>(o + 16) = u32x2{M[0], M[1]};
    *reinterpret_cast<u32x2*>(o + 32) = u32x2{L[0], L[1]};
}

// 8 consecutive columns c ... c + 7 (c % 8 == 0) of one row: the three whole 16-byte chunks of k-group c / 8 (48 contiguous bytes)
__device__ __forceinline__ void store_split3_oct(char* row, int c, f32x4 lo, f32x4 hi) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    uint32_t H[4], M[4], L[4];
    split3_pair(lo[0], lo[1], H[0], M[0], L[0]);
    split3_pair(lo[2], lo[3], H[1], M[1], L[1]);
    split3_pair(hi[0], hi[1], H[2], M[2], L[2]);
    split3_pair(hi[2], hi[3], H[3], M[3], L[3]);
    u32x4_t* o = reinterpret_cast<u32x4_t*>(row + (c >> 3) * 48);
    o[0] = u32x4_t{H[0], H[1], H[2], H[3]};
    o[1] = u32x4_t{M[0], M[1], M[2], M[3]};
    o[2] = u32x4_t{L[0], L[1], L[2], L[3]};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- device-scope (sc1) relaxed atomics: data that crosses workgroups INSIDE a kernel without cache maintenance ----
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes of a device-coherent activation row as two 8-byte device-scope loads
__device__ __forceinline__ f32x4 ld_dev4(const float* p) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    union { unsigned long long u; f32x2_t f; } a, b;
    a.u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{a.f.x, a.f.y, b.f.x, b.f.y};
}

// ---- parameters of the MLP-Mixer stack (mixer_fused.hip; also the distributed tail of the persistent decoder kernel) ----
struct MixerLayerW {
    const float *ln1w, *ln1b, *wt1, *bt1, *wt2, *bt2, *ln2w, *ln2b, *wc1, *bc1, *wc2, *bc2;
};
struct MixerParams {
    MixerLayerW L[4];
    const float *tln_w, *tln_b;          // mixer_trans.ff.1 LayerNorm(10240)
    const float *wn, *bn, *nln_w, *nln_b;   // mixer_norm_layer: Linear(64,64) + LayerNorm(64) + ReLU
    const float* mt;                     // (B, 10240) mixer_trans Linear output
    float* out;                          // (B, 160, 64) classifier features -> class_pred_layer
};


// ---- the persistent decoder kernel (decoder_fused.hip) ----
struct DecLayerW {        // one decoder layer (pose_transformer.py:191-201), pointers into the weight arena
    const float *n0w, *n0b, *wv, *wo1, *bo1;          // PreNorm + self-attention (v slice of to_qkv, to_out)
    const float *n1w, *n1b, *wq, *wo2, *bo2;          // PreNorm + cross-attention (to_q, to_out); K/V come from the KV GEMM
    const float *n2w, *n2b, *w1, *b1, *w2, *b2;       // PreNorm + FeedForward
};
struct DecParams {
    DecLayerW L[6];
    const float *tok_bias, *pos;                      // layer-0 input = bias + pos_embedding (the input token is zero)
    const float* kv;                                  // (B*192, ldkv): per layer [K (512) | V (512)]
    const float *ro_w, *ro_b, *mt_w, *mt_b;           // read-outs (31 rows + zero row), mixer_trans.ff.0 (10240 rows)
    float *dx, *dv, *dq, *dca, *dff, *ro, *mt;        // scratch: (B,1024) (B,512) (B,512) (B,512) (B,1024) (B,32) (B,10240)
    unsigned* sync;
    unsigned* host_err;                               // host-mapped sticky error word (null = none), see GridSync
    int64_t ldkv;
    int depth, B;
    int timeline;                                     // 1: workgroup 0 stamps the wall clock into sync[16..] (diagnostics)
    int debug_fail;                                   // 1 (THMR_DEC_FORCE_TIMEOUT=1, tests only): report a barrier timeout although none happened
    int barrier_a2a;                                  // 1: all-to-all grid barrier (THMR_DEC_BARRIER=1, A/B only: measured slower than the two-hop form)
    int max_blocks;                                   // compute units of the device: at most one workgroup per CU is launched
    // Distributed MLP-Mixer tail (decoder_fused.hip mixer_cluster_stage): with mixer_cluster = 10 / 5 / 2 the kernel also runs the
    // mixer stack, that many workgroups per crop (1 / 2 / 5 of the ten 16-token tiles each), and the separate mixer_stack_kernel
    // launch is skipped; 0 = off.  Needs mixer_cluster * B <= CUs.  mixy: two (B, 160, 64) exchange buffers for the LayerNorm-ed rows
    // (alternating per layer).
    MixerParams mx;
    float* mixy[2];
    int mixer_cluster;
};

int launch_decoder_fused(const DecParams& p, hipStream_t s);
int decoder_max_coresident_blocks(int device);      // > 0, or negative if the kernel cannot be resident at all

// ---- the MLP-Mixer stack kernel (mixer_fused.hip; parameter structs above DecParams) ----
int launch_mixer_fused(const MixerParams& p, int B, hipStream_t s);

// host-side launch helpers (defined in the .hip files); all return 0 / negative
int launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s);          // gemm_f32.hip
// mid-size batches: the same big LDS-DMA tiles with K split `ksplit` ways into part[ksplit][M][N] (raw partial sums, summed in a
// fixed order by launch_splitk_resid_ln / launch_splitk_epilogue); variant as for launch_gemm (-1 = cost model over tiles * ksplit)
int launch_gemm_splitk(const GemmArgs& a, int variant, int ksplit, float* part, hipStream_t s);
// small-M path: 64x64 tiles, `ring`-deep LDS-DMA ring (4 or 8), optional split-K into part[ksplit][M][N] (epilogue then
// applied by launch_splitk_epilogue / launch_splitk_resid_ln)
int launch_gemm_ring(const GemmArgs& a, int epi, int ring, int ksplit, float* part, hipStream_t s);   // gemm_f32.hip
int launch_gemm_ring16(const GemmArgs& a, int epi, hipStream_t s);                // gemm_f32.hip: small M on 16x16x4 tiles (64 x 48)
// gemm_split.hip: fp32 -> three bf16 pieces ("split3"), and the GEMM over split3 operands on the bf16 matrix pipe (a.A / a.W = split3)
int launch_split3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int K, hipStream_t s);
int launch_gemm_split3(const GemmArgs& a, int epi, int variant, hipStream_t s);
// split-K on the split3 big tiles: ksplit copies of the tile grid in one launch, raw partial sums into part[ksplit][M][N]
int launch_gemm_split3_splitk(const GemmArgs& a, int ksplit, float* part, hipStream_t s);
// gemm_split_persist.hip: the same product on 256 persistent workgroups (a tile stream per CU, ragged last round split along K with the
// accumulators handed over through `ws`; bit-identical to launch_gemm_split3).  mode 0 = fp32 C; 1 / 2 = split3 output (a.c_split) through the
// LDS transposition / through swapped operand roles.  ws: gemm_split3_persist_ws_bytes() of device memory zeroed once, one launch at a time.
size_t gemm_split3_persist_ws_bytes();
bool gemm_split3_persist_ok(const GemmArgs& a);        // shape served by the persistent kernel (M % 128, N % 256, >= 256 tiles, no split-K)
int launch_gemm_split3_persist(const GemmArgs& a, int epi, int mode, void* ws, hipStream_t s);
// round 6: the same stream over 128 x 128 tiles with the three-stage K ring (few crops: 257 ... kPersistNarrowMaxTiles tiles of 128 x 128)
bool gemm_split3_persist_narrow_ok(const GemmArgs& a);   // N % 128, >= 256 tiles of 128 x 128 (M may be ragged), row-major A, no split-K
int launch_gemm_split3_persist_narrow(const GemmArgs& a, int epi, void* ws, hipStream_t s);
int launch_gemm_split3_splitk_stream(const GemmArgs& a, int ksplit, float* part, void* ws, hipStream_t s);      // a.ksplit > 1 units: (tile, K slice), epilogue none
int gemm_split3_persist_error(void* ws, hipStream_t s, unsigned* err_out);   // synchronises s; *err_out != 0: a hand-over spin timed out
int gemm_split3_persist_bind_host_err(void* ws, unsigned* const* host_err_slot, hipStream_t s);   // a timed-out consumer ALSO writes the host-mapped word *slot (slot: stable storage; after zeroing ws)
void* gemm_split3_persist_op_ws(hipStream_t s);        // zeroed workspace per (device, stream) for the stateless operators
// gemm_split16.hip: the split3 GEMM on v_mfma_f32_16x16x32_bf16 (what launch_gemm_split3 / _splitk / _persist run since round 4):
// one workgroup per tile (wide = 128 x 256 on 8 waves, else 128 x 128 on 4; a.ksplit copies of the grid) or 256 persistent workgroups
int launch_split16_tiles(const GemmArgs& a, int epi, int shape, hipStream_t s);      // shape 0: 128 x 256 / 8 waves, 1: 128 x 128 / 4 waves, 2: 128 x 128 / 8 waves
int launch_split16_persist(const GemmArgs& a, int epi, void* ws, bool narrow, hipStream_t s);
// the wide grid with its ragged last round as 128 x 128 half tiles; 0 launched, 1 = does not apply to this shape (nothing launched), < 0 error
int launch_split16_tiles_tail(const GemmArgs& a, int epi, int cus, bool tail8, hipStream_t s);
// small-M split3 GEMM (64x64 tiles, LDS-DMA ring, optional split-K into part[ksplit][M][N] without epilogue)
int launch_gemm_split3_ring(const GemmArgs& a, int epi, int ksplit, float* part, hipStream_t s);
// LayerNorm (D = 1280) whose result is written as a split3 operand [rows][D/8][3][8] instead of fp32 (same arithmetic as launch_layernorm)
int launch_layernorm_split3(const float* x, const float* g, const float* b, void* y_split, int rows, int D, float eps, hipStream_t s);
int launch_gemm_skinny(const GemmArgs& a, int epi, hipStream_t s);                // gemm_skinny.hip
int launch_vit_attention(const float* qkv, float* out, int B, hipStream_t s);     // attention.hip
// the same with the output written as a split3 operand [B*192][1280/8][3][8] (the proj GEMM's A in the split3 mode) instead of fp32
int launch_vit_attention_split3(const float* qkv, void* out_split, int B, hipStream_t s);
int launch_vit_attention_keysplit(const float* qkv, float* out, int B, hipStream_t s);   // few crops: keys split over the 4 waves
int launch_vit_attention_variant(const float* qkv, float* out, int B, int variant, hipStream_t s);   // 0 = rule; 1 / 3 / 5 / 12 / 6
// attention_b16.hip: the same attention with every product as 3 x 3 bf16 pieces on v_mfma_f32_16x16x32_bf16 (six products, fp32 accumulate);
// out = fp32 (B,192,1280) or the split3 operand; qt = 0 (batch-size rule), 1 (64-query workgroups), 3 (one workgroup per (crop, head))
int launch_vit_attention_b16(const float* qkv, void* out, int B, bool out_split, int qt, hipStream_t s);
// rowops.hip
int launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, int relu,
                     hipStream_t s);
int launch_splitk_epilogue(const GemmArgs& a, int epi, const float* part, int S, hipStream_t s);
int launch_splitk_resid_ln(const float* part, int S, int rows, int D, const float* bias, const float* resid, float* xout,
                           const float* gamma, const float* beta, float* y, float eps, hipStream_t s, bool y_is_split3 = false);
int launch_add_ln64(const float* x, const float* y, const float* g, const float* b, float* s_out, float* z_out, int rows,
                    float eps, hipStream_t s);
int launch_im2col_patch(const float* img, float* A, int B, hipStream_t s);
int launch_im2col_patch_split3(const float* img, void* A_split, int B, hipStream_t s);      // the same operand as [B*192][768/8][3][8] bf16 pieces
int launch_transpose(const float* in, float* out, int Bn, int R, int C, hipStream_t s);
int launch_softmax_argmax2048(const float* logits, float* probs, int32_t* idx, int rows, hipStream_t s);
int launch_conv3_gather(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tout, int C, int dil,
                        int prerelu, hipStream_t s);
int launch_conv_repack(const float* w, float* wp, int co, int ci, int kk, hipStream_t s);
int launch_conv_gather_general(const float* in, float* out, const int32_t* src, int Bn, int Tin, int Tsrc, int Tout, int C,
                               int Cp, int ks, int stride, int pad, hipStream_t s);
int launch_conv_repack_pad(const float* w, float* wp, int co, int ci, int cp, int kk, hipStream_t s);
// head.hip
int launch_decoder_init(const float* bias, const float* pos, float* x, int B, int E, hipStream_t s);
int launch_cross_attn(const float* q, const float* kv, int64_t ldkv, int koff, float* out, int B, hipStream_t s);
int launch_assemble(const float* ro, int ldro, const float* bpose, const float* init_pose, const float* init_betas,
                    const float* init_cam, float* pose6d, float* rotmat, float* betas, float* cam, float* cam_t,
                    float* focal, float focal_length, float img_size, int B, hipStream_t s);
int launch_rot6d(const float* x, float* R, int n, hipStream_t s);
int launch_aa_to_rotmat(const float* aa, float* R, int n, hipStream_t s);
int launch_cam_t(const float* cam, float* cam_t, float focal_length, float img_size, int B, hipStream_t s);
int launch_vq_argmin_rows(const float* x, const float* dot, const float* cnorm, int32_t* idx, float* dist, int rows,
                          hipStream_t s);
int launch_code_norm(const float* cb, float* cn, int ncode, hipStream_t s);
// lbs.hip
int launch_lbs_jreg(const float* Jreg, const float* vt, const float* sd, float* Jt, float* Jsd, hipStream_t s);
// dirsT (20670 x 224): [shapedirs | posedirs | 0]^T built once by launch_lbs_build_dirs; scratch A (B,24,12),
// xf (B, THMR_LBS_XF) [224 operand floats per crop, then 27*57 regressor partial sums per crop], Jtr (B,24,3), vposed (B,20670)
constexpr int THMR_LBS_KX = 224, THMR_LBS_XF = 224 + 27 * 57;
int launch_lbs_build_dirs(const float* sd, const float* pd, float* dirsT, hipStream_t s);
int launch_lbs(const float* rotmat, const float* betas, const float* cam_t, const float* Jt, const float* Jsd,
               const int32_t* parents, const float* vt, const float* dirsT, const float* W, const float* J19,
               const int32_t* extra, const int32_t* jmap, const int32_t* update_hips, float* A, float* xf, float* Jtr,
               float* vposed, float* verts, float* joints, float* kp2d, float focal_over_size, int B, float* xv,
               unsigned* cnt, hipStream_t s);
// extra scratch of the fused skin + joints kernel: xv (B, 21, 3) picked extra-joint vertices, cnt (B) arrival counters (zeroed ONCE)
int launch_rodrigues(const float* aa, float* R, int n, hipStream_t s);
// eval.hip
int launch_eval_pose(const float* pred, const float* gt, int nj, int gt_stride, const int32_t* kp, int nkp, int pelvis_ind,
                     int pelvis_mode, float* mpjpe, float* re, float* pelv, int B, hipStream_t s);
int launch_eval_pve(const float* pv, const float* gv, const float* pelv, int nv, float* pve, int B, hipStream_t s);
int launch_regress_joints(const float* J, const float* verts, int nj, int nv, float* out, int B, hipStream_t s);
