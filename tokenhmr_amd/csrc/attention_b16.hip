// ViT global self-attention on the bf16 matrix pipe: softmax(q k^T) v for one (crop, head) with every fp32 operand multiplied as three
// bf16 pieces, six products per pair, fp32 accumulate ("split3"; HISTORY.md 10.6) — what the ViT GEMMs of the split3 mode do, applied to
// vit.py:113-122 (Attention.forward between the qkv and proj Linears).  Input / output as attention.hip: qkv (B,192,3840) fp32 with q
// pre-scaled, out (B,192,1280) fp32 or the split3 operand of the proj GEMM.
//
// Why: in the split3 mode the fp32-MFMA attention kernel runs behind the GEMMs at the clock they leave (1.4-1.6 GHz): 129 us per launch
// at 64 crops, 4.2 of a 73.5 ms step (profiles/r4n_kernel_stats_split3.csv), its matrix work alone 12.1 GFLOP / 157 TFLOP/s = 77 us at
// full clock.  The same products as 6 x bf16 MFMAs are 72.5 GFLOP per launch on a pipe that sustains 1.4-1.8 PFLOP/s here: 40-50 us.
//
// Shape of the work.  v_mfma_f32_16x16x32_bf16: lane (l15 = l & 15, g = l >> 4) of an operand holds the 8 bf16 of row / column l15,
// k = 8 g ... 8 g + 7 (one 16-byte register quad); the result D[i = 4 g + r][j = l15] in register r.
//   S^T = K Q^T : A = K fragment (i = key, k = d), B = Q fragment (k = d, j = query).  d = 80 is three k steps of 32, the last one
//                 half empty (zero columns in the K image, zero registers in Q).  As in attention.hip the transposed product leaves a
//                 query's scores in 4 lanes x registers: s[qt][kt][r] = S[query 16 qt + l15][key 16 kt + 4 g + r].
//   O^T = V^T P^T : A = V^T fragment (i = d, k = key), B = P (k = key, j = query).  A k step is 32 keys = two score tiles: the lane's 8
//                 k slots are {tile 2 st: 4 g + r} then {tile 2 st + 1: 4 g + r} — its OWN score registers, split into pieces in place; the
//                 V^T image stores a key at the slot that order implies.  Each lane ends with 4 consecutive d of one query (16-byte stores).
// A workgroup = 4 waves x QT tiles of 16 queries; K and V are staged per BLOCK of 64 keys as bf16 pieces in ONE 72 KB LDS image — two
// workgroups per CU, one's load / split / softmax phases under the other's MFMAs.  Layout of the two images (round 5; the lane-group model
// of tests/test_host_logic.py checks every access below, and a PMC pass counts what is left):
//   K block  [piece][key][96 d]: rows of 224 bytes (192 + 32 pad = 14 bank slots of 16 bytes, 2 x odd): gfx950 serves a ds_read_b128 in groups
//            of {8 rows of one g, the other 8 rows of g + 1} (MI355X_MICROARCH.md, LDS) and a ds_write_b64 in groups of 16 consecutive lanes
//            (here 4 keys x 4 eight-byte parts); both are conflict-free at this stride.  (Round 4's 208 bytes suited 16 CONSECUTIVE lanes per
//            read group: every fragment read was a 2-way conflict, 5.8 M of 12.0 M LDS cycles per launch, profiles/r4ah_pmc_lds.json.)
//   V^T block [piece][d][64 key slots]: rows of 160 bytes (128 + 32 pad = 10 slots, 2 x odd), the 16-byte chunk c of row d stored at chunk
//            c ^ ((d >> 2) & 1).  Reads: lane (l15, g) takes chunk (4 st + g) ^ ((l15 >> 2) & 1) of row 16 dt + l15 — conflict-free under the
//            same groups, and st / dt stay immediate offsets of ONE per-lane base (a 3-bit XOR on unpadded 128-byte rows is conflict-free
//            too but needs a base per k step: the 192-query instantiation has no register to spare).  Writes: the staging threads are laid
//            out so that 16 consecutive lanes hold 8 consecutive d x both 8-byte halves of one chunk; with the swizzle those are 32 distinct
//            banks (round 4: 16 rows x one half at a 144-byte stride = 2-way; the 160-byte stride alone would make it 4-way).  The two key blocks are combined the flash-attention way (running row maximum, accumulators rescaled once):
// the scores of a block live in 24 QT registers instead of 48 QT for the whole row, which is what lets Q pieces, P pieces, the output
// accumulators and a block of loads in flight fit 256 registers.  global -> registers -> split3_pair (v_cvt_pk_bf16_f32) -> ds_write:
// LDS-DMA cannot convert, and a split3 q / k / v from the qkv GEMM would cost its epilogue +50 % stores for operands read once.
//
// Per query the instruction sequence does not depend on QT, the grid or the batch size: QT = 3 (one workgroup per (crop, head)) and
// QT = 1 (three workgroups of 64 queries: few crops) are bit-identical.  NOT bit-identical to attention.hip (fp32 products there,
// 2^-24-truncated six-product sums here; another order of the key sum): the split3 mode's own attention, same error class as its GEMMs.
#include "common.h"
#include "attention_device.h"
#include "gemm_split_device.h"

namespace {

constexpr int KB = 64;                   // keys per block (3 blocks)
constexpr int NBLK = NTOK / KB;
constexpr int KRS = 224;                 // K image row stride, bytes: 96 d x 2 (80 + 16 zero columns) + 32 pad
constexpr int VRS = 160;                 // V^T image row stride, bytes: 64 key slots x 2 + 32 pad; chunk ^= (row >> 2) & 1
constexpr int KPL = KB * KRS;            // one piece plane of the K image: [64 keys][96 d]
constexpr int VPL = HD * VRS;            // one piece plane of the V^T image: [80 d][64 key slots]
constexpr int K_IMG = 3 * KPL, V_IMG = 3 * VPL;          // 43,008 + 38,400 bytes: two workgroups per CU
static_assert(2 * (K_IMG + V_IMG) <= 160 * 1024, "two workgroups per CU");
static_assert(V_IMG >= 4 * 16 * 480, "the split3 epilogue assembles 4 waves x 16 rows x 480 bytes in the dead V^T image");
constexpr float LOG2E_F = 1.44269504088896340736f;

typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 pieces8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d});
}

// QT = 16-query tiles per wave: 3 = one workgroup per (crop, head), 1 = three workgroups of 64 queries.  `nitems` = B * 16 * (3 / QT) work
// items; at most 512 workgroups (two per CU) walk them (workgroup (xcd, i) takes items i, i + grid / 8, ... of its XCD's contiguous share),
// and the block pipeline does not drain between items: the next item's first K block and Q step are requested under the current item's
// last P.V, exactly as block n + 1's are under block n's.
// ABL (experiments build only, timing-only ablations with garbage results — DESIGN.md 3.2's diagnosis, round 6): bit 0 = q is loaded for an
// item's FIRST key block only (blocks 1, 2 re-split stale registers: the q re-reads' memory traffic gone, the vector work kept); bit 1 = K / V
// are loaded for the first block of a workgroup's first item only (later blocks stage score registers into the images: no K / V traffic at all,
// the staging's vector + LDS work kept); bit 2 = q is re-SPLIT for the first block only (the conversion's vector work gone, its loads kept).
// EARLY (experiments build, round 6): the block's V and the next block's K are requested at the START of the S^T phase's k step EARLY - 1
// (1 = before step 0, 2 = before step 1, 3 = before step 2) instead of after it — a whole MFMA phase for the loads to land instead of the
// softmax's ~270 vector instructions; 40 more registers live through the phase.
// OCC (round 6 experiment): workgroups per CU the kernel is COMPILED for.  2 = 256 registers per wave (what ships); 1 = one workgroup per CU
// and 512 registers (256 + 256 accumulation) per wave: the grid experiment showed the second co-resident workgroup adds only 9 %, so the
// registers it costs may be worth more than it — no spills, and room for the earlier loads that spill at OCC 2.
template <int QT, bool SPLIT, int ABL = 0, int EARLY = 0, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void vit_attention_b16_kernel(const float* __restrict__ qkv, float* __restrict__ out, int nitems) {
    constexpr int QB = 3 / QT;
    static_assert(QT == 1 || QT == 3, "192 queries = QB workgroups x 4 waves x QT tiles of 16");
    __shared__ __attribute__((aligned(16))) char smem[K_IMG + V_IMG];
    char* const kimg = smem;
    char* const vimg = smem + K_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware order (attention.hip): workgroup b runs on XCD b % 8; an XCD's share of the items is contiguous, so the 16 heads of a crop
    // stream its 15 KB token rows through ONE L2.  nitems and the grid are multiples of 8 (launcher).
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, wpx = gridDim.x >> 3, per_xcd = nitems >> 3;
    int it = within;
    if (it >= per_xcd) return;
    // an item: crop (its rows are the buffer resource), head column, first query of this wave
    struct Item { int crop, hcol, q0; };
    auto item_of = [&](int i) {
        const int lg = xcd * per_xcd + i, bh = lg / QB, qb = lg - bh * QB;
        return Item{bh / NH, (bh % NH) * HD, (qb * 4 + wave) * 16 * QT};
    };
    // ---- global -> registers: buffer loads = {a crop's rows as the resource} + {wave-uniform SGPR offset} + {ONE 32-bit per-lane offset}.
    //      (Plain pointers: hipcc folds uniform pointer + lane offset + constant into a 64-bit VGPR address per load: 16 registers a batch.) ----
    auto rsrc_of = [&](int crop) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + (int64_t)crop * NTOK * QKV_LD), 0, NTOK * QKV_LD * 4, 0x00020000); };
    auto ld4 = [&](const Item& im, int uoff_floats, uint32_t off) {        // qkv[crop][uoff_floats + off / 4 ... + 3], uoff wave-uniform
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(im.crop), off, uoff_floats * 4, 0);
        return f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
    };
    auto ld1 = [&](const Item& im, int uoff_floats, uint32_t off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_of(im.crop), off, uoff_floats * 4, 0)); };

    // ---- staging roles (all 256 threads, 20 registers a block).
    //      K block: thread (key tid >> 2, part tid & 3) takes the 4-d groups part + 4 m, m = 0..4, of its key.
    //      V block: thread (key quad kq, dg) takes d = dg + 16 m, m = 0..4, of keys 4 kq ... 4 kq + 3.  A k step of P.V is
    //      32 keys: slot 8 gk + 4 a + r of the step <-> key 16 a + 4 gk + r (the lane's own score registers of two 16-key tiles): key quad
    //      kq = 8 st + 4 a + gk lands in chunk 4 st + gk, 8-byte half a.  tid -> (kq, dg) puts 8 consecutive d x the two halves a of one
    //      chunk into 16 consecutive lanes (the conflict-free ds_write_b64 group, see the layout note above); per load instruction a wave
    //      still reads 64 contiguous bytes of 4 key rows. ----
    const int kkey = tid >> 2, kpart = tid & 3;
    const int vdg = (tid & 7) + 8 * ((tid >> 4) & 1), vkq = ((tid >> 5) & 3) | (((tid >> 3) & 1) << 2) | ((tid >> 7) << 3);
    const uint32_t koff = (uint32_t)(kkey * QKV_LD + kpart * 4) * 4u;
    const uint32_t voff = (uint32_t)(vkq * 4 * QKV_LD + vdg) * 4u;
    const uint32_t qoff = (uint32_t)(l15 * QKV_LD + g * 8) * 4u;
    char* const kdst = kimg + kkey * KRS + kpart * 8;
    char* const vdst = vimg + vdg * VRS + (((4 * (vkq >> 3) + (vkq & 3)) ^ ((vdg >> 2) & 1)) * 16) + 8 * ((vkq >> 2) & 1);      // rows dg + 16 m: the same swizzle
    f32x4 kr[5];
    float vr[5][4];
    auto load_k = [&](const Item& im, int blk) {
        const int ub = im.hcol + DIM + blk * KB * QKV_LD;
#pragma unroll
        for (int m = 0; m < 5; ++m) kr[m] = ld4(im, ub + m * 16, koff);
    };
    auto write_k = [&]() {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            uint32_t H[2], M[2], L[2];
            split3_pair(kr[m][0], kr[m][1], H[0], M[0], L[0]);
            split3_pair(kr[m][2], kr[m][3], H[1], M[1], L[1]);
            char* o = kdst + m * 32;
            *reinterpret_cast<u32x2v*>(o) = u32x2v{H[0], H[1]};
            *reinterpret_cast<u32x2v*>(o + KPL) = u32x2v{M[0], M[1]};
            *reinterpret_cast<u32x2v*>(o + 2 * KPL) = u32x2v{L[0], L[1]};
        }
    };
    auto load_v = [&](const Item& im, int blk) {
        const int ub = im.hcol + 2 * DIM + blk * KB * QKV_LD;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) vr[m][r] = ld1(im, ub + r * QKV_LD + m * 16, voff);
    };
    auto write_v = [&]() {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            uint32_t H[2], M[2], L[2];
            split3_pair(vr[m][0], vr[m][1], H[0], M[0], L[0]);
            split3_pair(vr[m][2], vr[m][3], H[1], M[1], L[1]);
            char* o = vdst + m * 16 * VRS;
            *reinterpret_cast<u32x2v*>(o) = u32x2v{H[0], H[1]};
            *reinterpret_cast<u32x2v*>(o + VPL) = u32x2v{M[0], M[1]};
            *reinterpret_cast<u32x2v*>(o + 2 * VPL) = u32x2v{L[0], L[1]};
        }
    };

    // ---- Q pieces of k step s: B operand of S^T = K Q^T, lane (query l15, g) holds d = 32 s + 8 g ... + 7 (nothing past d = 79) ----
    f32x4 qraw[QT][2];
    auto load_q = [&](const Item& im, int s) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int ub = im.hcol + (im.q0 + qt * 16) * QKV_LD + s * 32;
            if (s < 2 || g < 2) {
                qraw[qt][0] = ld4(im, ub, qoff);
                qraw[qt][1] = ld4(im, ub + 4, qoff);
            } else {
                qraw[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                qraw[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    bf16x8 qf[QT][3];
    auto split_q = [&]() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            uint32_t H[4], M[4], L[4];
            split3_pair(qraw[qt][0][0], qraw[qt][0][1], H[0], M[0], L[0]);
            split3_pair(qraw[qt][0][2], qraw[qt][0][3], H[1], M[1], L[1]);
            split3_pair(qraw[qt][1][0], qraw[qt][1][1], H[2], M[2], L[2]);
            split3_pair(qraw[qt][1][2], qraw[qt][1][3], H[3], M[3], L[3]);
            qf[qt][0] = pieces8(H[0], H[1], H[2], H[3]);
            qf[qt][1] = pieces8(M[0], M[1], M[2], M[3]);
            qf[qt][2] = pieces8(L[0], L[1], L[2], L[3]);
        }
    };

    const char* const kfr = kimg + l15 * KRS + g * 16;   // K fragment of (tile kt, step s): + kt * 16 * KRS + s * 64 (+ plane)
    // V^T fragment of (tile dt, step st): chunk (4 st + g) ^ ((l15 >> 2) & 1) of row 16 dt + l15: + dt * 16 * VRS + st * 64 (+ plane)
    const char* const vfr = vimg + l15 * VRS + ((g ^ ((l15 >> 2) & 1)) * 16);

    Item cur = item_of(it);
    load_k(cur, 0);
    load_q(cur, 0);
    // d = 80 ... 95 of every row of the K image are zero, once: 3 x 64 rows x 2 chunks of 16 bytes
    for (int idx = tid; idx < 3 * KB * 2; idx += 256) {
        const int pl = idx / (2 * KB), rem = idx - pl * (2 * KB);
        *reinterpret_cast<u32x4*>(kimg + pl * KPL + (rem >> 1) * KRS + 160 + (rem & 1) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    write_k();
    __syncthreads();

#pragma unroll 1
    for (;;) {
        const bool has_next = it + wpx < per_xcd;            // wave-uniform
        const Item nxt = has_next ? item_of(it + wpx) : cur;
        f32x4 o[QT][5];
        float lsum[QT], cneg[QT];            // per-lane partial row sums; -(running row maximum) log2 e
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            lsum[qt] = 0.f;
            cneg[qt] = 3.0e38f;              // "no maximum yet": the first block's rescale factor is 2^(-huge) = 0 on sums and outputs that are 0
        }
#pragma unroll 1
        for (int blk = 0; blk < NBLK; ++blk) {
            const bool more = blk + 1 < NBLK;                // this item has another block; else the next item's first block follows (if any)
            // ---- S^T of the block: 3 k steps x 4 key tiles x (6 products x QT) MFMAs; the fragment of the next (step, tile) is read before this one's MFMAs ----
            f32x4 s[QT][4];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 kc[3], kn[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) kc[pc] = *reinterpret_cast<const bf16x8*>(kfr + pc * KPL);
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                if constexpr (EARLY > 0 && (ABL & 2) == 0) {
                    if (st == EARLY - 1) {
                        load_v(cur, blk);
                        if (more) load_k(cur, blk + 1);
                        else if (has_next) load_k(nxt, 0);
                    }
                }
                if ((ABL & 4) == 0 || blk == 0) split_q();
                if (st + 1 < 3 && ((ABL & 1) == 0 || blk == 0)) load_q(cur, st + 1);          // in flight under this step's MFMAs
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int nx = st * 4 + kt + 1;
                    if (nx < 12) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            kn[pc] = *reinterpret_cast<const bf16x8*>(kfr + pc * KPL + (nx % 4) * 16 * KRS + (nx / 4) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < NPROD; ++p)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[piece_w(p)], qf[qt][piece_a(p)], s[qt][kt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) kc[pc] = kn[pc];
                }
            }
            // ---- this block's V and the next block's K (this item's, or the next item's first): in flight under the softmax ----
            if constexpr ((ABL & 2) == 0) {
                if constexpr (EARLY == 0) {
                    load_v(cur, blk);
                    if (more) load_k(cur, blk + 1);
                    else if (has_next) load_k(nxt, 0);
                }
            } else {                                           // ablation: the staging registers filled from live score registers instead of memory
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    kr[m] = s[0][m & 3];
                    asm volatile("" : "+v"(kr[m]));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vr[m][r] = s[0][(m + r) & 3][r];
                        asm volatile("" : "+v"(vr[m][r]));
                    }
                }
            }
            __syncthreads();                                   // every wave is done with the K image (and with the V^T image: its P.V came first)
            // ---- running softmax: m = max(m, block max); e = 2^(s log2 e - m log2 e); earlier sums and outputs scaled by 2^((m_old - m) log2 e) ----
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float m = s[qt][0][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][kt][r]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const float c = fminf(-(m * LOG2E_F), cneg[qt]);                   // -(maximum so far) log2 e
                const float alpha = __builtin_amdgcn_exp2f(c - cneg[qt]);          // 1 exactly when the maximum did not move
                cneg[qt] = c;
                lsum[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) o[qt][dt] = o[qt][dt] * alpha;
                const f32x2 c2 = splat2(c), l2 = splat2(LOG2E_F);
                f32x2 sum2 = splat2(0.f);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x2 t = __builtin_elementwise_fma(f32x2{s[qt][kt][2 * hh], s[qt][kt][2 * hh + 1]}, l2, c2);
                        const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                        s[qt][kt][2 * hh] = e.x;
                        s[qt][kt][2 * hh + 1] = e.y;
                        sum2 += e;
                    }
                lsum[qt] += sum2.x + sum2.y;
            }
            // ---- V block -> its image (transposed: rows = d, columns = key slots), next K block -> its image ----
            write_v();
            if (more || has_next) write_k();
            __syncthreads();
            if constexpr ((ABL & 1) == 0) {
                if (more) load_q(cur, 0);                      // the next block's first Q step, in flight under P.V
                else if (has_next) load_q(nxt, 0);
            } else if (!more && has_next) load_q(nxt, 0);
            // ---- O^T += V^T P^T: 2 k steps of 32 keys x 5 d tiles x (6 products x QT) MFMAs ----
            bf16x8 vc[3], vn[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) vc[pc] = *reinterpret_cast<const bf16x8*>(vfr + pc * VPL);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf[QT][3];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    uint32_t H[4], M[4], L[4];
                    split3_pair(s[qt][2 * st][0], s[qt][2 * st][1], H[0], M[0], L[0]);
                    split3_pair(s[qt][2 * st][2], s[qt][2 * st][3], H[1], M[1], L[1]);
                    split3_pair(s[qt][2 * st + 1][0], s[qt][2 * st + 1][1], H[2], M[2], L[2]);
                    split3_pair(s[qt][2 * st + 1][2], s[qt][2 * st + 1][3], H[3], M[3], L[3]);
                    pf[qt][0] = pieces8(H[0], H[1], H[2], H[3]);
                    pf[qt][1] = pieces8(M[0], M[1], M[2], M[3]);
                    pf[qt][2] = pieces8(L[0], L[1], L[2], L[3]);
                }
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) {
                    const int nx = st * 5 + dt + 1;
                    if (nx < 10) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            vn[pc] = *reinterpret_cast<const bf16x8*>(vfr + pc * VPL + (nx % 5) * 16 * VRS + (nx / 5) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < NPROD; ++p)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vc[piece_w(p)], pf[qt][piece_a(p)], o[qt][dt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) vc[pc] = vn[pc];
                }
            }
        }

        // ---- normalise + store: lane (l15, g) holds d = 16 dt + 4 g ... + 3 of query 16 qt + l15 ----
        float inv[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float l = lsum[qt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            inv[qt] = 1.0f / l;
        }
        if constexpr (SPLIT) {
            // A head's slice of a split3 row is 480 contiguous bytes (k-groups 10 h ... 10 h + 9 x 3 pieces x 16 bytes), but the MFMA leaves a
            // lane with ONE row and 8 of its columns: written from the registers, a wave's store instruction touches 16 rows x two lines for
            // 16 bytes each, three times over (the pieces H, M, L of a k-group are 16 bytes apart).  The V^T image is dead between the
            // item's last P.V and the next block's write_v (one barrier away on both sides), so each wave assembles its 16-row tile there —
            // rows of exactly 480 bytes, lane (l15, g) writes its chunks — and streams it out as 7.5 instructions of 64 x 16 CONSECUTIVE
            // bytes: chunk c = 64 i + lane of the tile is LDS byte 16 c and row c / 30 of the operand.  Measured (profiles/r4y_ vs r4r_):
            // 64-query workgroups 106.5 vs 111.0 us per launch at 64 crops, 27.1 vs 28.3 at 16; 192-query workgroups 98.9 vs 99.1 — the
            // 16 us that split3 output costs over fp32 output there (82.9 us) are not the store pattern.
            __syncthreads();                                   // every wave is done with the V^T image
            char* const stg = vimg + wave * (16 * 480);
            char* const orow = reinterpret_cast<char*>(out) + ((int64_t)cur.crop * NTOK + cur.q0) * (DIM * 6) + cur.hcol * 6;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                char* const srow = stg + l15 * 480;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {               // tiles (dt, dt + 1) = (0, 1), (2, 3): after the swap an even-g lane holds 8 columns of the first, an odd-g lane of the second
                    f32x4 x = o[qt][2 * pr] * inv[qt], y = o[qt][2 * pr + 1] * inv[qt];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                        const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[j]), __float_as_uint(y[j]), false, false);
                        x[j] = __uint_as_float(sw.x);
                        y[j] = __uint_as_float(sw.y);
                    }
                    uint32_t H[4], M[4], L[4];
                    split3_pair(x[0], x[1], H[0], M[0], L[0]);
                    split3_pair(x[2], x[3], H[1], M[1], L[1]);
                    split3_pair(y[0], y[1], H[2], M[2], L[2]);
                    split3_pair(y[2], y[3], H[3], M[3], L[3]);
                    char* c = srow + ((2 * pr + (g & 1)) * 2 + (g >> 1)) * 48;      // k-group (16 dt + 8 (g >> 1)) / 8 of the slice, dt = 2 pr + (g & 1)
                    *reinterpret_cast<u32x4*>(c) = u32x4{H[0], H[1], H[2], H[3]};
                    *reinterpret_cast<u32x4*>(c + 16) = u32x4{M[0], M[1], M[2], M[3]};
                    *reinterpret_cast<u32x4*>(c + 32) = u32x4{L[0], L[1], L[2], L[3]};
                }
                {                                              // tile dt = 4: columns 64 + 4 g ... + 3 = the 8-byte half (g & 1) of k-group 8 + (g >> 1)
                    const f32x4 v = o[qt][4] * inv[qt];
                    uint32_t H[2], M[2], L[2];
                    split3_pair(v[0], v[1], H[0], M[0], L[0]);
                    split3_pair(v[2], v[3], H[1], M[1], L[1]);
                    char* c = srow + (8 + (g >> 1)) * 48 + (g & 1) * 8;
                    *reinterpret_cast<u32x2v*>(c) = u32x2v{H[0], H[1]};
                    *reinterpret_cast<u32x2v*>(c + 16) = u32x2v{M[0], M[1]};
                    *reinterpret_cast<u32x2v*>(c + 32) = u32x2v{L[0], L[1]};
                }
                // wave-private: the wave's own LDS writes are complete before its reads are issued (same queue, in order)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = i * 64 + lane;               // chunk of the 16 x 30 tile
                    if (i < 7 || lane < 32) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + c * 16);
                        const int r = c / 30;
                        *reinterpret_cast<u32x4*>(orow + (int64_t)(qt * 16 + r) * (DIM * 6) + (c - r * 30) * 16) = v;
                    }
                }
            }
        } else {
            float* obase = out + (int64_t)cur.crop * NTOK * DIM + cur.hcol;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int dt = 0; dt < 5; ++dt)
                    *reinterpret_cast<f32x4*>(obase + (int64_t)(cur.q0 + qt * 16 + l15) * DIM + dt * 16 + g * 4) = o[qt][dt] * inv[qt];
        }
        if (!has_next) break;
        cur = nxt;
        it += wpx;
    }
}

}  // namespace

// out_split == true: `out` is the split3 operand [B*192][1280/8][3][8] bf16.  qt = 0: the batch-size rule (64-query workgroups up to 20 crops);
// 1 / 3 force a shape (bit-identical).
int launch_vit_attention_b16(const float* qkv, void* out, int B, bool out_split, int qt, hipStream_t s) {
    if (B <= 0 || (qt != 0 && qt != 1 && qt != 3)) return -1;
    if (qt == 0) qt = B <= 20 ? 1 : 3;        // stand-alone, us per launch qt = 1 / 3: 20.2 / 25.7 at 8 crops, 28.3 / 31.6 at 16, 53.1 / 50.0 at 32 (profiles/r4q_, r4r_attention_b16_*.jsonl)
    float* o = reinterpret_cast<float*>(out);
    const int nitems = B * NH * (3 / qt);                        // a multiple of 16: the kernel splits items and grid by the 8 XCDs
    static_assert(NH % 8 == 0, "items per crop must divide by the XCD count");
    dim3 grid(nitems < 512 ? nitems : 512);
#ifdef THMR_EXPERIMENTS
    {
        // THMR_ATTN_GRID=<n>: cap the grid (256 = ONE workgroup per CU: how much do the two co-resident workgroups overlap each other?)
        const char* gk = thmr_knob("THMR_ATTN_GRID");
        const int gcap = gk ? atoi(gk) : 0;
        if (gcap >= 8 && (gcap % 8) == 0 && (unsigned)gcap < grid.x) grid = dim3(gcap);
        const char* ek = thmr_knob("THMR_ATTN_EARLY");
        const int early = ek ? atoi(ek) : 0;
        const char* ok = thmr_knob("THMR_ATTN_OCC");
        if (ok && ok[0] == '1' && qt == 3 && out_split) {      // compiled for one workgroup per CU (512 registers per wave), at most 256 workgroups
            if (grid.x > 256) grid = dim3(256);
            switch (early) {
                case 0: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 0, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 1: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 1, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 2: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 2, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 3: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 3, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                default: return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        if (early > 0 && qt == 3 && out_split) {
            switch (early) {
                case 1: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 2: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 2>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 3: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 0, 3>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                default: return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        const char* k = thmr_knob("THMR_ATTN_ABL");
        const int abl = k ? atoi(k) : 0;
        if (abl > 0 && qt == 3 && out_split) {
            switch (abl) {
                case 1: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 1>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 2: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 2>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 3: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 3>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 4: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 4>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 5: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 5>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                case 7: hipLaunchKernelGGL((vit_attention_b16_kernel<3, true, 7>), grid, dim3(256), 0, s, qkv, o, nitems); break;
                default: return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
    }
#endif
    if (qt == 1) {
        if (out_split) hipLaunchKernelGGL((vit_attention_b16_kernel<1, true>), grid, dim3(256), 0, s, qkv, o, nitems);
        else hipLaunchKernelGGL((vit_attention_b16_kernel<1, false>), grid, dim3(256), 0, s, qkv, o, nitems);
    } else {
        if (out_split) hipLaunchKernelGGL((vit_attention_b16_kernel<3, true>), grid, dim3(256), 0, s, qkv, o, nitems);
        else hipLaunchKernelGGL((vit_attention_b16_kernel<3, false>), grid, dim3(256), 0, s, qkv, o, nitems);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
