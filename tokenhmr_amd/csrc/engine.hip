// TokenHMR inference engine + C ABI (include/tokenhmr_hip.h).
//
// One engine per GPU: owns a packed weight arena (reference checkpoint tensors at fixed offsets, so a
// single RCCL broadcast of the arena replicates the model) and a static activation arena sized for
// max_batch crops; thmr_forward launches the whole path on the caller's stream with no allocation
// and no host synchronisation.
//
// Hot path orchestrated here (reference: tokenhmr/lib/models/tokenhmr.py:135-188 forward_step):
//   ViT-H            vit.py:320-343          -> vit_forward()
//   decoder + head   token_head.py:65-128    -> head_forward()
//   SMPL + camera    smpl_wrapper.py:27-41, geometry.py:86-124, tokenhmr.py:165-187 -> lbs()
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tokenhmr_hip.h"
#include "common.h"

namespace {

constexpr int TOK = 192, DIM = 1280, HEADS = 16, MLP = 5120;
constexpr int E = 1024, INNER = 512, DEC_MLP = 1024;
constexpr int TN = 160, NCLS = 2048, HID = 64, HID_INTER = 256, TOK_INTER = 64, MIX = 4;
constexpr int CODE = 256, VQW = 512, VQJ = 21;
constexpr int NV = 6890, NJ = 24, NB = 10, NP = 207;
constexpr float VIT_EPS = 1e-6f, LN_EPS = 1e-5f;
constexpr float FOCAL = 5000.0f, IMG = 256.0f;
// small-batch ViT path (gemm_ring_kernel): used while M = 192*B <= kSmallM; crossovers measured in profiles/r1_small_gemm_variants.log
constexpr int kSmallM = 1152, kSplitKMax = 4;     // B <= 6
constexpr int kKeysplitMaxB = 2;                  // key-split attention kernel: one and two crops
constexpr bool kAttnB16 = true;                   // split3 mode's attention: true = attention_b16.hip (the mode's arithmetic: three bf16 pieces per operand on the bf16
                                                  // matrix pipe; 4.03 -> 3.48 ms per 64-crop step, profiles/r4r_engine_b64_attention_b16_ab.log), false = the fp32-MFMA kernels with split3 output
// mid-size batches: from 7 to 16 crops proj / fc2 (N = 1280: 110-240 output tiles of 128x128 on 512 resident slots) run split-K 2
// on the big LDS-DMA tiles (measured per batch size and per GEMM, profiles/r3d_mid_batch_splitk_sweep.log, r3f_mid_batch_forced_tile.log:
// -10 % per call at 9 and 10 crops, -2.5 ... -3.7 % at 11 ... 16, -2 % at 7 and 8 with the 64x128 tile; from 17 on the unsplit launch
// is the faster one, and 4 ways never beats 2).  That range is its own regime of the K sum.
constexpr int kMidLoM = 7 * 192, kMidHiM = 16 * 192, kMidSplit = 2;
// thmr_set_vit_gemm(1): the split3 mode serves calls of kSplit3LowMinB (3) crops and more, in ranges.  From THIS many crops on the
// N = 1280 GEMM proj runs with its K sum unsplit (128 x 256 tiles, one workgroup per CU); below it proj / fc2 split K (next constants)
constexpr int kSplit3MinB = 16;
// ... and 5 ... 15 crops run them with proj / fc2 split K two ways (60-120 tiles of 128 x 256 otherwise): the mode's own mid regime
constexpr int kSplit3MidMinB = 5, kSplit3MidSplit = 2;
// ... and 3 and 4 crops four ways (100-120 workgroups of 128 x 128 otherwise: fc2 3.3 vs 2.0 ms per call for the exact-fp32 ring kernel);
// with them 5 / 6 crops run at 447 / 478 crops/s against 413 / 423 (profiles/r3am_split3_mid_regime_3_to_6_crops.log).  One and two
// crops (2-3 row tiles of 128) stay with the exact-fp32 kernels.
constexpr int kSplit3LowMinB = 3, kSplit3LowSplit = 4;
// fc2 (K = 5120) splits K two ways up to 31 crops: 128 x 256 tiles of K = 5120 are ~400 us blocks, and halving them shortens the ragged
// last round — op level 198 vs 229 us at 16 crops, 547 vs 666 at 40, 596 vs 741 at 48 incl. the reduce
// (profiles/r3af_split3_n1280_tile_splitk_sweep.log); per call 702 vs 658 crops/s at 16 crops, 756 vs 707 at 48 — but 773 vs 784 at 32 and
// 775 vs 784 at 64, whose 240 / 480 tiles fill the rounds anyway and where the LayerNorm kernel then reads two partial planes for nothing
// (profiles/r3ag_split3_fc2_splitk_all_batches.log).  One factor per RANGE (batch invariance): split up to 31 crops, unsplit from 32 on —
// the boundary keeps the reference README's batch of 32 at its best (784) and gives up the 7 % at 40-48 crops.
// proj (K = 1280) splits only up to 15 crops.
constexpr int kSplit3Fc2Split = 2, kSplit3Fc2MaxB = 31;
// decoder + mixer stack: the persistent decoder kernel and the one-workgroup-per-crop mixer kernel win while the work is
// latency-bound (B = 1: 0.96 vs 1.01 ms, B = 64: 1.58 vs 1.93 ms per head); from a few hundred crops on the same products are
// real GEMMs (M = B and M = 160 B rows) and the tiled MFMA kernels win (B = 512: 6.7 vs 7.5 ms) — profiles/r2e_head_fused_vs_chain.log
constexpr int kFusedHeadMaxB = 128;

thread_local std::string g_last_error;

struct Slot {
    size_t off = 0;       // float offset in the weight arena
    int64_t numel = 0;
    bool loaded = false;
};

struct VitBlockW {      // weights of one ViT block (vit.py:128-151), pointers into the weight arena
    const float *n1w, *n1b, *qkvw, *qkvb, *pw, *pb, *n2w, *n2b, *f1w, *f1b, *f2w, *f2b;
};

struct ProfRec {
    int cls;
    double flops, bytes;
    hipEvent_t e0, e1;
};

}  // namespace

struct thmr_engine {
    thmr_config cfg{};
    int vit_depth = 32, dec_depth = 6, max_batch = 0;
    float* warena = nullptr;
    float* sarena = nullptr;
    bool own_w = false, own_s = false;
    size_t wfloats = 0, sfloats = 0;
    std::unordered_map<std::string, Slot> slots;
    std::vector<std::string> required;
    std::vector<VitBlockW> vitw;      // filled by thmr_finalize_weights
    DecParams dec{};                  // decoder weight pointers + scratch, resolved once (finalize)
    MixerParams mix{};                // MLP-Mixer stack weight pointers
    struct HotW {                     // every other weight the default path touches, resolved once (no name hashing per call)
        const float *pe_w, *pe_b, *pos, *lastn_w, *lastn_b, *cls_w, *cls_b, *init_pose, *init_betas, *init_cam;
        const float* conv_b[9];       // biases of the nine k = 3 convs of the VQ decoder, execution order (kConv3)
        const float *res_w[2], *res_b[2];   // the two 1x1 convs of the ResConv blocks
    } hot{};
    bool counted = false;             // registered in the per-device engine count (decoder turnstile)
    bool no_persistent = false;       // THMR_CFG_NO_PERSISTENT: launch-chain head, per-tile split3 GEMMs, no hand-over workspace
    bool legacy_head = false;         // THMR_LEGACY_HEAD=1: force the chain-of-GEMMs head at every batch size (A/B only)
    bool mixer_cluster = true;        // THMR_MIXER_CLUSTER=0: always run the mixer stack as its own one-workgroup-per-crop kernel (A/B only)
    bool tiny_gemm = true;            // THMR_TINY_GEMM=0: the VQ decoder's GEMMs stay on the ring kernel in the small-batch regime (A/B only)
    bool qkv_ring16 = true;           // THMR_QKV_RING16=0: one and two crops keep the 64x64 ring kernel for qkv (A/B only)
    bool attn_keysplit = true;        // THMR_ATTN_KEYSPLIT=0: one and two crops keep the 64-query attention workgroups (A/B only)
    bool attn_b16 = kAttnB16;         // split3 mode: the attention on the bf16 matrix pipe too (csrc/attention_b16.hip); THMR_ATTN_B16=0 / 1: A/B only
    int mid_split_force[2] = {-1, -1};   // THMR_MID_SPLIT=<p><f> (digits 0|2|4): force the split factors of proj and fc2 above 6 crops where the partial-sum buffer allows (A/B only)
    bool smpl_loaded = false, finalized = false;
    // 1 (the DEFAULT since round 5 / ABI 4: what thmr_forward, the facade and bench.py's `value` all run): the four ViT GEMMs, the attention
    // and the decoder's to_kv GEMM of batches of at least kSplit3LowMinB (3) crops run on the bf16 matrix pipe with fp32 operands carried as
    // three bf16 pieces (csrc/gemm_split16.hip, attention_b16.hip).  Engine-owned memory, built by thmr_finalize_weights: the split3 copies of
    // the ViT weights (1.5 x their fp32 size) and the split3 activation operands (M x (1280 + 5120) x 6 bytes).  thmr_set_vit_gemm(0) = the
    // opt-out: exact-fp32 MFMA everywhere.  An engine whose max_batch is below 3 never runs the mode and builds nothing for it.
    int vit_gemm_mode = 1;
    bool split3_small = false;        // THMR_SPLIT3_SMALL=1: the split3 mode also serves up to six crops (ring kernel on split3 operands) — measured SLOWER, A/B only
    int split3_fc2_split = 2;         // THMR_SPLIT3_FC2_SPLIT=1: fc2 of the split3 mode unsplit from 16 crops on (A/B only)
    // round 6: qkv (bit 0), fc1 (bit 1) and proj (bit 2) of few-crop calls as 256 persistent workgroups over the 128 x 128 tile stream (three-stage ring) when
    // that grid is more than one round, at most s3_pn_max tiles, and the 128 x 256 grid would fill its rounds to at most s3_pn_fill per cent (gemm_split16.hip launch_split16_persist narrow); bit-identical to the
    // per-tile kernels.  THMR_SPLIT3_PN_MASK / THMR_SPLIT3_PN_MAX (experiments build)
    // fc1 takes the 128 x 256 stream where its grid is MORE than two rounds and at most s3_pw_fill per cent full (17 / 18 crops: 520 / 540
    // tiles = three rounds of time for 2.03 / 2.1 of work: fc1 -0.43 / -0.75 ms per call, profiles/r6z_*); below two rounds its GELU + split3
    // epilogue (spills in the persistent form) loses what the rounds gain (12 crops: +0.16).  THMR_SPLIT3_PW_FC1=0: off (A/B)
    int s3_pw_fc1 = 1;
    int s3_pw_fill = 72;              // the 128 x 256 stream for qkv when its per-tile grid fills its rounds to at most this many per cent (THMR_SPLIT3_PW_FILL)
    int s3_pn_fill_proj = 60;     // (proj: mask bit 2, OFF — 36-40 crops: proj -0.46 ... -0.59 ms per call, the call as a whole equal: profiles/r6x_*)
    int s3_pn_mask = 11, s3_pn_max = 600, s3_pn_fill = 72, s3_pk_max = 1000;      // mask bit 3: split-K launches (proj / fc2 partial sums) as (tile, K slice) units, up to s3_pk_max units      // s3_pn_fill: use the stream when the 128 x 256 grid fills its rounds to at most this many per cent
    int s3_tile_opts = 0;             // GemmArgs::tile_opts of the split3 GEMMs (THMR_SPLIT3_NARROW8=1 -> 1, THMR_SPLIT3_TAIL8=1 -> 2; A/B only)
    int split3_min_b = 0;             // THMR_SPLIT3_MIN_B=<n>: A/B knob for the smallest batch the split3 mode serves (0 = kSplit3LowMinB)
    char* split_w = nullptr;          // split3 weight copies: shared, reference-counted, among the engines of one weight arena (split_share())
    bool split_w_counted = false;     // this engine holds a reference in split_share()
    char* split_act = nullptr;
    // persistent split3 GEMM (csrc/gemm_split_persist.hip; bit-identical to the one-workgroup-per-tile kernel, so purely a matter of time):
    // hand-over slabs + flags, allocated with the split3 weights on a 256-CU device.  s3_persist: 0 off / 1 on (THMR_SPLIT3_PERSIST);
    // s3_fc1_mode: how fc1's split3 output leaves the persistent kernel — 2 swapped operand roles, 1 LDS transposition, 0 = fc1 stays on
    // the per-tile kernel (THMR_SPLIT3_FC1_MODE)
    void* s3_ws = nullptr;
    int s3_persist = 1, s3_fc1_mode = 2;
    // Which GEMMs run the persistent decomposition: bit 0 qkv, 1 proj, 2 fc1, 3 fc2, 4 the decoder's to_kv.  It pays per tile boundary
    // (hand-over slabs, segment bookkeeping, register spills around its epilogue) and wins the ragged last round: with the 16x16x32
    // kernel it is the faster one for fc2 (K = 5120: 671 vs 715 us) and ties or loses at K = 1280 (qkv 518 vs 515, proj 199 vs 187,
    // fc1 with split3 output 728 vs 702; profiles/r4k_split3_gemm_b64_mfma16.jsonl; whole path 834 vs 822 crops/s with all four,
    // profiles/r4l_engine_b64_persist_min_k_ab.log).  Round 5, same-box interleaved, the whole path at 64 crops (profiles/r5j_ab_mask8_vs_*):
    // + proj +0.33 ms per step (a build whose proj instantiation had NO scratch access in its K loop), + qkv +0.96 ms: at 40 K tiles per
    // tile the hand-over costs more than the ragged round's 6 %.  Same bits either way.  THMR_SPLIT3_PERSIST_MASK (experiments build).
    int s3_persist_mask = 8;
    bool s3_forced_once = false;      // experiments build: THMR_SPLIT3_FORCE_TIMEOUT=1 was honoured already
    struct SplitW { const char *qkv, *proj, *fc1, *fc2; };
    std::vector<SplitW> vitw_s;
    const char* kv_s = nullptr;       // split3 copy of the decoder's stacked to_kv weights (dec_depth * 1024 rows x 1280)
    const char* pe_s = nullptr;       // split3 copy of the patch-embed Conv2d weight as a (1280, 768) matrix
    unsigned* host_err = nullptr;     // host-mapped sticky error words (hipHostMalloc, 64 bytes): [0] the persistent decoder kernel's grid barrier, [1] the persistent split3 GEMM's hand-over
    unsigned* s3_host_err = nullptr;  // = host_err + 1; its ADDRESS is the stable source of the copy that binds it into the hand-over workspace
    std::string err;
    // derived / constant regions (float offsets in weight arena)
    size_t o_kv_all = 0, o_ro_w = 0, o_ro_b = 0;
    size_t o_convp[7] = {0};      // repacked k=3 convs: 0,3,6,9,12, res0.conv1, res1.conv1, 14.1, 15 -> see conv_names
    size_t o_cbT = 0, o_cnorm = 0, o_idx = 0, o_inv = 0;   // idx / inverse idx tables as int32 within the float arena
    size_t o_smpl_vt = 0, o_smpl_sd = 0, o_smpl_pd = 0, o_smpl_jr = 0, o_smpl_w = 0, o_smpl_j19 = 0, o_smpl_int = 0,
           o_smpl_jt = 0, o_smpl_jsd = 0, o_smpl_dirs = 0;
    std::vector<size_t> convp;    // repacked conv offsets, index by conv id
    // optional tokenizer ENCODER (EncodeTokens, vanilla_pose_vqvae.py:304-346): 'encoder.encoder.*' tensors
    std::vector<std::string> enc_names;
    std::vector<size_t> enc_convp;   // repacked encoder convs, index by kEnc id
    size_t o_idx_enc = 0, o_flags = 0;
    bool enc_ready = false;
    int32_t flag_host[2] = {0, 0}, hips_host = 0;
    std::vector<int32_t> idx_host, eidx_host, inv_host;   // staging for the index tables (must outlive the async copy)
    int vq_len[5] = {160, 125, 90, 55, 21};
    // scratch offsets (floats)
    struct {
        size_t x, h, big, part;
        size_t dx, dh, dv, dq, dca, dff, ro;
        size_t mt, cf, cf2, y1, tT, u, yt, y, s, z0, zh, nl, nl2;
        size_t part_floats;       // capacity of `part`
        size_t feat, gat, gat2, act0, act1, act2, bpose, tokidx, sync;
        size_t A, pf, Jtr, vposed, rot, betas, cam, camt, verts, joints, pose6d, xv, lcnt;
        size_t total;
    } so{};
    // profiler
    int prof_on = 0;            // 0 off, 1 every kernel class, 2 only the four ViT GEMM classes, 3 only fc1 (the dominant kernel)
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;

    float* W(const std::string& name) {
        auto it = slots.find(name);
        if (it == slots.end()) { err = "internal: unknown weight " + name; return nullptr; }
        return warena + it->second.off;
    }
    float* S(size_t off) { return sarena + off; }
};

namespace {

size_t align64(size_t f) { return (f + 63) & ~size_t(63); }   // 256-byte alignment in floats

int fail(thmr_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    g_last_error = msg;
    return code;
}

#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(e, THMR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));  \
    } while (0)

#define LAUNCH_OK(call)                                                                       \
    do {                                                                                      \
        int _r = (call);                                                                      \
        if (_r != 0) {                                                                        \
            hipError_t _e = hipGetLastError();                                                \
            return fail(e, _r == -1 ? THMR_ERR_INVALID : THMR_ERR_HIP,                        \
                        std::string(#call) + " failed: " + hipGetErrorString(_e));            \
        }                                                                                     \
    } while (0)

// ---- the reference checkpoint contract, mirrored by tokenhmr_amd/weights.py::spec ----
void build_spec(int vit_depth, int dec_depth, std::vector<std::pair<std::string, int64_t>>& out) {
    auto add = [&](const std::string& n, int64_t numel) { out.emplace_back(n, numel); };
    auto lin = [&](const std::string& n, int64_t o, int64_t i, bool bias = true) {
        add(n + ".weight", o * i);
        if (bias) add(n + ".bias", o);
    };
    auto ln = [&](const std::string& n, int64_t d) { add(n + ".weight", d); add(n + ".bias", d); };
    add("backbone.pos_embed", (TOK + 1) * DIM);
    add("backbone.patch_embed.proj.weight", (int64_t)DIM * 768);
    add("backbone.patch_embed.proj.bias", DIM);
    for (int i = 0; i < vit_depth; ++i) {
        const std::string p = "backbone.blocks." + std::to_string(i) + ".";
        ln(p + "norm1", DIM);
        lin(p + "attn.qkv", 3 * DIM, DIM);
        lin(p + "attn.proj", DIM, DIM);
        ln(p + "norm2", DIM);
        lin(p + "mlp.fc1", MLP, DIM);
        lin(p + "mlp.fc2", DIM, MLP);
    }
    ln("backbone.last_norm", DIM);
    const std::string T = "smpl_head.transformer.";
    add(T + "pos_embedding", E);
    lin(T + "to_token_embedding", E, 1);
    for (int l = 0; l < dec_depth; ++l) {
        const std::string p = T + "transformer.layers." + std::to_string(l) + ".";
        ln(p + "0.norm", E);
        lin(p + "0.fn.to_qkv", 3 * INNER, E, false);
        lin(p + "0.fn.to_out.0", E, INNER);
        ln(p + "1.norm", E);
        lin(p + "1.fn.to_kv", 2 * INNER, DIM, false);
        lin(p + "1.fn.to_q", INNER, E, false);
        lin(p + "1.fn.to_out.0", E, INNER);
        ln(p + "2.norm", E);
        lin(p + "2.fn.net.0", DEC_MLP, E);
        lin(p + "2.fn.net.3", E, DEC_MLP);
    }
    lin("smpl_head.decpose_grot", 6, E);
    lin("smpl_head.decshape", 10, E);
    lin("smpl_head.deccam", 3, E);
    lin("smpl_head.decpose_hands", 12, E);
    const std::string C = "smpl_head.decpose.";
    lin(C + "mixer_trans.ff.0", (int64_t)TN * HID, E);
    ln(C + "mixer_trans.ff.1", (int64_t)TN * HID);
    for (int m = 0; m < MIX; ++m) {
        const std::string p = C + "mixer_head." + std::to_string(m) + ".";
        ln(p + "layernorm1", HID);
        lin(p + "MLP_token.ff.0", TOK_INTER, TN);
        lin(p + "MLP_token.ff.3", TN, TOK_INTER);
        ln(p + "layernorm2", HID);
        lin(p + "MLP_channel.ff.0", HID_INTER, HID);
        lin(p + "MLP_channel.ff.3", HID, HID_INTER);
    }
    lin(C + "mixer_norm_layer.ff.0", HID, HID);
    ln(C + "mixer_norm_layer.ff.1", HID);
    lin(C + "class_pred_layer", NCLS, HID);
    add("smpl_head.init_body_pose", 144);
    add("smpl_head.init_betas", 10);
    add("smpl_head.init_cam", 3);
    // tokenizer.pth ['net']
    auto conv = [&](const std::string& n, int64_t co, int64_t ci, int64_t k) { add(n + ".weight", co * ci * k); add(n + ".bias", co); };
    conv("decoder.decoder.0", VQW, CODE, 3);
    for (int i : {3, 6, 9, 12}) conv("decoder.decoder." + std::to_string(i), VQW, VQW, 3);
    for (int b : {0, 1}) {
        conv("decoder.decoder.14.0.model." + std::to_string(b) + ".conv1", VQW, VQW, 3);
        conv("decoder.decoder.14.0.model." + std::to_string(b) + ".conv2", VQW, VQW, 1);
    }
    conv("decoder.decoder.14.1", VQW, VQW, 3);
    conv("decoder.decoder.15", 6, VQW, 3);
    add("quantizer.codebook", (int64_t)NCLS * CODE);
}

// k=3 convs that need the [co][k*ci+ci] repack, in execution order
const char* const kConv3[] = {"decoder.decoder.0", "decoder.decoder.3", "decoder.decoder.6", "decoder.decoder.9",
                              "decoder.decoder.12", "decoder.decoder.14.0.model.0.conv1",
                              "decoder.decoder.14.0.model.1.conv1", "decoder.decoder.14.1", "decoder.decoder.15"};
const int kConv3Ci[] = {CODE, VQW, VQW, VQW, VQW, VQW, VQW, VQW, VQW};
const int kConv3Co[] = {VQW, VQW, VQW, VQW, VQW, VQW, VQW, VQW, 6};

// tokenizer encoder convs in execution order (PoseSPEncoderV1, vanilla_pose_vqvae.py:66-88 with the release ARCH:
// input_dim 6, width 512, token_size_mul 4, down_t 1, stride_t 2, depth 2, dilation 3, code_dim 256)
struct EncConv { const char* name; int ci, cp, co, ks; };
const EncConv kEnc[] = {
    {"encoder.encoder.0", 6, 32, VQW, 3},                      // T=21, +ReLU
    {"encoder.encoder.3", VQW, VQW, VQW, 3},                   // nearest 21->40, +ReLU
    {"encoder.encoder.6", VQW, VQW, VQW, 3},                   // x2 -> 80
    {"encoder.encoder.9", VQW, VQW, VQW, 3},                   // x2 -> 160
    {"encoder.encoder.12", VQW, VQW, VQW, 3},                  // x2 -> 320
    {"encoder.encoder.14.0", VQW, VQW, VQW, 4},                // k4 s2 p1: 320 -> 160 (no activation)
    {"encoder.encoder.14.1.model.0.conv1", VQW, VQW, VQW, 3},  // ResConv1DBlock dil 3
    {"encoder.encoder.14.1.model.0.conv2", VQW, VQW, VQW, 1},
    {"encoder.encoder.14.1.model.1.conv1", VQW, VQW, VQW, 3},  // dil 1
    {"encoder.encoder.14.1.model.1.conv2", VQW, VQW, VQW, 1},
    {"encoder.encoder.15", VQW, VQW, CODE, 3},                 // -> (B,160,256)
};
constexpr int kEncN = 11;

void layout_weights(thmr_engine* e) {
    std::vector<std::pair<std::string, int64_t>> spec;
    build_spec(e->vit_depth, e->dec_depth, spec);
    size_t off = 0;
    // contiguous groups first: to_kv of all layers -> one (dec_depth*1024, 1280) matrix; read-outs -> (31,1024)+(31)
    e->o_kv_all = off;
    for (int l = 0; l < e->dec_depth; ++l) {
        const std::string n = "smpl_head.transformer.transformer.layers." + std::to_string(l) + ".1.fn.to_kv.weight";
        e->slots[n] = Slot{off, (int64_t)2 * INNER * DIM, false};
        off += (size_t)2 * INNER * DIM;
    }
    off = align64(off);
    e->o_ro_w = off;
    const char* ro_names[] = {"smpl_head.decpose_grot", "smpl_head.decshape", "smpl_head.deccam", "smpl_head.decpose_hands"};
    const int ro_n[] = {6, 10, 3, 12};
    for (int i = 0; i < 4; ++i) {
        e->slots[std::string(ro_names[i]) + ".weight"] = Slot{off, (int64_t)ro_n[i] * E, false};
        off += (size_t)ro_n[i] * E;
    }
    off += E;   // 32nd (padding) row so a clamped row read stays inside the arena
    off = align64(off);
    e->o_ro_b = off;
    for (int i = 0; i < 4; ++i) {
        e->slots[std::string(ro_names[i]) + ".bias"] = Slot{off, ro_n[i], false};
        off += ro_n[i];
    }
    off = align64(off + 1);
    for (auto& kv : spec) {
        e->required.push_back(kv.first);
        if (e->slots.count(kv.first)) continue;
        e->slots[kv.first] = Slot{off, kv.second, false};
        off = align64(off + (size_t)kv.second);
    }
    // derived regions
    e->convp.resize(9);
    for (int i = 0; i < 9; ++i) {
        e->convp[i] = off;
        off = align64(off + (size_t)kConv3Co[i] * kConv3Ci[i] * 3);
    }
    e->o_cbT = off;   off = align64(off + (size_t)NCLS * CODE);
    e->o_cnorm = off; off = align64(off + NCLS);
    e->o_idx = off;   off = align64(off + 4 * 160);
    e->o_inv = off;   off = align64(off + 4 * 160);
    // SMPL constants
    e->o_smpl_vt = off;  off = align64(off + (size_t)NV * 3);
    e->o_smpl_sd = off;  off = align64(off + (size_t)NV * 30);
    e->o_smpl_pd = off;  off = align64(off + (size_t)NP * NV * 3);
    e->o_smpl_jr = off;  off = align64(off + (size_t)NJ * NV);
    e->o_smpl_w = off;   off = align64(off + (size_t)NV * NJ);
    e->o_smpl_j19 = off; off = align64(off + (size_t)19 * NV);
    e->o_smpl_int = off; off = align64(off + 128);       // parents(24) | extra(21) | jmap(25) as int32
    e->o_smpl_jt = off;  off = align64(off + NJ * 3);
    e->o_smpl_jsd = off; off = align64(off + NJ * 30);
    e->o_smpl_dirs = off; off = align64(off + (size_t)NV * 3 * THMR_LBS_KX);   // [shapedirs | posedirs | 0]^T, derived
    // optional tokenizer encoder (tokenizer.pth 'encoder.encoder.*'): raw tensors, repacked convs, resample tables
    for (int i = 0; i < kEncN; ++i) {
        const std::string n = kEnc[i].name;
        e->enc_names.push_back(n + ".weight");
        e->enc_names.push_back(n + ".bias");
        e->slots[n + ".weight"] = Slot{off, (int64_t)kEnc[i].co * kEnc[i].ci * kEnc[i].ks, false};
        off = align64(off + (size_t)kEnc[i].co * kEnc[i].ci * kEnc[i].ks);
        e->slots[n + ".bias"] = Slot{off, kEnc[i].co, false};
        off = align64(off + kEnc[i].co);
    }
    e->enc_convp.resize(kEncN);
    for (int i = 0; i < kEncN; ++i) {
        e->enc_convp[i] = off;
        if (kEnc[i].ks > 1) off = align64(off + (size_t)kEnc[i].co * kEnc[i].cp * kEnc[i].ks);
    }
    e->o_idx_enc = off; off = align64(off + 640);
    e->o_flags = off;   off = align64(off + 64);     // int32 flag words travelling with the arena (0: encoder present)
    e->wfloats = off;
}

void layout_scratch(thmr_engine* e) {
    const size_t B = (size_t)e->max_batch, M = B * TOK;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align64(off + n); return o; };
    auto& s = e->so;
    s.x = take(M * DIM);
    s.h = take(M * DIM);
    s.big = take(M * 6144);
    {   // split-K partial sums of the proj / fc2 GEMMs: 4 x M x 1280 up to kSmallM rows (ring kernel), 2 x M x 1280 up to kMidHiM rows
        const size_t m2 = M < (size_t)kMidHiM ? M : (size_t)kMidHiM, need = (size_t)kMidSplit * m2;
        s.part_floats = (need > (size_t)kSplitKMax * kSmallM ? need : (size_t)kSplitKMax * kSmallM) * DIM;
        s.part = take(s.part_floats);
    }
    s.dx = take(B * E); s.dh = take(B * E); s.dv = take(B * INNER); s.dq = take(B * INNER); s.dca = take(B * INNER);
    s.dff = take(B * DEC_MLP); s.ro = take(B * 32);
    s.mt = take(B * TN * HID); s.cf = take(B * TN * HID); s.cf2 = take(B * TN * HID);
    s.y1 = take(B * TN * HID); s.tT = take(B * HID * TN); s.u = take(B * HID * TOK_INTER); s.yt = take(B * HID * TN);
    s.y = take(B * TN * HID); s.s = take(B * TN * HID); s.z0 = take(B * TN * HID); s.zh = take(B * TN * HID_INTER);
    s.nl = take(B * TN * HID); s.nl2 = take(B * TN * HID);
    s.feat = take(B * TN * CODE);
    s.gat = take(B * 125 * 3 * VQW);                 // largest conv operand: T=125, 3*512 (> 160*768)
    s.gat2 = take(B * 125 * 3 * VQW);                // the GEMM of conv i writes the operand of conv i+1: two buffers alternate
    s.act0 = take(B * TN * VQW); s.act1 = take(B * TN * VQW); s.act2 = take(B * TN * VQW);
    s.bpose = take(B * 128); s.tokidx = take(B * TN); s.sync = take(512);
    s.A = take(B * NJ * 12); s.pf = take(B * THMR_LBS_XF); s.Jtr = take(B * NJ * 3); s.vposed = take(B * NV * 3);
    s.rot = take(B * NJ * 9); s.betas = take(B * NB); s.cam = take(B * 3); s.camt = take(B * 3);
    s.verts = take(B * NV * 3); s.joints = take(B * 132); s.pose6d = take(B * 144);
    s.xv = take(B * 63); s.lcnt = take(B);
    s.total = off;
    e->sfloats = off;
}

// ---- profiler helpers ----
struct ProfScope {
    thmr_engine* e;
    hipStream_t s;
    bool on;
    size_t rec = 0;
    // `sampled`: mode 3 (only the dominant kernel, live in the timed region) instruments every 4th fc1 launch — all 32 per call have
    // the same shape, and 32 event pairs are still 5 % of a one-crop call
    ProfScope(thmr_engine* e_, hipStream_t s_, int cls, double flops, double bytes, bool sampled = true) : e(e_), s(s_), on(e_->prof_on == 1 || (e_->prof_on == 2 && cls <= THMR_PROF_GEMM_FC2) || (e_->prof_on == 3 && cls == THMR_PROF_GEMM_FC1 && sampled)) {
        if (!on) return;
        if (e->ev_next + 2 > e->ev_pool.size()) {
            for (int i = 0; i < 512; ++i) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) { on = false; return; }
                e->ev_pool.push_back(ev);
            }
        }
        ProfRec r{cls, flops, bytes, e->ev_pool[e->ev_next], e->ev_pool[e->ev_next + 1]};
        e->ev_next += 2;
        (void)hipEventRecord(r.e0, s);
        e->prof.push_back(r);
        rec = e->prof.size() - 1;
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(e->prof[rec].e1, s);
    }
};

GemmArgs mk(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
            float* C, int64_t ldc, int M, int N, int K) {
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.resid = resid; a.C = C;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K; a.qscale = 1.f; a.qcols = 0;
    return a;
}

int launch_decoder_serialised(thmr_engine* e, const DecParams& d, hipStream_t st);   // below (turnstile of the persistent kernels)
int launch_split3_persist_serialised(thmr_engine* e, const GemmArgs& a, int epi, int mode, hipStream_t st);

// ---------------------------------------------------------------------------------------------- ViT-H
int vit_forward(thmr_engine* e, const float* img, int B, float* feats_out, hipStream_t st) {
    const int M = B * TOK;
    float* x = e->S(e->so.x);
    float* h = e->S(e->so.h);
    float* big = e->S(e->so.big);
    const bool s3_on = e->vit_gemm_mode == 1 && B >= (e->split3_min_b > 0 ? e->split3_min_b : kSplit3LowMinB) && e->split_w && e->split_act;
    {   // patch embed: crop + pad + im2col, then GEMM (+bias, +pos_embed)   vit.py:341,170-176,327
        ProfScope ps(e, st, THMR_PROF_PATCH, 2.0 * M * 768.0 * DIM,
                     4.0 * (B * 3.0 * 256 * 192 + (double)M * DIM + 768.0 * DIM));
        if (s3_on && e->pe_s) {
            // default mode: the im2col operand written as three bf16 pieces (into the idle fc1 -> fc2 operand buffer) and the Conv2d as a
            // split3 product on the bf16 matrix pipe with the same bias + pos_embed epilogue (0.23 -> ~0.13 ms at 64 crops)
            char* ims = e->split_act + (size_t)M * DIM * 6;
            LAUNCH_OK(launch_im2col_patch_split3(img, ims, B, st));
            GemmArgs a = mk(reinterpret_cast<const float*>(ims), 768, reinterpret_cast<const float*>(e->pe_s), 768, e->hot.pe_b, e->hot.pos, 0, x, DIM, M, DIM, 768);
            a.tile_opts = e->s3_tile_opts;
            LAUNCH_OK(launch_gemm_split3(a, EPI_BIAS_POS, -1, st));
        } else {
            LAUNCH_OK(launch_im2col_patch(img, big, B, st));
            GemmArgs a = mk(big, 768, e->hot.pe_w, 768, e->hot.pe_b, e->hot.pos, 0, x, DIM, M, DIM, 768);
            LAUNCH_OK(launch_gemm(a, EPI_BIAS_POS, -1, st));
        }
    }
    const float qscale = 1.0f / sqrtf(80.0f);   // head_dim ** -0.5  (vit.py:101)
    // Few crops (M <= kSmallM): the N = 1280 GEMMs run split-K on the 64x64 ring kernel and their partial sums are reduced
    // inside the residual + LayerNorm kernel that follows them anyway; qkv / fc1 use the ring kernel up to M = 384.
    const bool small = M <= kSmallM, ring_wide = M <= 384;
    // one split factor for the whole regime, so a crop's result does not depend on how many crops share its batch (B <= 6)
    const int ks_proj = kSplitKMax, ks_fc2 = kSplitKMax;
    float* part = e->S(e->so.part);
    // mid-size batches (7 ... 16 crops): proj / fc2 split K two ways on the big tiles, reduced by the same residual + LayerNorm
    // kernel; ONE factor for the whole range and both GEMMs, so a crop's result does not depend on the batch it rides in within it
    auto pick = [&](int forced, int rule) {
        int sp = small ? 1 : forced >= 0 ? (forced > 1 ? forced : 1) : rule;
        if ((size_t)sp * M * DIM > e->so.part_floats) sp = 1;      // only reachable with the A/B knob
        return sp;
    };
    const int rule = (M >= kMidLoM && M <= kMidHiM) ? kMidSplit : 1;
    const int mid_proj = pick(e->mid_split_force[0], rule), mid_fc2 = pick(e->mid_split_force[1], rule);
    // x += Linear(A) + bias;  y = LayerNorm(x)      (vit.py:149 / :150 followed by the next norm)
    auto resid_linear_ln = [&](int cls, const float* A, int K, const float* Wt, const float* bias, int ks, int mid_split, const float* g,
                               const float* bt, float* y) -> int {
        const double fl = 2.0 * M * DIM * (double)K, by = 4.0 * ((double)M * K + (double)DIM * K + 2.0 * M * DIM);
        if (small) {
            {
                ProfScope ps(e, st, cls, fl, by);
                GemmArgs a = mk(A, K, Wt, K, nullptr, nullptr, 0, x, DIM, M, DIM, K);
                LAUNCH_OK(launch_gemm_ring(a, EPI_NONE, 4, ks, part, st));
            }
            ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (ks + 3.0) * M * DIM);
            LAUNCH_OK(launch_splitk_resid_ln(part, ks, M, DIM, bias, x, x, g, bt, y, VIT_EPS, st));
        } else if (mid_split > 1) {
            {
                ProfScope ps(e, st, cls, fl, by);
                GemmArgs a = mk(A, K, Wt, K, nullptr, nullptr, 0, x, DIM, M, DIM, K);
                LAUNCH_OK(launch_gemm_splitk(a, -1, mid_split, part, st));
            }
            ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (mid_split + 3.0) * M * DIM);
            LAUNCH_OK(launch_splitk_resid_ln(part, mid_split, M, DIM, bias, x, x, g, bt, y, VIT_EPS, st));
        } else {
            {
                ProfScope ps(e, st, cls, fl, by);
                GemmArgs a = mk(A, K, Wt, K, bias, x, DIM, x, DIM, M, DIM, K);
                LAUNCH_OK(launch_gemm(a, EPI_BIAS_RESID, -1, st));
            }
            ProfScope ps(e, st, THMR_PROF_LN, 0, 8.0 * M * DIM);
            LAUNCH_OK(launch_layernorm(x, g, bt, y, M, DIM, VIT_EPS, 0, st));
        }
        return 0;
    };
    const float* lastn_w = e->hot.lastn_w;
    const float* lastn_b = e->hot.lastn_b;
    if (s3_on) {
        // 3 ... 4 / 5 ... 15 crops: proj / fc2 split K four / two ways into `part`, reduced (in a fixed order) by the residual + LayerNorm kernel,
        // as in the exact-fp32 path's regimes; 16 ... 31: only fc2 (two ways); 32 and more: unsplit.  One factor per range: a crop's result is
        // batch-independent within it.
        const bool s3_low = B < kSplit3MidMinB;                  // 3 and 4 crops: both N = 1280 GEMMs four ways
        const int s3_split = s3_low ? kSplit3LowSplit : B < kSplit3MinB ? kSplit3MidSplit : 1;
        const int s3_fc2 = s3_low ? kSplit3LowSplit : B <= kSplit3Fc2MaxB ? e->split3_fc2_split : 1;
        // fc2's partial sums: the engine-owned planes behind the operand buffers (two planes for any batch size); the four planes of
        // 3 and 4 crops fit the scratch arena's `part` (4 x 1152 rows)
        float* part2 = s3_low ? part : reinterpret_cast<float*>(e->split_act + (size_t)e->max_batch * TOK * (DIM + MLP) * 6);
        // The four GEMMs as split3 products on the bf16 matrix pipe (csrc/gemm_split.hip); everything else — patch embed, attention,
        // LayerNorm arithmetic, epilogues — is the fp32 path's.  A operands: the LayerNorms, the attention kernel and fc1's GELU epilogue
        // write their results directly as three bf16 pieces (hs, bs): no conversion pass, no fp32 copy of those activations.
        char* hs = e->split_act;                                    // [M][1280] split3: LayerNorm / attention output
        char* bs = e->split_act + (size_t)M * DIM * 6;              // [M][5120] split3: GELU output
        // fc1's output = fc2's A in the ROW-BLOCKED form (common.h GemmArgs::a_blk) when fc2 runs the persistent kernel: the epilogue (one output
        // row per lane) then writes 256-512 contiguous bytes per 16-32 lanes instead of a different line per lane (same box: 830-832 -> 846-848
        // crops/s at 64 crops, profiles/r4h_row_blocked_ab_same_box_b64.log; with the 16x16x32 kernel fc1 718 -> 702 us and fc2 677 -> 671,
        // r4k_split3_gemm_b64_mfma16.jsonl).  A per-tile fc2 measured SLOWER with a blocked A (778 vs 715 us) and keeps the row-major form
        // (fewer than 32 crops, odd batches).  Same values either way.
        int bs_blk = 0;
        {
            GemmArgs t1 = mk(nullptr, DIM, nullptr, DIM, nullptr, nullptr, 0, nullptr, 0, M, MLP, DIM), t2 = mk(nullptr, MLP, nullptr, MLP, nullptr, nullptr, 0, nullptr, 0, M, DIM, MLP);
            (void)t1;
            bs_blk = (e->s3_ws && e->s3_persist && s3_fc2 <= 1 && (e->s3_persist_mask & 8) && gemm_split3_persist_ok(t2)) ? 1 : 0;      // fc2 runs the persistent kernel
            static const bool no_blk = [] { const char* k = thmr_knob("THMR_SPLIT3_BS_BLK"); return k && k[0] == '0'; }();      // A/B (experiments build)
            if (no_blk) bs_blk = 0;
        }
        // The 128 x 128 grid is more than one round of 256 workgroups, and the 128 x 256 grid it would otherwise run fills its rounds badly:
        // then the stream over 128 x 128 tiles wins (a K tile costs it ~1.3 us + ~10 us per launch, against 2.15 us per wide K tile and whole
        // rounds).  Measured per class, same box (profiles/r6k_*): qkv at 6 / 12 crops (135 / 270 wide tiles, rounds 53 % full) 2.75 -> 2.13 and
        // 4.87 -> 3.64 ms per call, fc1 at 6 / 10 crops (70 % / 59 %) 2.92 -> 2.63 and 5.10 -> 3.94; at 8 crops qkv (70 %) 2.70 -> 2.62; it
        // LOSES where the wide rounds are full (fc1 at 8 crops, 94 %: +0.26; qkv at 10, 88 %: +0.34) and from ~700 tiles on (qkv at 16: +0.25).
        auto narrow_stream = [&](const GemmArgs& a, int fill = 0) {
            const long rows = (a.M + 127) / 128, t128 = rows * ((a.N + 127) / 128), wide = rows * ((a.N + 255) / 256);
            const long rounds = (wide + 255) / 256;
            return e->s3_ws && e->s3_persist && t128 > 256 && t128 <= e->s3_pn_max && 100 * wide <= (fill ? fill : e->s3_pn_fill) * 256 * rounds &&
                   gemm_split3_persist_narrow_ok(a);
        };
        // ... and the stream over 128 x 256 tiles (round 4's persistent kernel; until round 6 fc2's only, where it wins at every size) for qkv
        // too WHEN its one-workgroup-per-tile grid fills its rounds badly and the 128 x 128 stream above does not apply: at 64 crops (94 % full)
        // the hand-overs cost qkv +6 %; at 24 crops its 540 tiles are 2.1 rounds = three rounds of time: 7.49 -> 6.93 ms per call, at 14 crops
        // (315 tiles, 62 %) 5.02 -> 4.45 (profiles/r6m_*).  fc1 only above two rounds (s3_pw_fc1); not proj: measured equal or slower (+0.12 at 36)
        auto wide_stream = [&](const GemmArgs& a, long min_tiles = 256) {
            const long wide = (long)((a.M + 127) / 128) * ((a.N + 255) / 256), rounds = (wide + 255) / 256;
            return e->s3_ws && e->s3_persist && wide >= min_tiles && 100 * wide <= e->s3_pw_fill * 256 * rounds && gemm_split3_persist_ok(a);
        };
        // split-K launches (proj / fc2 below 16 / 32 crops) through the same stream: units = (tile, K slice); taken where the grid the rule would
        // launch fills its rounds badly — e.g. fc2 at 18 crops = 270 workgroups of 128 x 256 x (K / 2) = two rounds for 1.05 rounds of work
        auto splitk_stream = [&](const GemmArgs& a, int ks) {
            const long rows = (a.M + 127) / 128, units = rows * ((a.N + 127) / 128) * ks, wide = rows * ((a.N + 255) / 256) * ks;
            const long rounds = (wide + 255) / 256;
            GemmArgs t = a;
            t.ksplit = ks; t.bias = nullptr; t.resid = nullptr;
            return e->s3_ws && e->s3_persist && (e->s3_pn_mask & 8) && units > 256 && units <= e->s3_pk_max && 100 * wide <= e->s3_pn_fill * 256 * rounds &&
                   gemm_split3_persist_narrow_ok(t);
        };
        auto gemm_s = [&](int cls, const char* A, int K, const char* Wt, const float* bias, const float* resid, float* C, int N, int epi, int a_blk = 0) -> int {
            ProfScope ps(e, st, cls, 2.0 * M * (double)N * K, 6.0 * ((double)M * K + (double)N * K) + 4.0 * M * N * (resid ? 2.0 : 1.0));
            GemmArgs a = mk(reinterpret_cast<const float*>(A), K, reinterpret_cast<const float*>(Wt), K, bias, resid, N, C, N, M, N, K);
            a.qscale = qscale; a.qcols = DIM;
            a.a_blk = a_blk;
            a.tile_opts = e->s3_tile_opts;
            const int bit = cls == THMR_PROF_GEMM_QKV ? 1 : cls == THMR_PROF_GEMM_PROJ ? 2 : cls == THMR_PROF_GEMM_FC2 ? 8 : 0;
            if (e->s3_ws && e->s3_persist && (e->s3_persist_mask & bit) && gemm_split3_persist_ok(a)) return launch_split3_persist_serialised(e, a, epi, 0, st);
            if (cls == THMR_PROF_GEMM_QKV && (e->s3_pn_mask & 1) && narrow_stream(a)) return launch_gemm_split3_persist_narrow(a, epi, e->s3_ws, st);
            // proj (unsplit from 16 crops on; K = 1280, 5 wide column tiles): only where its wide rounds are at most s3_pn_fill_proj per cent full
            if (cls == THMR_PROF_GEMM_PROJ && (e->s3_pn_mask & 4) && narrow_stream(a, e->s3_pn_fill_proj)) return launch_gemm_split3_persist_narrow(a, epi, e->s3_ws, st);
            if (cls == THMR_PROF_GEMM_QKV && wide_stream(a)) return launch_split3_persist_serialised(e, a, epi, 0, st);
            return launch_gemm_split3(a, epi, -1, st);
        };
        {
            ProfScope ps(e, st, THMR_PROF_LN, 0, 10.0 * M * DIM);
            LAUNCH_OK(launch_layernorm_split3(x, e->vitw[0].n1w, e->vitw[0].n1b, hs, M, DIM, VIT_EPS, st));
        }
        for (int i = 0; i < e->vit_depth; ++i) {
            const VitBlockW& w = e->vitw[i];
            const thmr_engine::SplitW& ws = e->vitw_s[i];
            const bool last = i + 1 == e->vit_depth;
            LAUNCH_OK(gemm_s(THMR_PROF_GEMM_QKV, hs, DIM, ws.qkv, w.qkvb, nullptr, big, 3 * DIM, EPI_BIAS_QSCALE));
            {   // attention, its output written directly as proj's split3 operand
                ProfScope ps(e, st, THMR_PROF_ATTN, 4.0 * B * HEADS * 192.0 * 192.0 * 80.0, 4.0 * (3.0 * M * DIM) + 6.0 * M * DIM);
                if (e->attn_b16) LAUNCH_OK(launch_vit_attention_b16(big, hs, B, true, 0, st));
                else LAUNCH_OK(launch_vit_attention_split3(big, hs, B, st));
            }
            if (s3_split > 1) {
                {
                    ProfScope ps(e, st, THMR_PROF_GEMM_PROJ, 2.0 * M * DIM * (double)DIM, 6.0 * ((double)M * DIM + (double)DIM * DIM) + 4.0 * s3_split * M * DIM);
                    GemmArgs a = mk(reinterpret_cast<const float*>(hs), DIM, reinterpret_cast<const float*>(ws.proj), DIM, nullptr, nullptr, 0, x, DIM, M, DIM, DIM);
                    a.tile_opts = e->s3_tile_opts;
                    if (splitk_stream(a, s3_split)) LAUNCH_OK(launch_gemm_split3_splitk_stream(a, s3_split, part, e->s3_ws, st));
                    else LAUNCH_OK(launch_gemm_split3_splitk(a, s3_split, part, st));
                }
                ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (s3_split + 2.0) * M * DIM + 6.0 * M * DIM);
                LAUNCH_OK(launch_splitk_resid_ln(part, s3_split, M, DIM, w.pb, x, x, w.n2w, w.n2b, reinterpret_cast<float*>(hs), VIT_EPS, st, true));
            } else {
                LAUNCH_OK(gemm_s(THMR_PROF_GEMM_PROJ, hs, DIM, ws.proj, w.pb, x, x, DIM, EPI_BIAS_RESID));
                ProfScope ps(e, st, THMR_PROF_LN, 0, 10.0 * M * DIM);
                LAUNCH_OK(launch_layernorm_split3(x, w.n2w, w.n2b, hs, M, DIM, VIT_EPS, st));
            }
            {   // fc1 + exact GELU, written directly as fc2's split3 operand (no fp32 copy of the hidden activations exists)
                ProfScope ps(e, st, THMR_PROF_GEMM_FC1, 2.0 * M * DIM * (double)MLP, 6.0 * ((double)M * DIM + (double)DIM * MLP + (double)M * MLP));
                GemmArgs a = mk(reinterpret_cast<const float*>(hs), DIM, reinterpret_cast<const float*>(ws.fc1), DIM, w.f1b, nullptr, 0, nullptr, 0, M, MLP, DIM);
                a.c_split = bs; a.ldcs = MLP;
                a.cs_blk = bs_blk;
                a.tile_opts = e->s3_tile_opts;
                if (e->s3_ws && e->s3_persist && e->s3_fc1_mode && ((e->s3_persist_mask & 4) || (e->s3_pw_fc1 && wide_stream(a, 512))) && gemm_split3_persist_ok(a))
                    LAUNCH_OK(launch_split3_persist_serialised(e, a, EPI_BIAS_GELU, 2, st));
                else if ((e->s3_pn_mask & 2) && narrow_stream(a))
                    LAUNCH_OK(launch_gemm_split3_persist_narrow(a, EPI_BIAS_GELU, e->s3_ws, st));
                else
                    LAUNCH_OK(launch_gemm_split3(a, EPI_BIAS_GELU, -1, st));
            }
            if (s3_fc2 > 1) {
                {
                    ProfScope ps(e, st, THMR_PROF_GEMM_FC2, 2.0 * M * DIM * (double)MLP, 6.0 * ((double)M * MLP + (double)DIM * MLP) + 4.0 * s3_fc2 * M * DIM);
                    GemmArgs a = mk(reinterpret_cast<const float*>(bs), MLP, reinterpret_cast<const float*>(ws.fc2), MLP, nullptr, nullptr, 0, x, DIM, M, DIM, MLP);
                    a.a_blk = bs_blk;
                    a.tile_opts = e->s3_tile_opts;
                    if (!a.a_blk && splitk_stream(a, s3_fc2)) LAUNCH_OK(launch_gemm_split3_splitk_stream(a, s3_fc2, part2, e->s3_ws, st));
                    else LAUNCH_OK(launch_gemm_split3_splitk(a, s3_fc2, part2, st));
                }
                ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (s3_fc2 + 3.0) * M * DIM);
                if (last)
                    LAUNCH_OK(launch_splitk_resid_ln(part2, s3_fc2, M, DIM, w.f2b, x, x, lastn_w, lastn_b, feats_out ? feats_out : h, VIT_EPS, st));
                else
                    LAUNCH_OK(launch_splitk_resid_ln(part2, s3_fc2, M, DIM, w.f2b, x, x, e->vitw[i + 1].n1w, e->vitw[i + 1].n1b,
                                                     reinterpret_cast<float*>(hs), VIT_EPS, st, true));
            } else {
                LAUNCH_OK(gemm_s(THMR_PROF_GEMM_FC2, bs, MLP, ws.fc2, w.f2b, x, x, DIM, EPI_BIAS_RESID, bs_blk));
                ProfScope ps(e, st, THMR_PROF_LN, 0, 10.0 * M * DIM);
                if (last) LAUNCH_OK(launch_layernorm(x, lastn_w, lastn_b, feats_out ? feats_out : h, M, DIM, VIT_EPS, 0, st));
                else LAUNCH_OK(launch_layernorm_split3(x, e->vitw[i + 1].n1w, e->vitw[i + 1].n1b, hs, M, DIM, VIT_EPS, st));
            }
        }
        return 0;
    }
#ifdef THMR_EXPERIMENTS
    if (e->vit_gemm_mode == 1 && small && e->split3_small) {
        // EXPERIMENT (THMR_SPLIT3_SMALL=1, off by default): a small-batch regime (up to six crops) of the split3 mode — the ring kernel
        // on split3 operands (64 x 64 tiles, 4-deep LDS-DMA ring; proj / fc2 split K four ways into `part`, reduced by the residual +
        // LayerNorm kernel as in the fp32 regime), producers writing split3 operands directly.  A wave's MFMA chain per K tile shrinks from
        // 16 x 64 to 12 x 32 cycles, but the call gets SLOWER: 4.28 vs 3.89 ms at one crop, 6.88 vs 5.84 at two, 16.3 vs 14.1 at six
        // (profiles/r3y_split3_small_batch_regime_ab.log).  At these sizes the GEMMs are bound by the bytes a CU can keep in flight
        // (three 24 KB stages) against a ~4 us loaded memory round trip, not by the matrix pipe, and split3 operands are 1.5x the bytes.
        char* hs = e->split_act;
        char* bs = e->split_act + (size_t)M * DIM * 6;
        auto sgemm = [&](const char* A, int K, const char* Wt, const float* bias, float* C, int N) {
            return mk(reinterpret_cast<const float*>(A), K, reinterpret_cast<const float*>(Wt), K, bias, nullptr, 0, C, N, M, N, K);
        };
        {
            ProfScope ps(e, st, THMR_PROF_LN, 0, 10.0 * M * DIM);
            LAUNCH_OK(launch_layernorm_split3(x, e->vitw[0].n1w, e->vitw[0].n1b, hs, M, DIM, VIT_EPS, st));
        }
        for (int i = 0; i < e->vit_depth; ++i) {
            const VitBlockW& w = e->vitw[i];
            const thmr_engine::SplitW& ws = e->vitw_s[i];
            const bool last = i + 1 == e->vit_depth;
            {
                ProfScope ps(e, st, THMR_PROF_GEMM_QKV, 2.0 * M * DIM * 3.0 * DIM, 6.0 * ((double)M * DIM + 3.0 * DIM * DIM) + 12.0 * M * DIM);
                GemmArgs a = sgemm(hs, DIM, ws.qkv, w.qkvb, big, 3 * DIM);
                a.qscale = qscale; a.qcols = DIM;
                LAUNCH_OK(launch_gemm_split3_ring(a, EPI_BIAS_QSCALE, 1, nullptr, st));
            }
            {
                ProfScope ps(e, st, THMR_PROF_ATTN, 4.0 * B * HEADS * 192.0 * 192.0 * 80.0, 4.0 * (3.0 * M * DIM) + 6.0 * M * DIM);
                LAUNCH_OK(launch_vit_attention_split3(big, hs, B, st));
            }
            {
                ProfScope ps(e, st, THMR_PROF_GEMM_PROJ, 2.0 * M * DIM * (double)DIM, 6.0 * ((double)M * DIM + (double)DIM * DIM) + 4.0 * kSplitKMax * M * DIM);
                GemmArgs a = sgemm(hs, DIM, ws.proj, nullptr, x, DIM);
                LAUNCH_OK(launch_gemm_split3_ring(a, EPI_NONE, kSplitKMax, part, st));
            }
            {
                ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (kSplitKMax + 2.0) * M * DIM + 6.0 * M * DIM);
                LAUNCH_OK(launch_splitk_resid_ln(part, kSplitKMax, M, DIM, w.pb, x, x, w.n2w, w.n2b, reinterpret_cast<float*>(hs), VIT_EPS, st, true));
            }
            {
                ProfScope ps(e, st, THMR_PROF_GEMM_FC1, 2.0 * M * DIM * (double)MLP, 6.0 * ((double)M * DIM + (double)DIM * MLP + (double)M * MLP), (i & 3) == 0);
                GemmArgs a = sgemm(hs, DIM, ws.fc1, w.f1b, nullptr, MLP);
                a.c_split = bs; a.ldcs = MLP;
                LAUNCH_OK(launch_gemm_split3_ring(a, EPI_BIAS_GELU, 1, nullptr, st));
            }
            {
                ProfScope ps(e, st, THMR_PROF_GEMM_FC2, 2.0 * M * DIM * (double)MLP, 6.0 * ((double)M * MLP + (double)DIM * MLP) + 4.0 * kSplitKMax * M * DIM);
                GemmArgs a = sgemm(bs, MLP, ws.fc2, nullptr, x, DIM);
                LAUNCH_OK(launch_gemm_split3_ring(a, EPI_NONE, kSplitKMax, part, st));
            }
            ProfScope ps(e, st, THMR_PROF_LN, 0, 4.0 * (kSplitKMax + 3.0) * M * DIM);
            if (last)
                LAUNCH_OK(launch_splitk_resid_ln(part, kSplitKMax, M, DIM, w.f2b, x, x, lastn_w, lastn_b, feats_out ? feats_out : h, VIT_EPS, st));
            else
                LAUNCH_OK(launch_splitk_resid_ln(part, kSplitKMax, M, DIM, w.f2b, x, x, e->vitw[i + 1].n1w, e->vitw[i + 1].n1b,
                                                 reinterpret_cast<float*>(hs), VIT_EPS, st, true));
        }
        return 0;
    }
#endif  // THMR_EXPERIMENTS
    {
        ProfScope ps(e, st, THMR_PROF_LN, 0, 8.0 * M * DIM);
        LAUNCH_OK(launch_layernorm(x, e->vitw[0].n1w, e->vitw[0].n1b, h, M, DIM, VIT_EPS, 0, st));
    }
    for (int i = 0; i < e->vit_depth; ++i) {
        const VitBlockW& w = e->vitw[i];
        const bool last = i + 1 == e->vit_depth;
        const float* nxt_w = last ? lastn_w : e->vitw[i + 1].n1w;      // the norm that follows this block's fc2
        const float* nxt_b = last ? lastn_b : e->vitw[i + 1].n1b;
        {   // qkv Linear; q columns scaled in the epilogue (vit.py:112,116)
            ProfScope ps(e, st, THMR_PROF_GEMM_QKV, 2.0 * M * DIM * 3.0 * DIM, 4.0 * ((double)M * DIM + 3.0 * DIM * DIM + 3.0 * M * DIM));
            GemmArgs a = mk(h, DIM, w.qkvw, DIM, w.qkvb, nullptr, 0, big, 3 * DIM, M, 3 * DIM, DIM);
            a.qscale = qscale; a.qcols = DIM;
            // one and two crops: 64 x 48 tiles of 16x16x4 MFMAs (240 / 480 workgroups whose waves walk 12.8 us chains) instead of 64x64
            // tiles of 32x32x2 (180 / 360 workgroups, 17.1 us chains); another order of the K sum, hence tied to that regime
            if (B <= kKeysplitMaxB && e->qkv_ring16) LAUNCH_OK(launch_gemm_ring16(a, EPI_BIAS_QSCALE, st));
            else if (ring_wide) LAUNCH_OK(launch_gemm_ring(a, EPI_BIAS_QSCALE, 4, 1, nullptr, st));
            else LAUNCH_OK(launch_gemm(a, EPI_BIAS_QSCALE, -1, st));
        }
        {
            ProfScope ps(e, st, THMR_PROF_ATTN, 4.0 * B * HEADS * 192.0 * 192.0 * 80.0, 4.0 * (4.0 * M * DIM));
            // one or two crops: keys split over the waves of a workgroup (192 / 96 workgroups per crop instead of 48, 120 / 240 instead
            // of 480 dependent MFMAs per wave): -3.9 % per call at one crop, -1 % at two, slower from three on, where its four-fold
            // re-reads of K / V cost more than the shorter chain saves (profiles/r3j_attention_keysplit_ab.log).  Its own association of
            // the key sum, hence its own regime {1, 2} inside the small-batch regime.
            if (B <= kKeysplitMaxB && e->attn_keysplit) LAUNCH_OK(launch_vit_attention_keysplit(big, h, B, st));
            else LAUNCH_OK(launch_vit_attention(big, h, B, st));
        }
        // proj + residual, then norm2 (vit.py:123,149,150)
        if (int rc = resid_linear_ln(THMR_PROF_GEMM_PROJ, h, DIM, w.pw, w.pb, ks_proj, mid_proj, w.n2w, w.n2b, h)) return rc;
        {   // fc1 + exact GELU (vit.py:83-84)
            ProfScope ps(e, st, THMR_PROF_GEMM_FC1, 2.0 * M * DIM * (double)MLP, 4.0 * ((double)M * DIM + (double)DIM * MLP + (double)M * MLP), (i & 3) == 0);
            GemmArgs a = mk(h, DIM, w.f1w, DIM, w.f1b, nullptr, 0, big, MLP, M, MLP, DIM);
            if (ring_wide) LAUNCH_OK(launch_gemm_ring(a, EPI_BIAS_GELU, 4, 1, nullptr, st));
            else LAUNCH_OK(launch_gemm(a, EPI_BIAS_GELU, -1, st));
        }
        // fc2 + residual (vit.py:85,150), then the next block's norm1 — or last_norm (vit.py:335), kept token-major: the
        // :337 permute is undone by token_head.py:69
        if (int rc = resid_linear_ln(THMR_PROF_GEMM_FC2, big, MLP, w.f2w, w.f2b, ks_fc2, mid_fc2, nxt_w, nxt_b, last && feats_out ? feats_out : h))
            return rc;
    }
    return 0;
}



// ---------------------------------------------------------------------------------------------- head
// DecodeTokens.forward (tokenization/models/vanilla_pose_vqvae.py:294-297): soft codebook lookup
// (quantize_cnn.py:92-93 dequantize_logits) + PoseSPDecoderV1 (:135-154).  probs (B,160,2048) -> bpose (B,21,6).
// 12 GEMMs and nothing else: every Conv1d(k = 3) is a GEMM over an im2col operand (B*T, 3*C) that the PREVIOUS GEMM's epilogue
// wrote (GemmArgs::cs_*: nearest-resample + taps + zero padding + the ResConv pre-activation ReLU), so no gather launch exists.
int vq_decode(thmr_engine* e, const float* probs, int B, float* bpose, hipStream_t st) {
    auto& so = e->so;
    // Up to six crops (the small-batch regime, B <= kSmallM / 192) these M = 21 B ... 160 B row products run on the tiny-M kernel
    // (32x32 tiles, K split over the 8 waves of a workgroup): as 8-24 ring-kernel workgroups walking the whole K they were 13
    // dependent launches of 12-32 us, a third of the head at one crop.  One choice for the whole regime: the K association differs.
    const int tv = (e->tiny_gemm && B * TOK <= kSmallM) ? 11 : -1;
    float* G[2] = {e->S(so.gat), e->S(so.gat2)};          // conv operands, alternating
    float *x0 = e->S(so.act0), *x1 = e->S(so.act1), *hid = e->S(so.act2);
    const int32_t* inv = reinterpret_cast<const int32_t*>(e->warena + e->o_inv);
    // what the consumer conv needs from its producer
    auto scatter = [&](GemmArgs& a, float* dst, const int32_t* table, int tin, int tout, int dil, int relu) {
        a.cs_out = dst; a.cs_inv = table; a.cs_tin = tin; a.cs_tout = tout; a.cs_dil = dil; a.cs_relu = relu;
    };
    // the conv itself: operand (B*T, 3*ci) x repacked weight (co, 3*ci)
    auto conv = [&](int id, const float* opnd, int T, const float* bias, float* plain) {
        const int ci = kConv3Ci[id], co = kConv3Co[id];
        return mk(opnd, 3 * ci, e->warena + e->convp[id], 3 * ci, bias, nullptr, 0, plain, co, B * T, co, 3 * ci);
    };
    {   // soft codebook lookup: probs @ codebook as a GEMM against codebook^T -> operand of decoder.0 (T = 160, C = 256)
        GemmArgs a = mk(probs, NCLS, e->warena + e->o_cbT, NCLS, nullptr, nullptr, 0, nullptr, CODE, B * TN, CODE, NCLS);
        scatter(a, G[0], nullptr, 160, 160, 1, 0);
        LAUNCH_OK(launch_gemm(a, EPI_NONE, tv, st));
    }
    int cur = 0;
    {   // decoder.0: Conv1d(256 -> 512) + ReLU at T = 160 -> operand of decoder.3 on the 160 -> 125 resample
        GemmArgs a = conv(0, G[cur], 160, e->hot.conv_b[0], nullptr);
        scatter(a, G[cur ^ 1], inv + 0 * 160, 160, e->vq_len[1], 1, 0);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS_RELU, tv, st));
        cur ^= 1;
    }
    for (int i = 0; i < 4; ++i) {   // decoder.3/6/9/12: nn.Upsample(size) (a down-sampling here) + Conv1d(512 -> 512) + ReLU
        const int T = e->vq_len[i + 1];
        GemmArgs a = conv(1 + i, G[cur], T, e->hot.conv_b[1 + i], i == 3 ? x0 : nullptr);
        if (i < 3) scatter(a, G[cur ^ 1], inv + (i + 1) * 160, T, e->vq_len[i + 2], 1, 0);
        else scatter(a, G[cur ^ 1], nullptr, VQJ, VQJ, 3, 1);      // -> ResConv1DBlock 0 (dilation 3, pre-activation ReLU); x0 kept as its residual
        LAUNCH_OK(launch_gemm(a, EPI_BIAS_RELU, tv, st));
        cur ^= 1;
    }
    const int Tq = VQJ;
    float* res = x0;
    float* nres = x1;
    for (int blk = 0; blk < 2; ++blk) {   // ResConv1DBlock, resnet.py:49-69: x + conv2(relu(conv1(relu(x)))), dilation 3 then 1
        {
            GemmArgs a = conv(5 + blk, G[cur], Tq, e->hot.conv_b[5 + blk], hid);
            LAUNCH_OK(launch_gemm(a, EPI_BIAS_RELU, tv, st));
        }
        GemmArgs a = mk(hid, VQW, e->hot.res_w[blk], VQW, e->hot.res_b[blk], res, VQW, blk == 0 ? nres : nullptr, VQW, B * Tq, VQW, VQW);
        // block 0 feeds block 1's conv1 (pre-activation ReLU) and stays its residual; block 1 feeds decoder.14.1 (no activation)
        scatter(a, G[cur ^ 1], nullptr, Tq, Tq, 1, blk == 0 ? 1 : 0);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS_RESID, tv, st));
        cur ^= 1;
        std::swap(res, nres);
    }
    {   // decoder.14.1: Conv1d(512 -> 512) -> operand of decoder.15
        GemmArgs a = conv(7, G[cur], Tq, e->hot.conv_b[7], nullptr);
        scatter(a, G[cur ^ 1], nullptr, Tq, Tq, 1, 0);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS, tv, st));
        cur ^= 1;
    }
    GemmArgs a = conv(8, G[cur], Tq, e->hot.conv_b[8], bpose);    // decoder.15: Conv1d(512 -> 6): the 21 x 6D body pose
    LAUNCH_OK(launch_gemm(a, EPI_BIAS, tv, st));
    return 0;
}

int head_forward(thmr_engine* e, const float* ctx, int B, const thmr_outputs* out, hipStream_t st) {
    auto& so = e->so;
    const int M = B * TOK;
    float* big = e->S(so.big);
    const int ldkv = e->dec_depth * 2 * INNER;
    {   // K8: to_kv of all decoder layers in ONE GEMM on the un-normalised context (pose_transformer.py:102,113)
        ProfScope ps(e, st, THMR_PROF_DEC_KV, 2.0 * M * DIM * (double)ldkv, 4.0 * ((double)M * DIM + (double)ldkv * DIM + (double)M * ldkv));
        const int s3_min = e->split3_min_b > 0 ? e->split3_min_b : kSplit3LowMinB;
        if (e->vit_gemm_mode == 1 && B >= s3_min && e->kv_s) {
            // split3 mode: the context (fp32: it may be a caller's buffer) converted into the ViT's idle operand buffer, then the same product
            // on the bf16 matrix pipe (1.5 -> 1.0 ms at 64 crops)
            LAUNCH_OK(launch_split3(ctx, DIM, e->split_act, DIM, M, DIM, st));
            GemmArgs a = mk(reinterpret_cast<const float*>(e->split_act), DIM, reinterpret_cast<const float*>(e->kv_s), DIM, nullptr, nullptr, 0, big, ldkv, M, ldkv, DIM);
            a.tile_opts = e->s3_tile_opts;
            if (e->s3_ws && e->s3_persist && (e->s3_persist_mask & 16) && gemm_split3_persist_ok(a)) LAUNCH_OK(launch_split3_persist_serialised(e, a, EPI_NONE, 0, st));
            else LAUNCH_OK(launch_gemm_split3(a, EPI_NONE, -1, st));
        } else {
            GemmArgs a = mk(ctx, DIM, e->warena + e->o_kv_all, DIM, nullptr, nullptr, 0, big, ldkv, M, ldkv, DIM);
            LAUNCH_OK(launch_gemm(a, EPI_NONE, -1, st));
        }
    }
    ProfScope ps_head(e, st, THMR_PROF_HEAD, 2.0 * B * (6.0 * 4.2e6 + 116.7e6 + 167.8e6 + 705.0e6), 0);
    float *dx = e->S(so.dx), *dh = e->S(so.dh), *dv = e->S(so.dv), *dq = e->S(so.dq), *dca = e->S(so.dca), *dff = e->S(so.dff);
    const std::string T = "smpl_head.transformer.";
    const std::string C = "smpl_head.decpose.";
    float* ro = e->S(so.ro);
    float *mt = e->S(so.mt), *cf = e->S(so.cf), *cf2 = e->S(so.cf2);
    const bool fused_head = !e->legacy_head && B <= kFusedHeadMaxB;
    bool mixer_in_decoder = false;
    if (fused_head) {
        // ONE persistent kernel: layer-0 input, the 6 decoder layers (42 dependent GEMV-class steps), the read-outs and the
        // classifier's first Linear (decoder_fused.hip)
        DecParams d = e->dec;
        d.B = B;
        // the MLP-Mixer stack runs inside the same kernel, 10 / 5 / 2 workgroups per crop while they fit one per CU (up to 25 / 51 /
        // 128 crops on 256 CUs; decoder_fused.hip mixer_cluster_stage, bit-identical to mixer_stack_kernel's one workgroup per crop)
        const int slots = d.max_blocks < 256 ? d.max_blocks : 256;
        d.mixer_cluster = !e->mixer_cluster ? 0 : 10 * B <= slots ? 10 : 5 * B <= slots ? 5 : 2 * B <= slots ? 2 : 0;
        mixer_in_decoder = d.mixer_cluster != 0;
        d.mx = e->mix;
        d.mixy[0] = cf; d.mixy[1] = cf2;
        LAUNCH_OK(launch_decoder_serialised(e, d, st));
    } else {
    LAUNCH_OK(launch_decoder_init(e->W(T + "to_token_embedding.bias"), e->W(T + "pos_embedding"), dx, B, E, st));
    for (int l = 0; l < e->dec_depth; ++l) {
        const std::string p = T + "transformer.layers." + std::to_string(l) + ".";
        // self-attention over ONE token: softmax of a single score == 1, so out = to_out(v)  (pose_transformer.py:75-86)
        LAUNCH_OK(launch_layernorm(dx, e->W(p + "0.norm.weight"), e->W(p + "0.norm.bias"), dh, B, E, LN_EPS, 0, st));
        {
            GemmArgs a = mk(dh, E, e->W(p + "0.fn.to_qkv.weight") + (size_t)2 * INNER * E, E, nullptr, nullptr, 0, dv, INNER, B, INNER, E);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_NONE, st));
        }
        {
            GemmArgs a = mk(dv, INNER, e->W(p + "0.fn.to_out.0.weight"), INNER, e->W(p + "0.fn.to_out.0.bias"), dx, E, dx, E, B, E, INNER);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS_RESID, st));
        }
        // cross-attention (pose_transformer.py:111-124)
        LAUNCH_OK(launch_layernorm(dx, e->W(p + "1.norm.weight"), e->W(p + "1.norm.bias"), dh, B, E, LN_EPS, 0, st));
        {
            GemmArgs a = mk(dh, E, e->W(p + "1.fn.to_q.weight"), E, nullptr, nullptr, 0, dq, INNER, B, INNER, E);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_NONE, st));
        }
        LAUNCH_OK(launch_cross_attn(dq, big, ldkv, l * 2 * INNER, dca, B, st));
        {
            GemmArgs a = mk(dca, INNER, e->W(p + "1.fn.to_out.0.weight"), INNER, e->W(p + "1.fn.to_out.0.bias"), dx, E, dx, E, B, E, INNER);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS_RESID, st));
        }
        // feed-forward (pose_transformer.py:40-52)
        LAUNCH_OK(launch_layernorm(dx, e->W(p + "2.norm.weight"), e->W(p + "2.norm.bias"), dh, B, E, LN_EPS, 0, st));
        {
            GemmArgs a = mk(dh, E, e->W(p + "2.fn.net.0.weight"), E, e->W(p + "2.fn.net.0.bias"), nullptr, 0, dff, DEC_MLP, B, DEC_MLP, E);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS_GELU, st));
        }
        {
            GemmArgs a = mk(dff, DEC_MLP, e->W(p + "2.fn.net.3.weight"), DEC_MLP, e->W(p + "2.fn.net.3.bias"), dx, E, dx, E, B, E, DEC_MLP);
            LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS_RESID, st));
        }
    }
    // read-outs: one (31,1024) GEMV-class GEMM (token_head.py:99-105)
    {
        GemmArgs a = mk(dx, E, e->warena + e->o_ro_w, E, e->warena + e->o_ro_b, nullptr, 0, ro, 32, B, 31, E);
        LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS, st));
    }
    // token classifier (token_classifier.py:89-104)
    {
        GemmArgs a = mk(dx, E, e->W(C + "mixer_trans.ff.0.weight"), E, e->W(C + "mixer_trans.ff.0.bias"), nullptr, 0, mt, TN * HID, B, TN * HID, E);
        LAUNCH_OK(launch_gemm_skinny(a, EPI_BIAS, st));
    }
    }
    if (out && out->token_out) HIP_OK(hipMemcpyAsync(out->token_out, dx, sizeof(float) * B * E, hipMemcpyDeviceToDevice, st));
    const int R = B * TN;
    float *nl = e->S(so.nl), *nl2 = e->S(so.nl2);
    if (fused_head) {
        // ONE kernel, one workgroup per crop: mixer_trans LayerNorm + ReLU, the 4 MixerLayers, mixer_norm_layer (mixer_fused.hip)
        if (!mixer_in_decoder) LAUNCH_OK(launch_mixer_fused(e->mix, B, st));
    } else {
        LAUNCH_OK(launch_layernorm(mt, e->W(C + "mixer_trans.ff.1.weight"), e->W(C + "mixer_trans.ff.1.bias"), cf, B, TN * HID, LN_EPS, 1, st));
        for (int m = 0; m < MIX; ++m) {   // MixerLayer, heads/modules.py:55-63
            const std::string p = C + "mixer_head." + std::to_string(m) + ".";
            float *y1 = e->S(so.y1), *tT = e->S(so.tT), *u = e->S(so.u), *yt = e->S(so.yt), *y = e->S(so.y), *s = e->S(so.s),
                  *z0 = e->S(so.z0), *zh = e->S(so.zh);
            LAUNCH_OK(launch_layernorm(cf, e->W(p + "layernorm1.weight"), e->W(p + "layernorm1.bias"), y1, R, HID, LN_EPS, 0, st));
            LAUNCH_OK(launch_transpose(y1, tT, B, TN, HID, st));                                   // (B,160,64)->(B,64,160)
            {
                GemmArgs a = mk(tT, TN, e->W(p + "MLP_token.ff.0.weight"), TN, e->W(p + "MLP_token.ff.0.bias"), nullptr, 0, u, TOK_INTER, B * HID, TOK_INTER, TN);
                LAUNCH_OK(launch_gemm(a, EPI_BIAS_GELU, -1, st));
            }
            {
                GemmArgs a = mk(u, TOK_INTER, e->W(p + "MLP_token.ff.3.weight"), TOK_INTER, e->W(p + "MLP_token.ff.3.bias"), nullptr, 0, yt, TN, B * HID, TN, TOK_INTER);
                LAUNCH_OK(launch_gemm(a, EPI_BIAS, -1, st));
            }
            LAUNCH_OK(launch_transpose(yt, y, B, HID, TN, st));                                    // (B,64,160)->(B,160,64)
            LAUNCH_OK(launch_add_ln64(cf, y, e->W(p + "layernorm2.weight"), e->W(p + "layernorm2.bias"), s, z0, R, LN_EPS, st));
            {
                GemmArgs a = mk(z0, HID, e->W(p + "MLP_channel.ff.0.weight"), HID, e->W(p + "MLP_channel.ff.0.bias"), nullptr, 0, zh, HID_INTER, R, HID_INTER, HID);
                LAUNCH_OK(launch_gemm(a, EPI_BIAS_GELU, -1, st));
            }
            {   // out = (x + y) + z
                GemmArgs a = mk(zh, HID_INTER, e->W(p + "MLP_channel.ff.3.weight"), HID_INTER, e->W(p + "MLP_channel.ff.3.bias"), s, HID, cf2, HID, R, HID, HID_INTER);
                LAUNCH_OK(launch_gemm(a, EPI_BIAS_RESID, -1, st));
            }
            std::swap(cf, cf2);
        }
        {
            GemmArgs a = mk(cf, HID, e->W(C + "mixer_norm_layer.ff.0.weight"), HID, e->W(C + "mixer_norm_layer.ff.0.bias"), nullptr, 0, nl, HID, R, HID, HID);
            LAUNCH_OK(launch_gemm(a, EPI_BIAS, -1, st));
        }
        LAUNCH_OK(launch_layernorm(nl, e->W(C + "mixer_norm_layer.ff.1.weight"), e->W(C + "mixer_norm_layer.ff.1.bias"), nl2, R, HID, LN_EPS, 1, st));
    }
    // logits / softmax / token index; KV in `big` is dead after the decoder, so logits+probs live there
    float* logits = (out && out->cls_logits) ? out->cls_logits : big;
    float* probs = (out && out->cls_logits_softmax) ? out->cls_logits_softmax : big + (size_t)R * NCLS;
    int32_t* tokidx = (out && out->token_idx) ? out->token_idx : reinterpret_cast<int32_t*>(e->S(so.tokidx));
    {
        GemmArgs a = mk(nl2, HID, e->hot.cls_w, HID, e->hot.cls_b, nullptr, 0, logits, NCLS, R, NCLS, HID);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS, -1, st));
    }
    LAUNCH_OK(launch_softmax_argmax2048(logits, probs, tokidx, R, st));
    float* bpose = e->S(so.bpose);
    if (int r = vq_decode(e, probs, B, bpose, st)) return r;
    // assemble + rot6d + camera (token_head.py:103-127, tokenhmr.py:165-169)
    float* rot = (out && out->rotmat) ? out->rotmat : e->S(so.rot);
    float* betas = (out && out->betas) ? out->betas : e->S(so.betas);
    float* cam = (out && out->pred_cam) ? out->pred_cam : e->S(so.cam);
    float* camt = (out && out->pred_cam_t) ? out->pred_cam_t : e->S(so.camt);
    LAUNCH_OK(launch_assemble(ro, 32, bpose, e->hot.init_pose, e->hot.init_betas, e->hot.init_cam,
                              out ? out->pose6d : nullptr, rot, betas, cam, camt,
                              out ? out->focal_length : nullptr, FOCAL, IMG, B, st));
    return 0;
}

int lbs(thmr_engine* e, const float* rot, const float* betas, const float* camt, int B, float* verts, float* joints,
        float* kp2d, hipStream_t st) {
    auto& so = e->so;
    ProfScope ps(e, st, THMR_PROF_LBS, 2.0 * B * 8.1e6, 4.0 * (B * (NV * 3.0 + 132 + 226) + 4.95e6));
    const int32_t* ints = reinterpret_cast<const int32_t*>(e->warena + e->o_smpl_int);
    if (!verts) verts = e->S(so.verts);
    LAUNCH_OK(launch_lbs(rot, betas, camt, e->warena + e->o_smpl_jt, e->warena + e->o_smpl_jsd, ints,
                         e->warena + e->o_smpl_vt, e->warena + e->o_smpl_dirs, e->warena + e->o_smpl_w,
                         e->warena + e->o_smpl_j19, ints + 24, ints + 48, ints + 80, e->S(so.A), e->S(so.pf), e->S(so.Jtr),
                         e->S(so.vposed), verts, joints, kp2d, FOCAL / IMG, B, e->S(so.xv),
                         reinterpret_cast<unsigned*>(e->S(so.lcnt)), st));
    return 0;
}

// nn.Upsample nearest index tables of the VQ decoder / tokenizer encoder, evaluated in fp32 exactly like ATen
// (nearest_neighbor_compute_source_index: min(floor(dst * (float)in / out), in - 1)).  They are part of the weight arena
// (so a broadcast replicates them) and are written by the engine that LOADED the weights — never by thmr_create, which
// may be handed another engine's live arena (Engine(weight_arena=...)).
void build_idx_tables(thmr_engine* e) {
    e->idx_host.assign(4 * 160, 0);
    for (int i = 0; i < 4; ++i) {
        const int tin = e->vq_len[i], tout = e->vq_len[i + 1];
        const float scale = (float)tin / (float)tout;
        for (int t = 0; t < tout; ++t) {
            int sidx = (int)floorf((float)t * scale);
            e->idx_host[i * 160 + t] = sidx < tin - 1 ? sidx : tin - 1;
        }
    }
    // inverse tables of the decoder's four down-samplings: position ts of the producer -> the resampled position tp that reads it
    // (src is strictly increasing, so at most one), -1 = dropped.  Used by the GEMM epilogue that writes the next conv's operand.
    e->inv_host.assign(4 * 160, -1);
    for (int i = 0; i < 4; ++i)
        for (int t = 0; t < e->vq_len[i + 1]; ++t) e->inv_host[i * 160 + e->idx_host[i * 160 + t]] = t;
    // encoder: nn.Upsample(size=40) from 21 (offset 0), then three nn.Upsample(scale_factor=2)
    // (ATen uses scale 1/scale_factor = 0.5: src = floor(dst*0.5)): 40->80 (offset 40), 80->160 (120), 160->320 (280)
    e->eidx_host.assign(640, 0);
    const float sc = 21.0f / 40.0f;
    for (int t = 0; t < 40; ++t) { int s0 = (int)floorf((float)t * sc); e->eidx_host[t] = s0 < 20 ? s0 : 20; }
    int o = 40;
    for (int tin = 40; tin <= 160; tin *= 2) {
        for (int t = 0; t < 2 * tin; ++t) { int s0 = (int)floorf((float)t * 0.5f); e->eidx_host[o + t] = s0 < tin - 1 ? s0 : tin - 1; }
        o += 2 * tin;
    }
}
constexpr int32_t kEncMagic = 0x454e4331;   // 'ENC1': arena flag word 0 = "tokenizer encoder tensors present"
constexpr int32_t kArenaMagic = 0x54484d32; // 'THM2': arena flag word 1 = "the non-checkpoint regions below were written by a loading engine"

// Everything in the weight arena that is neither a checkpoint tensor nor derived from one: the resample index tables, the zero
// padding row of the read-out matrix and the flag words.  Written by the engine that LOADS tensors, at load time — so the arena
// is complete for a broadcast the moment loading ends, whether or not the root has finalized yet (round 2 wrote them in
// thmr_finalize_weights(assume_all_loaded = 0) only: a root that broadcast BEFORE finalizing shipped uninitialised tables, and
// receivers, which finalize with assume_all_loaded = 1, never wrote them).  Receivers validate kArenaMagic instead of writing:
// their arena may be another engine's live one (Engine(weight_arena=...)).
int write_arena_constants(thmr_engine* e, hipStream_t st) {
    build_idx_tables(e);
    HIP_OK(hipMemcpyAsync(e->warena + e->o_idx, e->idx_host.data(), e->idx_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_inv, e->inv_host.data(), e->inv_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_idx_enc, e->eidx_host.data(), e->eidx_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(e->warena + e->o_ro_w + (size_t)31 * E, 0, E * sizeof(float), st));
    // optional encoder: present only when every 'encoder.encoder.*' tensor arrived (all-or-nothing, checked by finalize)
    size_t enc_loaded = 0;
    for (auto& n : e->enc_names) enc_loaded += e->slots[n].loaded ? 1 : 0;
    e->flag_host[0] = enc_loaded == e->enc_names.size() ? kEncMagic : 0;
    e->flag_host[1] = kArenaMagic;
    HIP_OK(hipMemcpyAsync(e->warena + e->o_flags, e->flag_host, 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    return 0;
}

// The persistent kernels need ALL their workgroups resident at once: the decoder kernel has a grid barrier and, from 49 crops on, asks
// for every CU; the persistent split3 GEMM's consumers wait for slabs their producers publish, one 147 KB workgroup per CU.  Two such
// launches issued concurrently by two engines on two streams could each grab part of the chip and wait for the rest until the bounded
// waits run out.  So when a process has more than one engine on a device, their forward-type calls are chained through one event per
// device (struct Turn below: one wait + one record per CALL since round 6, not per launch): each call waits for the previous one (of any
// engine) to finish.  One engine (the normal case) never touches the event, and a capturing stream does not either (an event from outside
// a capture cannot be waited on).  ANOTHER PROCESS on the same GPU is outside this protection: see THMR_CFG_NO_PERSISTENT (header).  (Engines of the shipped and of the experiments
// library in one process each have their own copy of this object: do not run them concurrently on one device — tests do not.)
struct DecoderTurnstile {
    std::mutex mu;
    std::map<int, int> engines;          // device -> live engines
    std::map<int, hipEvent_t> last;      // device -> event recorded at the end of the most recent forward-type call of any engine
};
DecoderTurnstile& turnstile() {
    static DecoderTurnstile t;
    return t;
}

// One TURN per forward-type call (round 6, ADVICE r5: round 5 took the mutex and waited / recorded the event around EVERY persistent launch,
// 33 and more per forward).  With more than one engine of this library on the device and no capture in progress, the call's stream first
// waits for the event the previous call (of any engine) recorded at its end, and records it again at its own end; the mutex is held for the
// host-side enqueue of the whole call (~0.3 ms), so two threads cannot both wait on the same earlier record and then overlap.  Whole
// forwards of different engines therefore run one after the other on the device — which is what kernels that need all 256 CUs resident
// amount to anyway.  One engine per device (the normal case): no lock held, no event touched.
struct Turn {
    std::unique_lock<std::mutex> lk;
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    int rc = 0;
    Turn(thmr_engine* e, hipStream_t s) : lk(turnstile().mu), st(s) {
        DecoderTurnstile& t = turnstile();
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (t.engines[e->cfg.device] <= 1 || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            lk.unlock();
            return;
        }
        auto it = t.last.find(e->cfg.device);
        if (it == t.last.end()) {
            hipEvent_t nev;
            if (hipEventCreateWithFlags(&nev, hipEventDisableTiming) != hipSuccess) { rc = -2; lk.unlock(); return; }
            it = t.last.emplace(e->cfg.device, nev).first;
        } else if (hipStreamWaitEvent(st, it->second, 0) != hipSuccess) { rc = -2; lk.unlock(); return; }
        ev = it->second;
    }
    // records the end of the turn; returns -2 if the record failed (the launches themselves were fine)
    int end() {
        int r = 0;
        if (ev) {
            if (hipEventRecord(ev, st) != hipSuccess) r = -2;
            ev = nullptr;
        }
        if (lk.owns_lock()) lk.unlock();
        return r;
    }
    ~Turn() { (void)end(); }
};

int launch_decoder_serialised(thmr_engine* e, const DecParams& d, hipStream_t st) {
    (void)e;
    return launch_decoder_fused(d, st);                // the caller's Turn (thmr_forward / thmr_head_forward) orders engines
}

int launch_split3_persist_serialised(thmr_engine* e, const GemmArgs& a, int epi, int mode, hipStream_t st) {
    return launch_gemm_split3_persist(a, epi, mode, e->s3_ws, st);
}

// split3 weight copies are a pure function of the weight arena's contents: engines that SHARE an arena (thmr_create with the same
// weight_arena_dev, e.g. several engines of one process on one GPU) share ONE copy, reference-counted (round 6, ADVICE r5: eight engines on
// one device held 30 GB of duplicates).  Every finalize still runs the conversion into it — idempotent, the same bytes.
struct SplitShare {
    std::mutex mu;
    struct Ent { char* p; size_t bytes; int refs; };
    std::map<std::pair<int, const void*>, Ent> m;
};
SplitShare& split_share() {
    static SplitShare s;
    return s;
}

// The persistent decoder kernel's bounded grid barrier timed out in an earlier call (its workgroups were not resident together:
// another process on the GPU, a partition with fewer CUs than reported ...).  That call's outputs are garbage and its barrier
// words are in an undefined state.  Recover instead of staying poisoned: drain the device, zero the barrier words and the
// sticky error, and run this engine's head as the launch chain from now on (it needs no co-residency), so the caller can
// simply re-submit.  Returns the error ONCE.
int recover_decoder_timeout(thmr_engine* e, hipStream_t st = nullptr) {
    // hipDeviceSynchronize / hipMemset are illegal while a stream of this thread is being captured into a hipGraph (they would also
    // invalidate the capture): report, leave the device alone, and let the caller recover outside the capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(e, THMR_ERR_HIP, "persistent decoder kernel: grid barrier timed out in a previous forward, and this call is inside a "
                                     "stream capture: end the capture, call thmr_engine_status() (it resets the engine), then re-capture");
    (void)hipDeviceSynchronize();
    (void)hipMemset(e->sarena + e->so.sync, 0, 512 * sizeof(float));
    if (e->host_err) *e->host_err = 0;
    e->legacy_head = true;
    return fail(e, THMR_ERR_HIP, "persistent decoder kernel: grid barrier timed out in a previous forward (its workgroups could not be "
                                 "resident together); that call's outputs are invalid.  The engine has reset its barrier and switched to "
                                 "the launch-chain head: re-submit the batch");
}

// zero the hand-over workspace of the persistent split3 GEMM (slabs, epochs, control words) and bind the host-mapped error word again
int reset_s3_workspace(thmr_engine* e, hipStream_t st) {
    HIP_OK(hipMemsetAsync(e->s3_ws, 0, gemm_split3_persist_ws_bytes(), st));
    if (e->s3_host_err && gemm_split3_persist_bind_host_err(e->s3_ws, &e->s3_host_err, st) != 0)
        return fail(e, THMR_ERR_HIP, "binding the host-mapped error word of the split3 hand-over workspace failed");
    return 0;
}

// A consumer of the persistent split3 GEMM gave up waiting for its producer's slab in an earlier call (the launch's workgroups were not
// resident together for ~0.5 s: another kernel held the device).  That call's outputs are invalid.  The epochs in the flags make later
// launches safe by themselves (a late producer cannot be mistaken for a later launch's), but the cause is likely to persist: report ONCE,
// reset the workspace and fall back to the one-workgroup-per-tile kernel (same results), so the caller can simply re-submit.
int recover_split3_timeout(thmr_engine* e, hipStream_t st = nullptr) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(e, THMR_ERR_HIP, "persistent split3 GEMM: a hand-over wait timed out in a previous forward, and this call is inside a "
                                     "stream capture: end the capture, call thmr_engine_status() (it resets the engine), then re-capture");
    (void)hipDeviceSynchronize();
    if (e->s3_host_err) *e->s3_host_err = 0;
    e->s3_persist = 0;
    if (e->s3_ws) {
        if (int r = reset_s3_workspace(e, nullptr)) return r;
        (void)hipDeviceSynchronize();
    }
    return fail(e, THMR_ERR_HIP, "persistent split3 GEMM: a hand-over wait timed out in a previous forward; that call's outputs are invalid. "
                                 "The engine has reset the workspace and switched to the per-tile kernel: re-submit the batch");
}

// the persistent split3 GEMM's decomposition is 8 XCDs x 32 CUs: only offered on a 256-CU device (cached per device)
bool device_has_256_cus() {
    static std::mutex mu;
    static std::map<int, bool> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    hipDeviceProp_t prop;
    const bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount == 256;
    cache[dev] = ok;
    return ok;
}

int check_ready(thmr_engine* e, int B, hipStream_t st = nullptr) {
    if (!e) return fail(nullptr, THMR_ERR_INVALID, "null engine");
    if (e->host_err && *static_cast<volatile unsigned*>(e->host_err) != 0) return recover_decoder_timeout(e, st);
    if (e->s3_host_err && *static_cast<volatile unsigned*>(e->s3_host_err) != 0) return recover_split3_timeout(e, st);
    if (!e->finalized) return fail(e, THMR_ERR_STATE, "weights not finalized: call thmr_finalize_weights first");
    if (B < 1 || B > e->max_batch) return fail(e, THMR_ERR_INVALID, "batch " + std::to_string(B) + " outside [1, max_batch=" + std::to_string(e->max_batch) + "]");
    return 0;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

int thmr_abi_version(void) { return THMR_ABI_VERSION; }
#ifndef THMR_SRC_HASH
#define THMR_SRC_HASH "unhashed"      // __graft_entry__.build() passes the content hash of csrc/ + include/ + flags
#endif
#ifdef THMR_EXPERIMENTS
#define THMR_BUILD_KIND " experiments"      // environment knobs, debug hooks and the kernels that lost their A/B (csrc/common.h)
#else
#define THMR_BUILD_KIND ""
#endif
const char* thmr_build_info(void) { return "tokenhmr_hip gfx950 fp32-mfma src:" THMR_SRC_HASH THMR_BUILD_KIND " " __DATE__ " " __TIME__; }

const char* thmr_last_error(const thmr_engine* e) { return e ? e->err.c_str() : g_last_error.c_str(); }

static int validate_cfg(const thmr_config* cfg) {
    if (!cfg) return fail(nullptr, THMR_ERR_INVALID, "null config");
    if (cfg->abi_version != THMR_ABI_VERSION) return fail(nullptr, THMR_ERR_INVALID, "abi_version mismatch");
    if (cfg->vit_depth < 1 || cfg->vit_depth > 64 || cfg->dec_depth < 1 || cfg->dec_depth > 6 || cfg->max_batch < 1 ||
        cfg->max_batch > 4096)
        return fail(nullptr, THMR_ERR_INVALID, "config out of range (vit_depth 1..64, dec_depth 1..6, max_batch 1..4096)");
    if ((cfg->flags & ~(THMR_CFG_VIT_GEMM_F32 | THMR_CFG_NO_PERSISTENT)) != 0 || cfg->reserved[0] != 0 || cfg->reserved[1] != 0)
        return fail(nullptr, THMR_ERR_INVALID, "unknown config flag / non-zero reserved field");
    return 0;
}

// bytes of the engine-owned allocations of the split3 mode (build_split_weights): weight copies, activation operands + fc2's split-K planes
static size_t split_w_bytes_of(int vit_depth, int dec_depth) {
    const size_t per_block = (size_t)DIM * (3 * DIM) + (size_t)DIM * DIM + 2 * (size_t)DIM * MLP;      // weights of one block
    const size_t kv_rows = (size_t)dec_depth * 2 * INNER;                                              // + the decoder's to_kv of all layers
    return (per_block * vit_depth + kv_rows * DIM + (size_t)DIM * 768) * 6;                            // + the patch-embed matrix
}
static size_t split_act_bytes_of(int max_batch) {
    const size_t M = (size_t)max_batch * TOK;
    return M * (size_t)(DIM + MLP) * 6 + (size_t)kSplit3Fc2Split * M * DIM * 4;
}

int thmr_mode_bytes(const thmr_config* cfg, int32_t vit_gemm_mode, size_t* split_weight_bytes, size_t* split_act_bytes, size_t* workspace_bytes) {
    if (int r = validate_cfg(cfg)) return r;
    if (vit_gemm_mode != 0 && vit_gemm_mode != 1) return fail(nullptr, THMR_ERR_INVALID, "vit gemm mode must be 0 or 1");
    const bool on = vit_gemm_mode == 1 && cfg->max_batch >= kSplit3LowMinB;
    if (split_weight_bytes) *split_weight_bytes = on ? split_w_bytes_of(cfg->vit_depth, cfg->dec_depth) : 0;
    if (split_act_bytes) *split_act_bytes = on ? split_act_bytes_of(cfg->max_batch) : 0;
    if (workspace_bytes) *workspace_bytes = on && !(cfg->flags & THMR_CFG_NO_PERSISTENT) ? gemm_split3_persist_ws_bytes() : 0;
    return 0;
}

int thmr_arena_bytes(const thmr_config* cfg, size_t* weight_bytes, size_t* scratch_bytes) {
    if (int r = validate_cfg(cfg)) return r;
    thmr_engine tmp;
    tmp.vit_depth = cfg->vit_depth; tmp.dec_depth = cfg->dec_depth; tmp.max_batch = cfg->max_batch;
    layout_weights(&tmp);
    layout_scratch(&tmp);
    if (weight_bytes) *weight_bytes = tmp.wfloats * sizeof(float);
    if (scratch_bytes) *scratch_bytes = tmp.sfloats * sizeof(float);
    return 0;
}

int thmr_spec(const thmr_config* cfg, int32_t index, const char** name, int64_t* numel) {
    if (int r = validate_cfg(cfg)) return r;
    static thread_local std::vector<std::pair<std::string, int64_t>> spec;
    spec.clear();
    build_spec(cfg->vit_depth, cfg->dec_depth, spec);
    if (index >= 0 && index < (int32_t)spec.size()) {
        if (name) *name = spec[index].first.c_str();
        if (numel) *numel = spec[index].second;
    }
    return (int)spec.size();
}

int thmr_create(const thmr_config* cfg, void* weight_arena_dev, void* scratch_arena_dev, thmr_engine** out) {
    if (!out) return fail(nullptr, THMR_ERR_INVALID, "null out");
    *out = nullptr;
    if (int r = validate_cfg(cfg)) return r;
    thmr_engine* e = new thmr_engine();
    e->cfg = *cfg;
    e->vit_depth = cfg->vit_depth; e->dec_depth = cfg->dec_depth; e->max_batch = cfg->max_batch;
    layout_weights(e);
    layout_scratch(e);
    auto bail = [&](int code, const std::string& m) { fail(nullptr, code, m); thmr_destroy(e); return code; };
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(THMR_ERR_HIP, "hipSetDevice failed");
    if (weight_arena_dev) e->warena = static_cast<float*>(weight_arena_dev);
    else {
        if (hipMalloc(&e->warena, e->wfloats * sizeof(float)) != hipSuccess) return bail(THMR_ERR_NOMEM, "hipMalloc(weights) failed");
        e->own_w = true;
    }
    if (scratch_arena_dev) e->sarena = static_cast<float*>(scratch_arena_dev);
    else {
        if (hipMalloc(&e->sarena, e->sfloats * sizeof(float)) != hipSuccess) return bail(THMR_ERR_NOMEM, "hipMalloc(scratch) failed");
        e->own_s = true;
    }
    // grid-barrier words of the persistent decoder kernel (the scratch arena is private to this engine; a caller-provided
    // one may hold garbage)
    if (hipMemset(e->sarena + e->so.sync, 0, 512 * sizeof(float)) != hipSuccess) return bail(THMR_ERR_HIP, "hipMemset(sync words) failed");
    // arrival counters of the fused skin + joints kernel (self-resetting; zero before the first call)
    if (hipMemset(e->sarena + e->so.lcnt, 0, (size_t)e->max_batch * sizeof(float)) != hipSuccess) return bail(THMR_ERR_HIP, "hipMemset(lbs counters) failed");
    // sticky error word of the persistent decoder kernel, host-mapped so that the next call sees a timeout without a D2H copy
    if (hipHostMalloc(reinterpret_cast<void**>(&e->host_err), 64, hipHostMallocMapped) != hipSuccess) return bail(THMR_ERR_NOMEM, "hipHostMalloc(error word) failed");
    e->host_err[0] = e->host_err[1] = 0;
    e->s3_host_err = e->host_err + 1;
    { const char* lg = thmr_knob("THMR_LEGACY_HEAD"); e->legacy_head = lg && lg[0] == '1'; }
    { const char* mc = thmr_knob("THMR_MIXER_CLUSTER"); e->mixer_cluster = !(mc && mc[0] == '0'); }
    { const char* tg = thmr_knob("THMR_TINY_GEMM"); e->tiny_gemm = !(tg && tg[0] == '0'); }
    { const char* qr = thmr_knob("THMR_QKV_RING16"); e->qkv_ring16 = !(qr && qr[0] == '0'); }
    { const char* ak = thmr_knob("THMR_ATTN_KEYSPLIT"); e->attn_keysplit = !(ak && ak[0] == '0'); }
    { const char* ab = thmr_knob("THMR_ATTN_B16"); if (ab && (ab[0] == '0' || ab[0] == '1')) e->attn_b16 = ab[0] == '1'; }
    { const char* ss = thmr_knob("THMR_SPLIT3_SMALL"); e->split3_small = ss && ss[0] == '1'; }
    { const char* fs = thmr_knob("THMR_SPLIT3_FC2_SPLIT"); e->split3_fc2_split = (fs && fs[0] == '1') ? 1 : kSplit3Fc2Split; }
    { const char* sm = thmr_knob("THMR_SPLIT3_MIN_B"); e->split3_min_b = sm ? atoi(sm) : 0; }
    { const char* n8 = thmr_knob("THMR_SPLIT3_NARROW8"); if (n8 && n8[0] == '1') e->s3_tile_opts |= 1; }
    { const char* t8 = thmr_knob("THMR_SPLIT3_TAIL8"); if (t8 && t8[0] == '1') e->s3_tile_opts |= 2; }
    { const char* r3 = thmr_knob("THMR_SPLIT3_RING3"); if (r3 && r3[0] == '0') e->s3_tile_opts |= 4; }
    { const char* fr = thmr_knob("THMR_SPLIT3_FRONT"); if (fr && fr[0] == '0') e->s3_tile_opts |= 8; }
    { const char* pm = thmr_knob("THMR_SPLIT3_PN_MASK"); if (pm) e->s3_pn_mask = atoi(pm); }
    { const char* px = thmr_knob("THMR_SPLIT3_PN_MAX"); if (px) e->s3_pn_max = atoi(px); }
    { const char* pf = thmr_knob("THMR_SPLIT3_PN_FILL"); if (pf) e->s3_pn_fill = atoi(pf); }
    { const char* pk = thmr_knob("THMR_SPLIT3_PK_MAX"); if (pk) e->s3_pk_max = atoi(pk); }
    { const char* pp = thmr_knob("THMR_SPLIT3_PN_FILL_PROJ"); if (pp) e->s3_pn_fill_proj = atoi(pp); }
    { const char* pw = thmr_knob("THMR_SPLIT3_PW_FILL"); if (pw) e->s3_pw_fill = atoi(pw); }
    { const char* p1 = thmr_knob("THMR_SPLIT3_PW_FC1"); if (p1 && p1[0] == '0') e->s3_pw_fc1 = 0; }
    { const char* sp = thmr_knob("THMR_SPLIT3_PERSIST"); if (sp && sp[0] >= '0' && sp[0] <= '1') e->s3_persist = sp[0] - '0'; }
    { const char* fm = thmr_knob("THMR_SPLIT3_FC1_MODE"); if (fm && fm[0] >= '0' && fm[0] <= '2') e->s3_fc1_mode = fm[0] - '0'; }
    { const char* mk_ = thmr_knob("THMR_SPLIT3_PERSIST_MASK"); if (mk_) e->s3_persist_mask = atoi(mk_); }
    { const char* ms = thmr_knob("THMR_MID_SPLIT"); if (ms && ms[0] && ms[1]) { e->mid_split_force[0] = ms[0] - '0'; e->mid_split_force[1] = ms[1] - '0'; } }
    if (cfg->flags & THMR_CFG_VIT_GEMM_F32) e->vit_gemm_mode = 0;      // created in the opt-out mode: finalize builds nothing for split3
    if (cfg->flags & THMR_CFG_NO_PERSISTENT) {                         // no kernel that needs all its workgroups resident at once
        e->s3_persist = 0;
        e->legacy_head = true;
        e->no_persistent = true;
    }
    { DecoderTurnstile& t = turnstile(); std::lock_guard<std::mutex> lk(t.mu); t.engines[cfg->device] += 1; e->counted = true; }
    *out = e;
    return 0;
}

void thmr_destroy(thmr_engine* e) {
    if (!e) return;
    if (e->counted) {
        DecoderTurnstile& t = turnstile();
        std::lock_guard<std::mutex> lk(t.mu);
        if (--t.engines[e->cfg.device] <= 0) {              // the device's last engine: its turnstile event goes with it
            auto it = t.last.find(e->cfg.device);
            if (it != t.last.end()) { (void)hipEventDestroy(it->second); t.last.erase(it); }
        }
    }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->host_err) (void)hipHostFree(e->host_err);
    if (e->split_w_counted) {
        SplitShare& sh = split_share();
        std::lock_guard<std::mutex> lk(sh.mu);
        auto it = sh.m.find({e->cfg.device, static_cast<const void*>(e->warena)});
        if (it != sh.m.end() && --it->second.refs <= 0) { (void)hipFree(it->second.p); sh.m.erase(it); }
    } else if (e->split_w) (void)hipFree(e->split_w);
    if (e->split_act) (void)hipFree(e->split_act);
    if (e->s3_ws) (void)hipFree(e->s3_ws);
    if (e->own_w && e->warena) (void)hipFree(e->warena);
    if (e->own_s && e->sarena) (void)hipFree(e->sarena);
    delete e;
}

int thmr_load_weights(thmr_engine* e, const thmr_tensor_desc* t, size_t n, void* stream) {
    if (!e || (!t && n)) return fail(e, THMR_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t i = 0; i < n; ++i) {
        if (!t[i].name || !t[i].data) return fail(e, THMR_ERR_INVALID, "tensor desc with null name/data");
        auto it = e->slots.find(t[i].name);
        if (it == e->slots.end()) return fail(e, THMR_ERR_INVALID, std::string("unexpected tensor '") + t[i].name + "' (strict load)");
        if (it->second.numel != t[i].numel)
            return fail(e, THMR_ERR_INVALID, std::string("tensor '") + t[i].name + "': expected " + std::to_string(it->second.numel) +
                                                 " elements, got " + std::to_string(t[i].numel));
        HIP_OK(hipMemcpyAsync(e->warena + it->second.off, t[i].data, sizeof(float) * t[i].numel,
                              t[i].on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        it->second.loaded = true;
    }
    e->finalized = false;
    return write_arena_constants(e, st);
}

int thmr_load_smpl(thmr_engine* e, const thmr_smpl_desc* s, void* stream) {
    if (!e || !s) return fail(e, THMR_ERR_INVALID, "null argument");
    if (!s->v_template || !s->shapedirs || !s->posedirs || !s->J_regressor || !s->lbs_weights || !s->J19_regressor ||
        !s->parents || !s->extra_verts || !s->joint_map)
        return fail(e, THMR_ERR_INVALID, "thmr_smpl_desc has a null field");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const hipMemcpyKind k = s->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_vt, s->v_template, sizeof(float) * NV * 3, k, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_sd, s->shapedirs, sizeof(float) * NV * 30, k, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_pd, s->posedirs, sizeof(float) * (size_t)NP * NV * 3, k, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_jr, s->J_regressor, sizeof(float) * NJ * NV, k, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_w, s->lbs_weights, sizeof(float) * NV * NJ, k, st));
    HIP_OK(hipMemcpyAsync(e->warena + e->o_smpl_j19, s->J19_regressor, sizeof(float) * 19 * NV, k, st));
    int32_t* ints = reinterpret_cast<int32_t*>(e->warena + e->o_smpl_int);
    HIP_OK(hipMemcpyAsync(ints, s->parents, sizeof(int32_t) * 24, k, st));
    HIP_OK(hipMemcpyAsync(ints + 24, s->extra_verts, sizeof(int32_t) * 21, k, st));
    HIP_OK(hipMemcpyAsync(ints + 48, s->joint_map, sizeof(int32_t) * 25, k, st));
    e->hips_host = s->update_hips ? 1 : 0;      // SMPL(update_hips=...), smpl_wrapper.py:11,33-36; lives in the arena (ints[80])
    HIP_OK(hipMemcpyAsync(ints + 80, &e->hips_host, sizeof(int32_t), hipMemcpyHostToDevice, st));
    e->smpl_loaded = true;
    e->finalized = false;
    return 0;
}

// split3 copies of the four ViT GEMM weights of every block (6 bytes per weight) + the activation operand buffers, engine-owned
static int build_split_weights(thmr_engine* e, hipStream_t st) {
    if (e->max_batch < kSplit3LowMinB) return 0;      // no call of this engine can reach the mode (one and two crops run the exact-fp32 kernels)
    const size_t kv_rows = (size_t)e->dec_depth * 2 * INNER;                                           // the decoder's to_kv of all layers
    if (!e->split_w) {
        // one copy per (device, weight arena): engines that share an arena share it (split_share())
        SplitShare& sh = split_share();
        std::lock_guard<std::mutex> lk(sh.mu);
        const size_t bytes = split_w_bytes_of(e->vit_depth, e->dec_depth);
        auto key = std::make_pair(e->cfg.device, static_cast<const void*>(e->warena));
        auto it = sh.m.find(key);
        if (it != sh.m.end() && it->second.bytes == bytes) {
            it->second.refs += 1;
            e->split_w = it->second.p;
        } else {
            if (it != sh.m.end())
                return fail(e, THMR_ERR_INVALID, "another engine with a different depth holds split3 copies of this weight arena");
            char* p = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&p), bytes) != hipSuccess)
                return fail(e, THMR_ERR_NOMEM, "hipMalloc(split3 ViT weights) failed");
            sh.m[key] = SplitShare::Ent{p, bytes, 1};
            e->split_w = p;
        }
        e->split_w_counted = true;
    }
    // activations: [M][1280] + [M][5120] split3 operands, then the two fp32 partial-sum planes of fc2's split-K ([2][M][1280])
    if (!e->split_act && hipMalloc(reinterpret_cast<void**>(&e->split_act), split_act_bytes_of(e->max_batch)) != hipSuccess)
        return fail(e, THMR_ERR_NOMEM, "hipMalloc(split3 activations) failed");
    if (!e->s3_ws && !e->no_persistent) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, e->cfg.device) == hipSuccess && prop.multiProcessorCount == 256) {
            // the persistent kernel's decomposition is 8 XCDs x 32 CUs; elsewhere the per-tile kernel stays in charge (same results)
            if (hipMalloc(&e->s3_ws, gemm_split3_persist_ws_bytes()) != hipSuccess) return fail(e, THMR_ERR_NOMEM, "hipMalloc(split3 hand-over workspace) failed");
            if (int r = reset_s3_workspace(e, st)) return r;
        }
    }
    e->vitw_s.resize(e->vit_depth);
    char* p = e->split_w;
    for (int i = 0; i < e->vit_depth; ++i) {
        const VitBlockW& w = e->vitw[i];
        auto conv = [&](const float* src, int rows, int K, const char*& out) -> int {
            out = p;
            const int rc = launch_split3(src, K, p, K, rows, K, st);
            p += (size_t)rows * K * 6;
            return rc;
        };
        LAUNCH_OK(conv(w.qkvw, 3 * DIM, DIM, e->vitw_s[i].qkv));
        LAUNCH_OK(conv(w.pw, DIM, DIM, e->vitw_s[i].proj));
        LAUNCH_OK(conv(w.f1w, MLP, DIM, e->vitw_s[i].fc1));
        LAUNCH_OK(conv(w.f2w, DIM, MLP, e->vitw_s[i].fc2));
    }
    e->kv_s = p;
    LAUNCH_OK(launch_split3(e->warena + e->o_kv_all, DIM, p, DIM, (int64_t)kv_rows, DIM, st));
    p += kv_rows * DIM * 6;
    e->pe_s = p;
    LAUNCH_OK(launch_split3(e->hot.pe_w, 768, p, 768, DIM, 768, st));
    return 0;
}

int thmr_finalize_weights(thmr_engine* e, int32_t assume_all_loaded, void* stream) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!assume_all_loaded) {
        for (auto& n : e->required)
            if (!e->slots[n].loaded) return fail(e, THMR_ERR_STATE, "missing tensor '" + n + "' (strict load)");
        if (!e->smpl_loaded) return fail(e, THMR_ERR_STATE, "SMPL constants not loaded (thmr_load_smpl)");
    }
    int32_t* flags = reinterpret_cast<int32_t*>(e->warena + e->o_flags);
    if (!assume_all_loaded) {
        if (int r = write_arena_constants(e, st)) return r;      // idempotent (load time wrote them already)
        size_t enc_loaded = 0;
        for (auto& n : e->enc_names) enc_loaded += e->slots[n].loaded ? 1 : 0;
        if (enc_loaded != 0 && enc_loaded != e->enc_names.size())
            return fail(e, THMR_ERR_STATE, "tokenizer encoder partially loaded: " + std::to_string(enc_loaded) + " of " +
                                               std::to_string(e->enc_names.size()) + " tensors");
        e->enc_ready = enc_loaded == e->enc_names.size();
    } else {
        // the arena was filled by someone else (a broadcast from the loading rank, or it is another engine's live arena): it must
        // carry the loader's magic word, otherwise the index tables / padding row / flags are uninitialised memory
        int32_t f[2] = {0, 0};
        HIP_OK(hipMemcpyAsync(f, flags, sizeof(f), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (f[1] != kArenaMagic)
            return fail(e, THMR_ERR_STATE, "thmr_finalize_weights(assume_all_loaded=1): the weight arena does not carry a loaded model "
                                           "(no loader magic word): was it broadcast before the root called thmr_load_weights, or never received?");
        e->enc_ready = f[0] == kEncMagic;
    }
    for (int i = 0; i < 9; ++i)
        LAUNCH_OK(launch_conv_repack(e->W(std::string(kConv3[i]) + ".weight"), e->warena + e->convp[i], kConv3Co[i], kConv3Ci[i], 3, st));
    LAUNCH_OK(launch_transpose(e->W("quantizer.codebook"), e->warena + e->o_cbT, 1, NCLS, CODE, st));
    LAUNCH_OK(launch_code_norm(e->W("quantizer.codebook"), e->warena + e->o_cnorm, NCLS, st));
    LAUNCH_OK(launch_lbs_jreg(e->warena + e->o_smpl_jr, e->warena + e->o_smpl_vt, e->warena + e->o_smpl_sd,
                              e->warena + e->o_smpl_jt, e->warena + e->o_smpl_jsd, st));
    LAUNCH_OK(launch_lbs_build_dirs(e->warena + e->o_smpl_sd, e->warena + e->o_smpl_pd, e->warena + e->o_smpl_dirs, st));
    if (e->enc_ready)
        for (int i = 0; i < kEncN; ++i)
            if (kEnc[i].ks > 1)
                LAUNCH_OK(launch_conv_repack_pad(e->W(std::string(kEnc[i].name) + ".weight"), e->warena + e->enc_convp[i],
                                                 kEnc[i].co, kEnc[i].ci, kEnc[i].cp, kEnc[i].ks, st));
    // per-block weight pointers of the ViT loop, resolved once (the name map is for load time, not for the hot path)
    e->vitw.resize(e->vit_depth);
    for (int i = 0; i < e->vit_depth; ++i) {
        const std::string p = "backbone.blocks." + std::to_string(i) + ".";
        VitBlockW& w = e->vitw[i];
        w.n1w = e->W(p + "norm1.weight"); w.n1b = e->W(p + "norm1.bias");
        w.qkvw = e->W(p + "attn.qkv.weight"); w.qkvb = e->W(p + "attn.qkv.bias");
        w.pw = e->W(p + "attn.proj.weight"); w.pb = e->W(p + "attn.proj.bias");
        w.n2w = e->W(p + "norm2.weight"); w.n2b = e->W(p + "norm2.bias");
        w.f1w = e->W(p + "mlp.fc1.weight"); w.f1b = e->W(p + "mlp.fc1.bias");
        w.f2w = e->W(p + "mlp.fc2.weight"); w.f2b = e->W(p + "mlp.fc2.bias");
    }
    {   // decoder / read-out / mixer_trans pointers of the persistent decoder kernel, resolved once
        DecParams& d = e->dec;
        const std::string T = "smpl_head.transformer.";
        for (int l = 0; l < e->dec_depth; ++l) {
            const std::string p = T + "transformer.layers." + std::to_string(l) + ".";
            DecLayerW& w = d.L[l];
            w.n0w = e->W(p + "0.norm.weight"); w.n0b = e->W(p + "0.norm.bias");
            w.wv = e->W(p + "0.fn.to_qkv.weight") + (size_t)2 * INNER * E;            // v slice of to_qkv (rows 1024..1535)
            w.wo1 = e->W(p + "0.fn.to_out.0.weight"); w.bo1 = e->W(p + "0.fn.to_out.0.bias");
            w.n1w = e->W(p + "1.norm.weight"); w.n1b = e->W(p + "1.norm.bias");
            w.wq = e->W(p + "1.fn.to_q.weight");
            w.wo2 = e->W(p + "1.fn.to_out.0.weight"); w.bo2 = e->W(p + "1.fn.to_out.0.bias");
            w.n2w = e->W(p + "2.norm.weight"); w.n2b = e->W(p + "2.norm.bias");
            w.w1 = e->W(p + "2.fn.net.0.weight"); w.b1 = e->W(p + "2.fn.net.0.bias");
            w.w2 = e->W(p + "2.fn.net.3.weight"); w.b2 = e->W(p + "2.fn.net.3.bias");
        }
        d.tok_bias = e->W(T + "to_token_embedding.bias"); d.pos = e->W(T + "pos_embedding");
        d.kv = e->S(e->so.big); d.ldkv = (int64_t)e->dec_depth * 2 * INNER;
        d.ro_w = e->warena + e->o_ro_w; d.ro_b = e->warena + e->o_ro_b;
        d.mt_w = e->W("smpl_head.decpose.mixer_trans.ff.0.weight"); d.mt_b = e->W("smpl_head.decpose.mixer_trans.ff.0.bias");
        d.dx = e->S(e->so.dx); d.dv = e->S(e->so.dv); d.dq = e->S(e->so.dq); d.dca = e->S(e->so.dca); d.dff = e->S(e->so.dff);
        d.ro = e->S(e->so.ro); d.mt = e->S(e->so.mt);
        d.sync = reinterpret_cast<unsigned*>(e->S(e->so.sync));
        d.depth = e->dec_depth; d.B = 0;
        { const char* tl = thmr_knob("THMR_DEC_TIMELINE"); d.timeline = tl && tl[0] == '1'; }
        { const char* ft = thmr_knob("THMR_DEC_FORCE_TIMEOUT"); d.debug_fail = ft && ft[0] == '1'; }
        // all-to-all barrier: measured SLOWER (head 0.711 vs 0.664 ms at one crop, 2.87-2.90 vs 2.81-2.82 at 64: 128-256 workgroups x
        // 128-256 device-scope polls contend; profiles/r3k_decoder_barrier_all_to_all_ab.log) — kept behind the knob only
        { const char* bm = thmr_knob("THMR_DEC_BARRIER"); d.barrier_a2a = bm && bm[0] == '1'; }
        {
            // never more workgroups than can be resident at once (occupancy query x CUs): the grid barrier depends on it
            const int nb = decoder_max_coresident_blocks(e->cfg.device);
            if (nb < 1) return fail(e, THMR_ERR_HIP, "persistent decoder kernel cannot be resident on this device (occupancy query failed)");
            d.max_blocks = nb;
            d.host_err = e->host_err;
        }
        MixerParams& m = e->mix;
        const std::string C = "smpl_head.decpose.";
        for (int i = 0; i < MIX; ++i) {
            const std::string p = C + "mixer_head." + std::to_string(i) + ".";
            MixerLayerW& w = m.L[i];
            w.ln1w = e->W(p + "layernorm1.weight"); w.ln1b = e->W(p + "layernorm1.bias");
            w.wt1 = e->W(p + "MLP_token.ff.0.weight"); w.bt1 = e->W(p + "MLP_token.ff.0.bias");
            w.wt2 = e->W(p + "MLP_token.ff.3.weight"); w.bt2 = e->W(p + "MLP_token.ff.3.bias");
            w.ln2w = e->W(p + "layernorm2.weight"); w.ln2b = e->W(p + "layernorm2.bias");
            w.wc1 = e->W(p + "MLP_channel.ff.0.weight"); w.bc1 = e->W(p + "MLP_channel.ff.0.bias");
            w.wc2 = e->W(p + "MLP_channel.ff.3.weight"); w.bc2 = e->W(p + "MLP_channel.ff.3.bias");
        }
        m.tln_w = e->W(C + "mixer_trans.ff.1.weight"); m.tln_b = e->W(C + "mixer_trans.ff.1.bias");
        m.wn = e->W(C + "mixer_norm_layer.ff.0.weight"); m.bn = e->W(C + "mixer_norm_layer.ff.0.bias");
        m.nln_w = e->W(C + "mixer_norm_layer.ff.1.weight"); m.nln_b = e->W(C + "mixer_norm_layer.ff.1.bias");
        m.mt = e->S(e->so.mt); m.out = e->S(e->so.nl2);
        auto& h = e->hot;
        h.pe_w = e->W("backbone.patch_embed.proj.weight"); h.pe_b = e->W("backbone.patch_embed.proj.bias");
        h.pos = e->W("backbone.pos_embed");
        h.lastn_w = e->W("backbone.last_norm.weight"); h.lastn_b = e->W("backbone.last_norm.bias");
        h.cls_w = e->W(C + "class_pred_layer.weight"); h.cls_b = e->W(C + "class_pred_layer.bias");
        h.init_pose = e->W("smpl_head.init_body_pose"); h.init_betas = e->W("smpl_head.init_betas"); h.init_cam = e->W("smpl_head.init_cam");
        for (int i = 0; i < 9; ++i) h.conv_b[i] = e->W(std::string(kConv3[i]) + ".bias");
        for (int b = 0; b < 2; ++b) {
            const std::string p = "decoder.decoder.14.0.model." + std::to_string(b) + ".";
            h.res_w[b] = e->W(p + "conv2.weight"); h.res_b[b] = e->W(p + "conv2.bias");
        }
    }
    if (e->vit_gemm_mode == 1) {                                       // weights were (re)loaded with the split3 mode on
        if (int r = build_split_weights(e, st)) return r;              // (on failure the engine stays un-finalized: no forward can run on half-built operands)
        HIP_OK(hipStreamSynchronize(st));                              // as in thmr_set_vit_gemm: visible to forwards on any stream
    }
    e->finalized = true;
    return 0;
}

int thmr_set_vit_gemm(thmr_engine* e, int32_t mode, void* stream) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    if (mode != 0 && mode != 1) return fail(e, THMR_ERR_INVALID, "vit gemm mode must be 0 (exact-fp32 MFMA) or 1 (split3 on the bf16 matrix pipe)");
    if (mode == 1) {
        if (!e->finalized) return fail(e, THMR_ERR_STATE, "thmr_set_vit_gemm(1) needs finalized weights (thmr_finalize_weights)");
        if (hipSetDevice(e->cfg.device) != hipSuccess) return fail(e, THMR_ERR_HIP, "hipSetDevice failed");
        if (int r = build_split_weights(e, static_cast<hipStream_t>(stream))) return r;
        // the conversion (3.8 GB of writes) was enqueued on the caller's stream; a forward on ANOTHER stream right after this call must
        // not read half-converted weights: a one-time switch, so simply wait for it
        HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    }
    e->vit_gemm_mode = mode;
    return 0;
}

int thmr_get_vit_gemm(thmr_engine* e) { return e ? e->vit_gemm_mode : -1; }

int thmr_vq_decode(thmr_engine* e, const float* probs_dev, int32_t B, float* pose6d_dev, void* stream) {
    if (int r = check_ready(e, B)) return r;
    if (!probs_dev || !pose6d_dev) return fail(e, THMR_ERR_INVALID, "null buffer");
    return vq_decode(e, probs_dev, B, pose6d_dev, static_cast<hipStream_t>(stream));
}

// EncodeTokens.forward (tokenization/models/vanilla_pose_vqvae.py:334-342): PoseSPEncoderV1 (:66-111) -> preprocess
// (quantize_cnn.py:74-78) -> QuantizeEMAReset.quantize (:80-86).  pose (B,21,6) rot6d body pose -> idx (B,160).
int thmr_encode_tokens(thmr_engine* e, const float* pose_dev, int32_t B, int32_t* idx_dev, float* latent_dev, void* stream) {
    if (int r = check_ready(e, B)) return r;
    if (!e->enc_ready) return fail(e, THMR_ERR_STATE, "tokenizer encoder weights ('encoder.encoder.*') were not loaded");
    if (!pose_dev || !idx_dev) return fail(e, THMR_ERR_INVALID, "null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // all scratch comes out of the big time-shared buffer (the encode path never overlaps a forward on one engine)
    float* gat = e->S(e->so.big);
    float* a0 = gat + (size_t)B * 320 * 1536;
    float* a1 = a0 + (size_t)B * 320 * VQW;
    float* a2 = a1 + (size_t)B * 320 * VQW;
    const int32_t* tab = reinterpret_cast<const int32_t*>(e->warena + e->o_idx_enc);
    auto W = [&](int i) { return kEnc[i].ks > 1 ? e->warena + e->enc_convp[i] : e->W(std::string(kEnc[i].name) + ".weight"); };
    auto Bv = [&](int i) { return e->W(std::string(kEnc[i].name) + ".bias"); };
    // 0: Conv1d(6->512,k3,p1)+ReLU at T=21 (channels zero-padded 6->32 so that K = 96)
    LAUNCH_OK(launch_conv_gather_general(pose_dev, gat, nullptr, B, 21, 21, 21, 6, 32, 3, 1, 1, st));
    {
        GemmArgs a = mk(gat, 96, W(0), 96, Bv(0), nullptr, 0, a0, VQW, B * 21, VQW, 96);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS_RELU, -1, st));
    }
    // 1-4: nearest resample (21->40, then x2 three times) + Conv1d(512,k3,p1) + ReLU
    const int tin[4] = {21, 40, 80, 160}, tout[4] = {40, 80, 160, 320}, toff[4] = {0, 40, 120, 280};
    float* cur = a0;
    float* nxt = a1;
    for (int i = 0; i < 4; ++i) {
        LAUNCH_OK(launch_conv3_gather(cur, gat, tab + toff[i], B, tin[i], tout[i], VQW, 1, 0, st));
        GemmArgs a = mk(gat, 3 * VQW, W(1 + i), 3 * VQW, Bv(1 + i), nullptr, 0, nxt, VQW, B * tout[i], VQW, 3 * VQW);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS_RELU, -1, st));
        std::swap(cur, nxt);
    }
    // 5: Conv1d(512,512,k4,s2,p1): 320 -> 160, no activation
    LAUNCH_OK(launch_conv_gather_general(cur, gat, nullptr, B, 320, 320, 160, VQW, VQW, 4, 2, 1, st));
    {
        GemmArgs a = mk(gat, 4 * VQW, W(5), 4 * VQW, Bv(5), nullptr, 0, nxt, VQW, B * 160, VQW, 4 * VQW);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS, -1, st));
        std::swap(cur, nxt);
    }
    // 6-9: Resnet1D (dilation 3 then 1): x + conv2(relu(conv1(relu(x))))   (resnet.py:49-69)
    for (int blk = 0; blk < 2; ++blk) {
        const int c1 = 6 + 2 * blk, c2 = 7 + 2 * blk, dil = blk == 0 ? 3 : 1;
        LAUNCH_OK(launch_conv3_gather(cur, gat, nullptr, B, 160, 160, VQW, dil, 1, st));
        GemmArgs g1 = mk(gat, 3 * VQW, W(c1), 3 * VQW, Bv(c1), nullptr, 0, a2, VQW, B * 160, VQW, 3 * VQW);
        LAUNCH_OK(launch_gemm(g1, EPI_BIAS_RELU, -1, st));
        GemmArgs g2 = mk(a2, VQW, W(c2), VQW, Bv(c2), cur, VQW, nxt, VQW, B * 160, VQW, VQW);
        LAUNCH_OK(launch_gemm(g2, EPI_BIAS_RESID, -1, st));
        std::swap(cur, nxt);
    }
    // 10: Conv1d(512->256,k3,p1): the latent, already in the (N*T, C) layout QuantizeEMAReset.preprocess produces
    float* lat = latent_dev ? latent_dev : a2;
    LAUNCH_OK(launch_conv3_gather(cur, gat, nullptr, B, 160, 160, VQW, 1, 0, st));
    {
        GemmArgs a = mk(gat, 3 * VQW, W(10), 3 * VQW, Bv(10), nullptr, 0, lat, CODE, B * 160, CODE, 3 * VQW);
        LAUNCH_OK(launch_gemm(a, EPI_BIAS, -1, st));
    }
    // argmin-L2 against the codebook; the x.C^T scores go to the gather region (B*160*2048 <= B*320*1536)
    GemmArgs d = mk(lat, CODE, e->W("quantizer.codebook"), CODE, nullptr, nullptr, 0, gat, NCLS, B * 160, NCLS, CODE);
    LAUNCH_OK(launch_gemm(d, EPI_NONE, -1, st));
    LAUNCH_OK(launch_vq_argmin_rows(lat, gat, e->warena + e->o_cnorm, idx_dev, nullptr, B * 160, st));
    return 0;
}

int thmr_weight_arena(thmr_engine* e, void** ptr_dev, size_t* bytes) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    if (ptr_dev) *ptr_dev = e->warena;
    if (bytes) *bytes = e->wfloats * sizeof(float);
    return 0;
}

int thmr_engine_status(thmr_engine* e, void* stream) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned words[4] = {0, 0, 0, 0};
    HIP_OK(hipMemcpyAsync(words, e->sarena + e->so.sync, sizeof(words), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (words[3] != 0 || (e->host_err && *static_cast<volatile unsigned*>(e->host_err) != 0)) return recover_decoder_timeout(e);
    if (e->s3_ws) {
        // persistent split3 GEMM: a consumer's bounded wait for a hand-over slab ran out (see recover_split3_timeout; the next forward-type
        // call reports it too, through the host-mapped copy of this word).  Reported once.
        unsigned err = 0;
        if (gemm_split3_persist_error(e->s3_ws, st, &err) != 0) return fail(e, THMR_ERR_HIP, "reading the split3 hand-over error word failed");
        if (e->s3_host_err && *static_cast<volatile unsigned*>(e->s3_host_err) != 0) err = 1;
#ifdef THMR_EXPERIMENTS
        // tests only: report a hand-over timeout that did not happen, once per engine, to exercise the recovery
        if (e->s3_persist && e->vit_gemm_mode == 1 && !e->s3_forced_once) {
            const char* f = thmr_knob("THMR_SPLIT3_FORCE_TIMEOUT");
            if (f && f[0] == '1') { err = 1; e->s3_forced_once = true; }
        }
#endif
        if (err != 0) return recover_split3_timeout(e);
    }
    return 0;
}

int thmr_debug_decoder_timeline(thmr_engine* e, uint64_t* stamps_host, int32_t max_stamps, void* stream) {
    if (!e || !stamps_host || max_stamps < 1 || max_stamps > 240) return fail(e, THMR_ERR_INVALID, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_OK(hipMemcpyAsync(stamps_host, e->sarena + e->so.sync + 16, sizeof(uint64_t) * max_stamps, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

// a forward-type call = one turn of the per-device turnstile (struct Turn): r = the call's own result, then the turn's end
static int end_turn(thmr_engine* e, Turn& turn, int r) {
    const int t = turn.end();
    if (r) return r;
    return t ? fail(e, THMR_ERR_HIP, "hipEventRecord failed at the end of the call (per-device engine turnstile)") : 0;
}
#define THMR_TURN(turn, e, st)                                                                                         \
    Turn turn(e, st);                                                                                                  \
    if (turn.rc) return fail(e, THMR_ERR_HIP, "the per-device engine turnstile could not create / wait for its event")

int thmr_vit_forward(thmr_engine* e, const float* img_dev, int32_t B, float* feats_dev, void* stream) {
    if (int r = check_ready(e, B, static_cast<hipStream_t>(stream))) return r;
    if (!img_dev || !feats_dev) return fail(e, THMR_ERR_INVALID, "null buffer");
    THMR_TURN(turn, e, static_cast<hipStream_t>(stream));
    return end_turn(e, turn, vit_forward(e, img_dev, B, feats_dev, static_cast<hipStream_t>(stream)));
}

static int head_forward_call(thmr_engine* e, const float* ctx_dev, int32_t B, const thmr_outputs* out, hipStream_t st) {
    if (int r = head_forward(e, ctx_dev, B, out, st)) return r;
    if (out && (out->pred_vertices || out->pred_keypoints_3d || out->pred_keypoints_2d)) {
        const float* rot = out->rotmat ? out->rotmat : e->S(e->so.rot);
        const float* betas = out->betas ? out->betas : e->S(e->so.betas);
        const float* camt = out->pred_cam_t ? out->pred_cam_t : e->S(e->so.camt);
        return lbs(e, rot, betas, camt, B, out->pred_vertices, out->pred_keypoints_3d, out->pred_keypoints_2d, st);
    }
    return 0;
}

int thmr_head_forward(thmr_engine* e, const float* ctx_dev, int32_t B, const thmr_outputs* out, void* stream) {
    if (int r = check_ready(e, B, static_cast<hipStream_t>(stream))) return r;
    if (!ctx_dev) return fail(e, THMR_ERR_INVALID, "null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    THMR_TURN(turn, e, st);
    return end_turn(e, turn, head_forward_call(e, ctx_dev, B, out, st));
}

static int forward_call(thmr_engine* e, const float* img_dev, int32_t B, const thmr_outputs* out, hipStream_t st) {
    float* ctx = e->S(e->so.h);
    if (int r = vit_forward(e, img_dev, B, nullptr, st)) return r;
    if (out->vit_features)
        HIP_OK(hipMemcpyAsync(out->vit_features, ctx, sizeof(float) * (size_t)B * TOK * DIM, hipMemcpyDeviceToDevice, st));
    if (int r = head_forward(e, ctx, B, out, st)) return r;
    const float* rot = out->rotmat ? out->rotmat : e->S(e->so.rot);
    const float* betas = out->betas ? out->betas : e->S(e->so.betas);
    const float* camt = out->pred_cam_t ? out->pred_cam_t : e->S(e->so.camt);
#ifdef THMR_EXPERIMENTS
    // tests only (THMR_SPLIT3_FORCE_TIMEOUT=2): write what a timed-out hand-over consumer writes into the host-mapped error word, once per
    // engine, so that the NEXT forward-type call goes through check_ready's report-reset-fall-back path
    if (e->s3_ws && e->s3_persist && e->vit_gemm_mode == 1 && !e->s3_forced_once && e->s3_host_err) {
        const char* f = thmr_knob("THMR_SPLIT3_FORCE_TIMEOUT");
        if (f && f[0] == '2') { *e->s3_host_err = 1; e->s3_forced_once = true; }
    }
#endif
    return lbs(e, rot, betas, camt, B, out->pred_vertices, out->pred_keypoints_3d, out->pred_keypoints_2d, st);
}

int thmr_forward(thmr_engine* e, const float* img_dev, int32_t B, const thmr_outputs* out, void* stream) {
    if (int r = check_ready(e, B, static_cast<hipStream_t>(stream))) return r;
    if (!img_dev || !out) return fail(e, THMR_ERR_INVALID, "null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    THMR_TURN(turn, e, st);
    return end_turn(e, turn, forward_call(e, img_dev, B, out, st));
}

int thmr_lbs_forward(thmr_engine* e, const float* rotmat_dev, const float* betas_dev, const float* cam_dev, int32_t B,
                     float* verts_dev, float* joints_dev, float* cam_t_dev, float* kp2d_dev, void* stream) {
    if (int r = check_ready(e, B)) return r;
    if (!rotmat_dev || !betas_dev) return fail(e, THMR_ERR_INVALID, "null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* camt = nullptr;
    if (cam_dev) {
        float* ct = cam_t_dev ? cam_t_dev : e->S(e->so.camt);
        LAUNCH_OK(launch_cam_t(cam_dev, ct, FOCAL, IMG, B, st));
        camt = ct;
    }
    return lbs(e, rotmat_dev, betas_dev, camt, B, verts_dev, joints_dev, camt ? kp2d_dev : nullptr, st);
}

int thmr_vq_argmin(thmr_engine* e, const float* x_dev, int32_t rows, int32_t* idx_dev, float* dist_dev, void* stream) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    if (!e->finalized) return fail(e, THMR_ERR_STATE, "weights not finalized");
    if (!x_dev || !idx_dev || rows < 1) return fail(e, THMR_ERR_INVALID, "bad argument");
    if ((size_t)rows * NCLS > (size_t)e->max_batch * TOK * 6144) return fail(e, THMR_ERR_INVALID, "rows exceed scratch capacity");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* dot = e->S(e->so.big);
    GemmArgs a = mk(x_dev, CODE, e->W("quantizer.codebook"), CODE, nullptr, nullptr, 0, dot, NCLS, rows, NCLS, CODE);
    LAUNCH_OK(launch_gemm(a, EPI_NONE, -1, st));
    LAUNCH_OK(launch_vq_argmin_rows(x_dev, dot, e->warena + e->o_cnorm, idx_dev, dist_dev, rows, st));
    return 0;
}

// ---- stateless operator entry points ----
int thmr_op_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* resid, float* C, int64_t ldc,
                 int32_t M, int32_t N, int32_t K, int32_t epi, float qscale, int32_t qcols, int32_t variant, void* stream) {
    thmr_engine* e = nullptr;
    if (!A || !W || !C) return fail(e, THMR_ERR_INVALID, "null buffer");
    if (epi < 0 || epi >= EPI_NUM) return fail(e, THMR_ERR_INVALID, "bad epilogue id");
    if (epi != EPI_NONE && !bias) return fail(e, THMR_ERR_INVALID, "epilogue needs bias");
    if ((epi == EPI_BIAS_RESID || epi == EPI_BIAS_POS) && !resid) return fail(e, THMR_ERR_INVALID, "epilogue needs resid");
    GemmArgs a = mk(A, lda, W, K, bias, resid, ldc, C, ldc, M, N, K);
    a.qscale = qscale; a.qcols = qcols;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (variant >= 100 && variant < 120) {
        // small-M ring kernel: 100 + 10*(ring == 8) + log2(ksplit).  The stateless entry point keeps a grow-only partial-sum
        // workspace per device and stream (the engine uses its own scratch arena instead).
        const int ring = variant >= 110 ? 8 : 4, ksplit = 1 << (variant % 10);
        if (epi == EPI_BIAS_POS) return fail(e, THMR_ERR_INVALID, "ring GEMM has no pos-embed epilogue");
        float* ws = nullptr;
        static std::mutex mu;
        std::unique_lock<std::mutex> lk(mu, std::defer_lock);   // held across the launches that use the workspace
        if (ksplit > 1) {
            // grow-only workspace per (DEVICE, STREAM) — launches on one stream are ordered, so they may share a buffer;
            // different streams never do — guarded by a mutex; the device is synchronised before a buffer is replaced
            static std::map<std::pair<int, void*>, std::pair<float*, size_t>> pool;
            int dev = 0;
            HIP_OK(hipGetDevice(&dev));
            lk.lock();
            auto& slot = pool[{dev, stream}];
            const size_t need = (size_t)ksplit * M * N;
            if (need > slot.second) {
                if (slot.first) { HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(slot.first)); slot = {nullptr, 0}; }
                float* p = nullptr;
                HIP_OK(hipMalloc(&p, need * sizeof(float)));
                slot = {p, need};
            }
            ws = slot.first;
        }
        LAUNCH_OK(launch_gemm_ring(a, epi, ring, ksplit, ws, st));
        if (ksplit > 1) LAUNCH_OK(launch_splitk_epilogue(a, epi, ws, ksplit, st));
    } else if (variant >= 200 && variant < 500) {
        // split-K on the big LDS-DMA tiles: 200 + tile (7 / 8 / 10, 0 = cost model) = 2 ways, 400 + tile = 4 ways; partial sums in a
        // grow-only workspace, then the fixed-order reduce + epilogue (what the engine fuses into its residual + LayerNorm kernel)
        const int ksplit = variant >= 400 ? 4 : 2, tile = variant % 100;
        if (epi == EPI_BIAS_POS) return fail(e, THMR_ERR_INVALID, "split-K GEMM has no pos-embed epilogue");
        static std::mutex mu2;
        static std::map<std::pair<int, void*>, std::pair<float*, size_t>> pool2;
        int dev = 0;
        HIP_OK(hipGetDevice(&dev));
        std::unique_lock<std::mutex> lk(mu2);
        auto& slot = pool2[{dev, stream}];
        const size_t need = (size_t)ksplit * M * N;
        if (need > slot.second) {
            if (slot.first) { HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(slot.first)); slot = {nullptr, 0}; }
            float* p = nullptr;
            HIP_OK(hipMalloc(&p, need * sizeof(float)));
            slot = {p, need};
        }
        LAUNCH_OK(launch_gemm_splitk(a, tile == 0 ? -1 : tile, ksplit, slot.first, st));
        LAUNCH_OK(launch_splitk_epilogue(a, epi, slot.first, ksplit, st));
    } else if (variant == 120) {
        if (epi == EPI_BIAS_POS || epi == EPI_BIAS_RESID) return fail(e, THMR_ERR_INVALID, "ring16 GEMM: epilogues none / bias / gelu / relu / qscale only");
        LAUNCH_OK(launch_gemm_ring16(a, epi, st));
    } else if (variant == 2) {
        LAUNCH_OK(launch_gemm_skinny(a, epi, st));
    } else {
        LAUNCH_OK(launch_gemm(a, epi, variant, st));
    }
    return 0;
}

int thmr_op_split3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int32_t K, void* stream) {
    thmr_engine* e = nullptr;
    if (!src || !dst) return fail(e, THMR_ERR_INVALID, "null buffer");
    if (rows <= 0 || K <= 0 || (K % 8) != 0 || (ld_src % 4) != 0 || (ld_dst % 8) != 0 || ld_dst < K || ld_src < K)
        return fail(e, THMR_ERR_INVALID, "split3: K % 8, ld_src % 4, ld_dst % 8 must be 0 and the strides >= K");
    LAUNCH_OK(launch_split3(src, ld_src, dst, ld_dst, rows, K, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_gemm_split3(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, float* C,
                        int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, float qscale, int32_t qcols, int32_t variant,
                        void* stream) {
    thmr_engine* e = nullptr;
    if (!A || !W || !C) return fail(e, THMR_ERR_INVALID, "null buffer");
    if (epi != EPI_NONE && epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID && epi != EPI_BIAS_QSCALE && epi != EPI_BIAS_POS)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: epilogue must be 0, 1, 2, 4, 5 or 6");
    if (epi != EPI_NONE && !bias) return fail(e, THMR_ERR_INVALID, "epilogue needs bias");
    if ((epi == EPI_BIAS_RESID || epi == EPI_BIAS_POS) && !resid) return fail(e, THMR_ERR_INVALID, "epilogue needs resid");
    if (epi == EPI_BIAS_POS && ((N % 4) != 0 || (variant != -1 && variant != 0 && variant != 2 && variant != 6 && variant != 8 && variant != 9 && variant != 10 && variant != 11))) return fail(e, THMR_ERR_INVALID, "split3 GEMM: the pos-embed epilogue needs N % 4 == 0 and a per-tile variant (-1, 0, 2, 6, 8, 9)");
    if (M <= 0 || N <= 0 || K <= 0 || (K % 32) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 || lda < K || ldw < K || ldc < N)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: K % 32 == 0, lda / ldw multiples of 8 and >= K, ldc >= N");
    // + 1000: A is a ROW-BLOCKED split3 operand ([M / 32][K / 8][3][32][8], rows padded to 32; GemmArgs::a_blk) — tiles 0 / 2, split-K 202 / 204
    // and the persistent kernel 300, epilogues 0 and 4 (what fc2 runs)
    int a_blk = 0;
    if (variant >= 1000) {
        a_blk = 1;
        variant -= 1000;
        if ((variant != 0 && variant != 2 && variant != 6 && variant != 8 && variant != 9 && variant != 202 && variant != 204 && variant != 300) || (epi != EPI_NONE && epi != EPI_BIAS_RESID))
            return fail(e, THMR_ERR_INVALID, "split3 GEMM with a row-blocked A: variants 1000, 1002, 1006, 1008, 1202, 1204, 1300 and epilogues 0 / 4 only");
    }
    if (!(variant >= -1 && variant <= 11) && variant != 31 && variant != 32 && variant != 34 && variant != 37 && !(variant >= 100 && variant <= 102) &&
        variant != 202 && variant != 204 && variant != 300 && variant != 320 && variant != 322 && variant != 324 && variant != 20 && variant != 22 && variant != 310)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: variant -1 (rule), 0, 2, 6 (128 x 128 on 8 waves), 8 (128 x 128, three-stage ring), 5 / 7 (half-tile tail on 4 / 8 waves), 202, 204, 300; experiments build: 1, 4, 20, 22, 100-102, 310 (3, 31, 32, 34, 37: schedule experiments, epilogue 0 only)");
    GemmArgs a = mk(static_cast<const float*>(A), lda, static_cast<const float*>(W), ldw, bias, resid, ldc, C, ldc, M, N, K);
    a.qscale = qscale; a.qcols = qcols;
    a.a_blk = a_blk;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (variant == 320) {      // round 6: the persistent stream over 128 x 128 tiles with the three-stage ring
        if (!gemm_split3_persist_narrow_ok(a)) return fail(e, THMR_ERR_INVALID, "persistent 128 x 128 split3 GEMM: N % 128 == 0, >= 256 tiles, K >= 96, row-major A");
        if (!device_has_256_cus()) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: its 8 x 32 workgroup decomposition needs a 256-CU device");
        void* ws = gemm_split3_persist_op_ws(st);
        if (!ws) return fail(e, THMR_ERR_NOMEM, "persistent split3 GEMM: workspace allocation failed");
        LAUNCH_OK(launch_gemm_split3_persist_narrow(a, epi, ws, st));
        return 0;
    }
    if (variant == 300 || variant == 310) {
        // 256 persistent workgroups over a tile stream: M % 128 == 0, N % 256 == 0, at least 256 tiles.  300 = the product kernel (gemm_split16.hip),
        // 310 = the round-4 first version on 32x32x16 MFMAs (gemm_split_persist.hip; experiments build)
        if (!gemm_split3_persist_ok(a)) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: M % 128 == 0, N % 256 == 0, M / 128 * N / 256 >= 256, K >= 64");
        if (!device_has_256_cus()) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: its 8 x 32 workgroup decomposition needs a 256-CU device (use variant 0 / 2)");
        void* ws = gemm_split3_persist_op_ws(st);
        if (!ws) return fail(e, THMR_ERR_NOMEM, "persistent split3 GEMM: workspace allocation failed");
        LAUNCH_OK(launch_gemm_split3_persist(a, epi, variant == 300 ? 0 : 10, ws, st));
        return 0;
    }
    if (variant == 202 || variant == 204 || variant == 322 || variant == 324) {
        // split-K 2 / 4 on the big tiles (202 / 204) or as (tile, K slice) units of the 128 x 128 stream (322 / 324); partial sums in a grow-only
        // workspace per (device, stream), then the fixed-order reduce + epilogue
        const bool stream_k = variant >= 300;
        const int ksplit = variant - (stream_k ? 320 : 200);
        if ((K % (32 * ksplit)) != 0) return fail(e, THMR_ERR_INVALID, "split3 split-K GEMM: K must be a multiple of 32 * ksplit");
        static std::mutex mu4;
        static std::map<std::pair<int, void*>, std::pair<float*, size_t>> pool4;
        int dev = 0;
        HIP_OK(hipGetDevice(&dev));
        std::unique_lock<std::mutex> lk(mu4);
        auto& slot = pool4[{dev, stream}];
        const size_t need = (size_t)ksplit * M * N;
        if (need > slot.second) {
            if (slot.first) { HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(slot.first)); slot = {nullptr, 0}; }
            float* p = nullptr;
            HIP_OK(hipMalloc(&p, need * sizeof(float)));
            slot = {p, need};
        }
        if (stream_k) {
            void* ws = gemm_split3_persist_op_ws(st);
            if (!ws) return fail(e, THMR_ERR_NOMEM, "persistent split3 GEMM: workspace allocation failed");
            if (launch_gemm_split3_splitk_stream(a, ksplit, slot.first, ws, st) != 0)
                return fail(e, THMR_ERR_INVALID, "split-K through the 128 x 128 stream: N % 128 == 0, >= 256 (tile, slice) units, >= 3 K tiles per slice, row-major A");
        } else LAUNCH_OK(launch_gemm_split3_splitk(a, ksplit, slot.first, st));
        LAUNCH_OK(launch_splitk_epilogue(a, epi, slot.first, ksplit, st));
        return 0;
    }
#ifdef THMR_EXPERIMENTS
    if (variant >= 100) {
        // small-M ring kernel, split-K 2^(variant - 100); partial sums in a grow-only workspace per (device, stream), then the fixed-order
        // reduce + epilogue (the engine fuses that into its residual + LayerNorm kernel)
        const int ksplit = 1 << (variant - 100);
        if ((K % (32 * ksplit)) != 0) return fail(e, THMR_ERR_INVALID, "split3 ring GEMM: K must be a multiple of 32 * ksplit");
        float* ws = nullptr;
        static std::mutex mu3;
        static std::map<std::pair<int, void*>, std::pair<float*, size_t>> pool3;
        std::unique_lock<std::mutex> lk(mu3, std::defer_lock);
        if (ksplit > 1) {
            int dev = 0;
            HIP_OK(hipGetDevice(&dev));
            lk.lock();
            auto& slot = pool3[{dev, stream}];
            const size_t need = (size_t)ksplit * M * N;
            if (need > slot.second) {
                if (slot.first) { HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(slot.first)); slot = {nullptr, 0}; }
                float* p = nullptr;
                HIP_OK(hipMalloc(&p, need * sizeof(float)));
                slot = {p, need};
            }
            ws = slot.first;
        }
        LAUNCH_OK(launch_gemm_split3_ring(a, epi, ksplit, ws, st));
        if (ksplit > 1) LAUNCH_OK(launch_splitk_epilogue(a, epi, ws, ksplit, st));
        return 0;
    }
#else
    if (variant >= 100 || variant == 1 || variant == 3 || variant == 4 || variant == 6 || variant == 7 || variant == 9 || variant > 10)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: this variant exists only in the experiments build (libtokenhmr_hip_exp.so)");
#endif
    LAUNCH_OK(launch_gemm_split3(a, epi, variant, st));
    return 0;
}

int thmr_op_gemm_split3_out_split3(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Cs, int64_t ldcs,
                                   int32_t M, int32_t N, int32_t K, int32_t epi, float qscale, int32_t qcols, int32_t variant, void* stream) {
    thmr_engine* e = nullptr;
    if (!A || !W || !Cs) return fail(e, THMR_ERR_INVALID, "null buffer");
    if (epi != EPI_NONE && epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_BIAS_QSCALE)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM with split3 output: epilogue must be 0, 1, 2 or 5");
    if (epi != EPI_NONE && !bias) return fail(e, THMR_ERR_INVALID, "epilogue needs bias");
    if (M <= 0 || N <= 0 || K <= 0 || (K % 32) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 || lda < K || ldw < K || (N % 8) != 0 ||
        (ldcs % 8) != 0 || ldcs < N)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: K % 32 == 0, N % 8 == 0, lda / ldw / ldcs multiples of 8 and >= K / K / N");
    // + 1000: the result in the ROW-BLOCKED form ([M / 32][N / 8][3][32][8], Cs holds ceil(M / 32) * 32 rows; GemmArgs::cs_blk)
    int cs_blk = 0;
    if (variant >= 1000) {
        cs_blk = 1;
        variant -= 1000;
    }
    if ((variant < -1 || variant > 2) && variant != 4 && variant != 5 && variant != 6 && variant != 7 && variant != 8 && variant != 9 && variant != 10 && variant != 11 && variant != 100 && variant != 20 && variant != 22 && variant != 302 && variant != 320 && variant != 311 && variant != 312)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: variant -1 (rule), 0, 2, 6, 5 / 7 (half-tile tail), 302 (persistent workgroups); experiments build: 1, 4, 20, 22, 100 (ring kernel), 311 / 312 (32x32x16 persistent kernel: LDS / swapped-role epilogue)");
    GemmArgs a = mk(static_cast<const float*>(A), lda, static_cast<const float*>(W), ldw, bias, nullptr, 0, nullptr, 0, M, N, K);
    a.qscale = qscale; a.qcols = qcols;
    a.c_split = Cs; a.ldcs = ldcs;
    a.cs_blk = cs_blk;
    if (variant == 320) {
        if (!gemm_split3_persist_narrow_ok(a)) return fail(e, THMR_ERR_INVALID, "persistent 128 x 128 split3 GEMM: N % 128 == 0, >= 256 tiles, K >= 96, row-major A");
        if (!device_has_256_cus()) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: its 8 x 32 workgroup decomposition needs a 256-CU device");
        void* ws = gemm_split3_persist_op_ws(static_cast<hipStream_t>(stream));
        if (!ws) return fail(e, THMR_ERR_NOMEM, "persistent split3 GEMM: workspace allocation failed");
        LAUNCH_OK(launch_gemm_split3_persist_narrow(a, epi, ws, static_cast<hipStream_t>(stream)));
        return 0;
    }
    if (variant >= 302) {
#ifndef THMR_EXPERIMENTS
        if (variant != 302) return fail(e, THMR_ERR_INVALID, "split3 GEMM: this persistent variant exists only in the experiments build");
#endif
        if (variant != 302 && epi != EPI_NONE && epi != EPI_BIAS_GELU) return fail(e, THMR_ERR_INVALID, "32x32x16 persistent split3 GEMM with split3 output: epilogue must be 0 or 2");
        if (!gemm_split3_persist_ok(a)) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: M % 128 == 0, N % 256 == 0, M / 128 * N / 256 >= 256, K >= 64");
        if (!device_has_256_cus()) return fail(e, THMR_ERR_INVALID, "persistent split3 GEMM: its 8 x 32 workgroup decomposition needs a 256-CU device (use variant 0 / 2)");
        void* ws = gemm_split3_persist_op_ws(static_cast<hipStream_t>(stream));
        if (!ws) return fail(e, THMR_ERR_NOMEM, "persistent split3 GEMM: workspace allocation failed");
        LAUNCH_OK(launch_gemm_split3_persist(a, epi, variant == 302 ? 2 : variant - 300, ws, static_cast<hipStream_t>(stream)));
        return 0;
    }
#ifdef THMR_EXPERIMENTS
    if (variant == 100) {
        LAUNCH_OK(launch_gemm_split3_ring(a, epi, 1, nullptr, static_cast<hipStream_t>(stream)));
        return 0;
    }
#else
    if (variant == 100 || variant == 1 || variant == 4 || variant == 6 || variant == 7 || variant == 9 || variant == 11 || variant >= 20)
        return fail(e, THMR_ERR_INVALID, "split3 GEMM: this variant exists only in the experiments build (libtokenhmr_hip_exp.so)");
#endif
    LAUNCH_OK(launch_gemm_split3(a, epi, variant, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_layernorm(const float* x, const float* g, const float* b, float* y, int32_t rows, int32_t D, float eps,
                      int32_t relu, void* stream) {
    thmr_engine* e = nullptr;
    if (!x || !g || !b || !y) return fail(e, THMR_ERR_INVALID, "null buffer");
    LAUNCH_OK(launch_layernorm(x, g, b, y, rows, D, eps, relu, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_vit_attention(const float* qkv, float* out, int32_t B, void* stream) {
    thmr_engine* e = nullptr;
    if (!qkv || !out) return fail(e, THMR_ERR_INVALID, "null buffer");
    LAUNCH_OK(launch_vit_attention(qkv, out, B, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_vit_attention_split3(const float* qkv, void* out_split, int32_t B, void* stream) {
    thmr_engine* e = nullptr;
    if (!qkv || !out_split || B <= 0) return fail(e, THMR_ERR_INVALID, "bad argument");
    LAUNCH_OK(launch_vit_attention_split3(qkv, out_split, B, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_vit_attention_b16(const float* qkv, void* out, int32_t B, int32_t out_split, int32_t qt, void* stream) {
    thmr_engine* e = nullptr;
    if (!qkv || !out || B <= 0) return fail(e, THMR_ERR_INVALID, "bad argument");
    LAUNCH_OK(launch_vit_attention_b16(qkv, out, B, out_split != 0, qt, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_vit_attention_variant(const float* qkv, float* out, int32_t B, int32_t variant, void* stream) {
    thmr_engine* e = nullptr;
    if (!qkv || !out) return fail(e, THMR_ERR_INVALID, "null buffer");
    LAUNCH_OK(launch_vit_attention_variant(qkv, out, B, variant, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_rot6d(const float* x, float* R, int32_t n, void* stream) {
    thmr_engine* e = nullptr;
    if (!x || !R || n < 1) return fail(e, THMR_ERR_INVALID, "bad argument");
    LAUNCH_OK(launch_rot6d(x, R, n, static_cast<hipStream_t>(stream)));
    return 0;
}

int thmr_op_aa_to_rotmat(const float* aa, float* R, int32_t n, void* stream) {
    thmr_engine* e = nullptr;
    if (!aa || !R || n < 1) return fail(e, THMR_ERR_INVALID, "bad argument");
    LAUNCH_OK(launch_aa_to_rotmat(aa, R, n, static_cast<hipStream_t>(stream)));
    return 0;
}

// ---- stand-alone SMPL model ----
struct thmr_smpl {
    float* mem = nullptr;
    int max_batch = 0;
    size_t o_vt, o_sd, o_pd, o_jr, o_w, o_j19, o_int, o_jt, o_jsd, o_dirs, o_A, o_pf, o_Jtr, o_vposed, o_rot, o_joints, o_xv, o_cnt, total;
};

int thmr_smpl_create(const thmr_smpl_desc* d, int32_t max_batch, int32_t device, thmr_smpl** out) {
    thmr_engine* e = nullptr;
    if (!d || !out || max_batch < 1) return fail(e, THMR_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (!d->v_template || !d->shapedirs || !d->posedirs || !d->J_regressor || !d->lbs_weights || !d->J19_regressor ||
        !d->parents || !d->extra_verts || !d->joint_map)
        return fail(e, THMR_ERR_INVALID, "thmr_smpl_desc has a null field");
    HIP_OK(hipSetDevice(device));
    thmr_smpl* m = new thmr_smpl();
    m->max_batch = max_batch;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align64(off + n); return o; };
    m->o_vt = take((size_t)NV * 3); m->o_sd = take((size_t)NV * 30); m->o_pd = take((size_t)NP * NV * 3);
    m->o_jr = take((size_t)NJ * NV); m->o_w = take((size_t)NV * NJ); m->o_j19 = take((size_t)19 * NV);
    m->o_int = take(128); m->o_jt = take(NJ * 3); m->o_jsd = take(NJ * 30); m->o_dirs = take((size_t)NV * 3 * THMR_LBS_KX);
    const size_t B = (size_t)max_batch;
    m->o_A = take(B * NJ * 12); m->o_pf = take(B * THMR_LBS_XF); m->o_Jtr = take(B * NJ * 3); m->o_rot = take(B * NJ * 9);
    m->o_vposed = take(B * NV * 3);
    m->o_joints = take(B * 132);
    m->o_xv = take(B * 63); m->o_cnt = take(B);
    m->total = off;
    if (hipMalloc(&m->mem, off * sizeof(float)) != hipSuccess) { delete m; return fail(e, THMR_ERR_NOMEM, "hipMalloc(smpl) failed"); }
    if (hipMemset(m->mem + m->o_cnt, 0, B * sizeof(float)) != hipSuccess) { thmr_smpl_destroy(m); return fail(e, THMR_ERR_HIP, "hipMemset(lbs counters) failed"); }
    const hipMemcpyKind k = d->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto cp = [&](size_t o, const void* src, size_t bytes) { return hipMemcpy(m->mem + o, src, bytes, k) == hipSuccess; };
    int32_t* ints = reinterpret_cast<int32_t*>(m->mem + m->o_int);
    bool ok = cp(m->o_vt, d->v_template, sizeof(float) * NV * 3) && cp(m->o_sd, d->shapedirs, sizeof(float) * NV * 30) &&
              cp(m->o_pd, d->posedirs, sizeof(float) * (size_t)NP * NV * 3) && cp(m->o_jr, d->J_regressor, sizeof(float) * NJ * NV) &&
              cp(m->o_w, d->lbs_weights, sizeof(float) * NV * NJ) && cp(m->o_j19, d->J19_regressor, sizeof(float) * 19 * NV) &&
              hipMemcpy(ints, d->parents, sizeof(int32_t) * 24, k) == hipSuccess &&
              hipMemcpy(ints + 24, d->extra_verts, sizeof(int32_t) * 21, k) == hipSuccess &&
              hipMemcpy(ints + 48, d->joint_map, sizeof(int32_t) * 25, k) == hipSuccess;
    const int32_t hips = d->update_hips ? 1 : 0;
    ok = ok && hipMemcpy(ints + 80, &hips, sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok || launch_lbs_jreg(m->mem + m->o_jr, m->mem + m->o_vt, m->mem + m->o_sd, m->mem + m->o_jt, m->mem + m->o_jsd, nullptr) != 0 ||
        launch_lbs_build_dirs(m->mem + m->o_sd, m->mem + m->o_pd, m->mem + m->o_dirs, nullptr) != 0 ||
        hipDeviceSynchronize() != hipSuccess) {
        thmr_smpl_destroy(m);
        return fail(e, THMR_ERR_HIP, "SMPL constant upload failed");
    }
    *out = m;
    return 0;
}

void thmr_smpl_destroy(thmr_smpl* m) {
    if (!m) return;
    if (m->mem) (void)hipFree(m->mem);
    delete m;
}

int thmr_smpl_forward(thmr_smpl* m, const float* pose, int32_t pose2rot, const float* betas, int32_t B, float* verts,
                      float* joints, void* stream) {
    thmr_engine* e = nullptr;
    if (!m || !pose || !betas || !verts) return fail(e, THMR_ERR_INVALID, "null argument");
    if (B < 1 || B > m->max_batch) return fail(e, THMR_ERR_INVALID, "batch outside [1, max_batch]");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* rot = pose;
    if (pose2rot) {
        LAUNCH_OK(launch_rodrigues(pose, m->mem + m->o_rot, B * NJ, st));
        rot = m->mem + m->o_rot;
    }
    const int32_t* ints = reinterpret_cast<const int32_t*>(m->mem + m->o_int);
    LAUNCH_OK(launch_lbs(rot, betas, nullptr, m->mem + m->o_jt, m->mem + m->o_jsd, ints, m->mem + m->o_vt, m->mem + m->o_dirs,
                         m->mem + m->o_w, m->mem + m->o_j19, ints + 24, ints + 48, ints + 80, m->mem + m->o_A, m->mem + m->o_pf,
                         m->mem + m->o_Jtr, m->mem + m->o_vposed, verts, joints ? joints : m->mem + m->o_joints, nullptr,
                         FOCAL / IMG, B, m->mem + m->o_xv, reinterpret_cast<unsigned*>(m->mem + m->o_cnt), st));
    return 0;
}

// ---- evaluation metrics (stateless) ----
int thmr_eval_pose(const float* pred_j, const float* gt_j, int32_t nj, int32_t gt_stride, const int32_t* kp, int32_t nkp,
                   int32_t pelvis_ind, int32_t pelvis_mode, const float* pred_v, const float* gt_v, int32_t nv, int32_t B,
                   float* mpjpe, float* re, float* pve, float* pelv, void* stream) {
    thmr_engine* e = nullptr;
    if (!pred_j || !gt_j || !kp || !mpjpe || !re || !pelv) return fail(e, THMR_ERR_INVALID, "null buffer");
    if (nkp < 1 || nkp > 64 || gt_stride < 3 || B < 1 || pelvis_ind < 0 || pelvis_ind >= nj || nj < 3)
        return fail(e, THMR_ERR_INVALID, "bad evaluator arguments (1 <= n_kp <= 64, gt_stride >= 3)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LAUNCH_OK(launch_eval_pose(pred_j, gt_j, nj, gt_stride, kp, nkp, pelvis_ind, pelvis_mode, mpjpe, re, pelv, B, st));
    if (pred_v && gt_v && pve) LAUNCH_OK(launch_eval_pve(pred_v, gt_v, pelv, nv, pve, B, st));
    return 0;
}

int thmr_regress_joints(const float* J, const float* verts, int32_t nj, int32_t nv, int32_t B, float* out, void* stream) {
    thmr_engine* e = nullptr;
    if (!J || !verts || !out || nj < 1 || nv < 1 || B < 1) return fail(e, THMR_ERR_INVALID, "bad argument");
    LAUNCH_OK(launch_regress_joints(J, verts, nj, nv, out, B, static_cast<hipStream_t>(stream)));
    return 0;
}

// ---- profiler ----
int thmr_prof_enable(thmr_engine* e, int32_t on) {
    if (!e) return fail(e, THMR_ERR_INVALID, "null engine");
    e->prof_on = on < 0 ? 0 : (on > 3 ? 1 : on);
    return 0;
}

int thmr_prof_collect(thmr_engine* e, thmr_prof_entry* entries, int32_t reset) {
    if (!e || !entries) return fail(e, THMR_ERR_INVALID, "null argument");
    for (int i = 0; i < THMR_PROF_NUM; ++i) entries[i] = thmr_prof_entry{0, 0, 0, 0};
    for (auto& r : e->prof) {
        HIP_OK(hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, r.e0, r.e1));
        entries[r.cls].ms += ms;
        entries[r.cls].flops += r.flops;
        entries[r.cls].bytes += r.bytes;
        entries[r.cls].launches += 1;
    }
    if (reset) { e->prof.clear(); e->ev_next = 0; }
    return 0;
}

}  // extern "C"
