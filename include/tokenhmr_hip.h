/*
 * tokenhmr_hip.h — C ABI of the MI355X-native TokenHMR inference engine (libtokenhmr_hip.so).
 *
 * The reference (saidwivedi/TokenHMR) has no FFI: its seam is the Python method
 *     model, cfg = load_tokenhmr(ckpt, model_cfg)          tokenhmr/lib/models/__init__.py:3-26
 *     out = model(batch)                                    tokenhmr/lib/models/tokenhmr.py:330-338
 * called from tokenhmr/eval.py:147, tokenhmr/demo.py:78, tokenhmr/track.py:39.
 * This header is what a ctypes binding for that seam binds; the Python
 * facade tokenhmr_amd/model.py is exactly such a binding.
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is a device (HBM) pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); the engine owns its weight arena and scratch arena.
 *   - all tensors are fp32, contiguous, row-major in the reference's own layouts.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     on it.  thmr_forward performs no allocation and no host synchronisation.
 *   - return value: 0 on success, negative thmr_status on failure; thmr_last_error() gives text.
 *   - one engine per GPU / per caller thread; not re-entrant.
 */
#ifndef TOKENHMR_HIP_H
#define TOKENHMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): thmr_smpl_desc.reserved became update_hips (a round-1 host that left it uninitialised would get the hip shift at
 * random); the resample tables / padding row / flag words of the weight arena are written by thmr_load_weights on the LOADING
 * engine (round 1: thmr_create on every engine; round 2: thmr_finalize_weights(0)) and validated — not written — by
 * thmr_finalize_weights(assume_all_loaded = 1).
 * 3 (round 3): thmr_set_vit_gemm / thmr_get_vit_gemm and the split3 operators added (no struct changed).
 * 4 (round 5): an engine is CREATED in mode 1 ("split3") — thmr_finalize_weights builds the split3 weight copies, thmr_forward runs the bf16
 *   matrix pipe from 3 crops on — and thmr_set_vit_gemm(0) is the opt-out to exact-fp32 MFMA (up to ABI 3 it was the other way round); no
 *   struct or signature changed.
 * 5 (round 6): thmr_config.reserved[0] became `flags` (THMR_CFG_*: create in the opt-out mode, create without co-residency-dependent kernels;
 *   a zeroed field = ABI 4's behaviour), thmr_mode_bytes added; the split3 weight copies are shared among the engines of one weight arena. */
#define THMR_ABI_VERSION 5

typedef enum {
    THMR_OK = 0,
    THMR_ERR_INVALID = -1,      /* bad argument / shape / missing tensor */
    THMR_ERR_HIP = -2,          /* a HIP runtime call failed */
    THMR_ERR_STATE = -3,        /* call order violated (e.g. forward before weights are finalised) */
    THMR_ERR_NOMEM = -4
} thmr_status;

typedef struct thmr_engine thmr_engine;

/* Architecture knobs.  Everything else (192 tokens, 1280 dim, 16x80 heads, 160x2048 tokens ...)
 * is fixed by the reference (vit.py:12-24, tokenhmr_release.yaml:65-81) and baked into kernels. */
typedef struct {
    int32_t abi_version;        /* must be THMR_ABI_VERSION */
    int32_t vit_depth;          /* 32 for ViT-H (vit.py:17); smaller only for tests */
    int32_t dec_depth;          /* 6 (tokenhmr_release.yaml:74) */
    int32_t max_batch;          /* crops per thmr_forward call the scratch arena is sized for */
    int32_t device;             /* HIP device ordinal */
    int32_t flags;              /* THMR_CFG_* below, 0 = defaults */
    int32_t reserved[2];        /* must be 0 */
} thmr_config;

/* thmr_config.flags */
enum {
    /* Create the engine in the exact-fp32 mode (as if thmr_set_vit_gemm(0) had been called before thmr_finalize_weights): finalize then
     * builds NO split3 weight copies and allocates no split3 activation buffers (3.8 GB + 0.5 GB at release depth and 64 crops, outside the
     * caller's arenas: thmr_mode_bytes).  thmr_set_vit_gemm(1) later builds them on demand. */
    THMR_CFG_VIT_GEMM_F32 = 1,
    /* No kernel that needs ALL its workgroups resident at once: the head runs as the launch chain instead of the persistent decoder kernel
     * (grid barrier) and the split3 GEMMs one workgroup per tile instead of the 256-workgroup stream with slab hand-over (the ViT bit-identical,
     * the head to fp32 summation order; a few per cent slower).  For a GPU SHARED WITH ANOTHER PROCESS: there the persistent kernels can starve each other until their bounded
     * waits (~0.5 s) run out — one batch of invalid outputs, an error from the next call, then this mode anyway (thmr_engine_status).  Engines
     * of ONE process are ordered by the library itself (one turn per forward-type call) and do not need the flag. */
    THMR_CFG_NO_PERSISTENT = 2
};

/* One named tensor of the reference checkpoint contract:
 *   'backbone.*' / 'smpl_head.*'   tokenhmr/lib/utils/misc.py:242-256 (load_pretrained)
 *   'decoder.decoder.*', 'quantizer.codebook'   tokenization/models/vanilla_pose_vqvae.py:299-301 */
typedef struct {
    const char* name;
    const void* data;           /* fp32, contiguous, reference layout */
    int64_t     numel;
    int32_t     on_device;      /* 0: host pointer, 1: device pointer */
    int32_t     reserved;
} thmr_tensor_desc;

/* SMPL constants (replaces smplx.SMPLLayer buffers + joint_regressor_extra,
 * tokenhmr/lib/models/smpl_wrapper.py:11-25).  Host or device pointers (all same side). */
typedef struct {
    const float*   v_template;      /* (6890,3) */
    const float*   shapedirs;       /* (6890,3,10) */
    const float*   posedirs;        /* (207,20670)  smplx layout */
    const float*   J_regressor;     /* (24,6890) */
    const float*   lbs_weights;     /* (6890,24) */
    const float*   J19_regressor;   /* (19,6890)  SMPL_to_J19.pkl */
    const int32_t* parents;         /* (24) kinematic tree, parents[0] = -1 */
    const int32_t* extra_verts;     /* (21) smplx vertex_ids['smplh'] */
    const int32_t* joint_map;       /* (25) smpl_wrapper.py:19-20 */
    int32_t        on_device;
    int32_t        update_hips;     /* SMPL(update_hips=...) smpl_wrapper.py:11,33-36: 1 = shift the two hip joints (mapped joints 9, 12);
                                     * MUST be 0 or 1 (it was `reserved` in ABI 1: initialise it) */
} thmr_smpl_desc;

/* Output buffers of one forward (tokenhmr.py:156-188).  Any pointer may be NULL (= not wanted). */
typedef struct {
    float*   pred_cam;              /* (B,3) */
    float*   rotmat;                /* (B,24,3,3): [:,0] = global_orient, [:,1:] = body_pose */
    float*   betas;                 /* (B,10) */
    float*   cls_logits_softmax;    /* (B,160,2048) */
    float*   pred_cam_t;            /* (B,3) */
    float*   focal_length;          /* (B,2) */
    float*   pred_keypoints_3d;     /* (B,44,3) */
    float*   pred_vertices;         /* (B,6890,3) */
    float*   pred_keypoints_2d;     /* (B,44,2) */
    int32_t* token_idx;             /* (B,160) argmax_k logits, lowest index on ties */
    /* optional taps for parity tests */
    float*   vit_features;          /* (B,192,1280) token-major last_norm output */
    float*   token_out;             /* (B,1024) */
    float*   cls_logits;            /* (B,160,2048) raw logits */
    float*   pose6d;                /* (B,144) */
} thmr_outputs;

/* kernel-class ids for the built-in HIP-event profiler (bench.py roofline leg) */
enum {
    THMR_PROF_GEMM_QKV = 0, THMR_PROF_GEMM_PROJ = 1, THMR_PROF_GEMM_FC1 = 2, THMR_PROF_GEMM_FC2 = 3,
    THMR_PROF_ATTN = 4, THMR_PROF_LN = 5, THMR_PROF_PATCH = 6, THMR_PROF_DEC_KV = 7,
    THMR_PROF_HEAD = 8, THMR_PROF_LBS = 9, THMR_PROF_NUM = 10
};
typedef struct {
    double  ms;         /* sum of HIP-event durations on the launch stream */
    double  flops;      /* algorithmic flops (2*MAC) of those launches */
    double  bytes;      /* algorithmic HBM bytes of those launches */
    int64_t launches;
} thmr_prof_entry;

int         thmr_abi_version(void);
const char* thmr_build_info(void);

/* Bytes the engine needs; lets the caller allocate the arenas itself (e.g. as torch tensors so
 * that torch.distributed/RCCL can broadcast the weight arena). */
int thmr_arena_bytes(const thmr_config* cfg, size_t* weight_bytes, size_t* scratch_bytes);

/* Device memory the engine allocates ITSELF, outside the two arenas, for `vit_gemm_mode` (0 / 1, see thmr_set_vit_gemm): the split3 copies
 * of the ViT weights (shared by all engines of this process that were created on the same weight_arena_dev), the split3 activation operands
 * of max_batch crops, the hand-over workspace of the persistent GEMM.  All 0 for mode 0 and for max_batch < 3.  No GPU needed. */
int thmr_mode_bytes(const thmr_config* cfg, int32_t vit_gemm_mode, size_t* split_weight_bytes, size_t* split_act_bytes, size_t* workspace_bytes);

/* Enumerate the checkpoint contract the engine expects (name + element count), index = 0..count-1.
 * Returns the number of tensors; name/numel may be NULL.  No GPU needed. */
int thmr_spec(const thmr_config* cfg, int32_t index, const char** name, int64_t* numel);

/* weight_arena_dev / scratch_arena_dev may be NULL: the engine then hipMallocs and owns them. */
int thmr_create(const thmr_config* cfg, void* weight_arena_dev, void* scratch_arena_dev, thmr_engine** out);
void thmr_destroy(thmr_engine* e);
const char* thmr_last_error(const thmr_engine* e);   /* e may be NULL: last global error */

/* Weight ingest — replaces load_pretrained/prepare_statedict (misc.py:215-256) and
 * DecodeTokens.load_weights (vanilla_pose_vqvae.py:299-301).  May be called several times with
 * disjoint subsets; unknown names are an error (strict=True semantics). */
int thmr_load_weights(thmr_engine* e, const thmr_tensor_desc* tensors, size_t n, void* stream);
int thmr_load_smpl(thmr_engine* e, const thmr_smpl_desc* smpl, void* stream);
/* Checks every required tensor arrived and builds derived layouts (conv/codebook repacks,
 * J_template/J_shapedirs).  Also the hook to call after an external broadcast into the arena:
 * assume_all_loaded = 1 skips the per-tensor bookkeeping (this engine loaded nothing itself) and instead requires the loader's
 * magic word in the arena — THMR_ERR_STATE if the arena was never filled by a thmr_load_weights (e.g. broadcast too early).
 * The arena is complete for a broadcast as soon as the root's thmr_load_weights + thmr_load_smpl have returned; the root
 * may finalize before or after broadcasting. */
int thmr_finalize_weights(thmr_engine* e, int32_t assume_all_loaded, void* stream);
int thmr_weight_arena(thmr_engine* e, void** ptr_dev, size_t* bytes);

/* The hot path: TokenHMR.forward (tokenhmr.py:330-338 -> forward_step :135-188).
 * img_dev: (B,3,256,256) fp32 normalised RGB crops.  1 <= B <= max_batch. */
int thmr_forward(thmr_engine* e, const float* img_dev, int32_t B, const thmr_outputs* out, void* stream);

/* Device-side health of the engine: synchronises `stream` and returns THMR_ERR_HIP if a kernel of this engine reported an
 * error asynchronously — the bounded grid barrier of the persistent decoder kernel, or a bounded hand-over wait of the persistent split3 GEMM,
 * timed out instead of hanging the GPU.  thmr_forward itself never synchronises; it does look at host-mapped copies of the same error words
 * on entry, so a timeout is also reported by the NEXT forward-type call.  Either way the error is returned once: the engine drains the
 * device, resets the barrier words / hand-over workspace and switches to the launch chain / per-tile kernel (no co-residency needed), so
 * re-submitting the batch works. */
int thmr_engine_status(thmr_engine* e, void* stream);

/* Diagnostics: with THMR_DEC_TIMELINE=1 in the environment at thmr_finalize_weights, workgroup 0 of the persistent decoder kernel
 * stamps the 100 MHz wall clock after every step and barrier of the last call; this copies up to max_stamps (<= 240) of them. */
int thmr_debug_decoder_timeline(thmr_engine* e, uint64_t* stamps_host, int32_t max_stamps, void* stream);

/* Sub-paths (configs 2 of BASELINE.json and unit parity). */
int thmr_vit_forward(thmr_engine* e, const float* img_dev, int32_t B, float* feats_dev /*(B,192,1280)*/, void* stream);
int thmr_head_forward(thmr_engine* e, const float* ctx_dev /*(B,192,1280)*/, int32_t B, const thmr_outputs* out, void* stream);
/* SMPL forward (smpl_wrapper.py:27-41 over smplx lbs) + projection (geometry.py:86-124) */
int thmr_lbs_forward(thmr_engine* e, const float* rotmat_dev /*(B,24,3,3)*/, const float* betas_dev /*(B,10)*/,
                     const float* cam_dev /*(B,3) or NULL*/, int32_t B, float* verts_dev, float* joints_dev,
                     float* cam_t_dev, float* kp2d_dev, void* stream);
/* QuantizeEMAReset.quantize (tokenization/models/quantize_cnn.py:80-86): argmin_k ||x-c_k||^2 in the
 * reference's expanded form; x (rows,256) -> idx (rows) int32; optional dist (rows,2048). */
int thmr_vq_argmin(thmr_engine* e, const float* x_dev, int32_t rows, int32_t* idx_dev, float* dist_dev, void* stream);

/* Tokenizer encode path: EncodeTokens.forward (tokenization/models/vanilla_pose_vqvae.py:334-342)
 * = PoseSPEncoderV1 (:66-111) -> QuantizeEMAReset.preprocess/quantize (quantize_cnn.py:74-86).
 * Needs the optional 'encoder.encoder.*' tensors of tokenizer.pth (loaded through thmr_load_weights; all or none).
 * pose_dev (B,21,6) rot6d body pose -> idx_dev (B,160) int32 code indices; latent_dev (B,160,256) optional. */
int thmr_encode_tokens(thmr_engine* e, const float* pose_dev, int32_t B, int32_t* idx_dev, float* latent_dev, void* stream);
/* DecodeTokens.forward (vanilla_pose_vqvae.py:294-297): probs (B,160,2048) @ codebook -> PoseSPDecoderV1 -> pose6d (B,21,6).
 * One-hot probs give the hard decode of code indices (QuantizeEMAReset.dequantize, quantize_cnn.py:88-90). */
int thmr_vq_decode(thmr_engine* e, const float* probs_dev, int32_t B, float* pose6d_dev, void* stream);

/* Stateless operator entry points (unit parity of individual kernels; no engine needed). */
/* C[M,N] = epilogue(A[M,K] . W[N,K]^T) in exact fp32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32; csrc/gemm_f32.hip).  epi: 0 none, 1 +bias,
 * 2 +bias gelu(erf), 3 +bias relu, 4 resid + (acc+bias), 5 (+bias)*qscale on cols < qcols, 6 +bias +pos_embed (patch embed).  K % 32 == 0;
 * lda/ldc in elements, multiples of 4, < 2^22.  variant (the ids the ENGINE uses; the A/B-only ids are listed in tokenhmr_amd/ops.py):
 *   -1 = tile picked by the cost model over the 128x128 / 128x160 / 128x96 / 64x64 LDS-DMA tiles (7 / 8 / 10 / 9 force one);
 *   2 = skinny (M <= 64); 100 + j = 64x64 ring kernel, 4-deep, split-K 2^j (few crops); 11 = tiny-M kernel (32x32 tiles, K split over the
 *   8 waves; K % 256 == 0; epilogues 0-5; the VQ decoder up to six crops); 120 = small-M kernel on 16x16x4 tiles (qkv at one and two crops);
 *   200 + t / 400 + t = split-K 2 / 4 on big tile t.  Every variant sums K in the same order except split-K, the tiny-M kernel and 120. */
int thmr_op_gemm(const float* A_dev, int64_t lda, const float* W_dev, const float* bias_dev, const float* resid_dev,
                 float* C_dev, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, float qscale, int32_t qcols,
                 int32_t variant, void* stream);
/* fp32 GEMM on the bf16 matrix pipe (csrc/gemm_split16.hip) — what the engine's DEFAULT mode (thmr_set_vit_gemm 1) runs for the ViT GEMMs;
 * also an operator of its own, measured beside thmr_op_gemm.  Every fp32 operand is carried as three bf16 pieces h + m + l (x == h + m + l
 * up to 2^-24 |x|) in the "split3" layout [rows][K/8][3][8] bf16 (row stride 6 * ld bytes); the product keeps the six piece pairs down to
 * 2^-16 of |a b| (what is dropped is below one fp32 rounding of the product), accumulation is fp32 in the MFMA.
 * thmr_op_split3 converts (K % 8 == 0, ld_dst % 8 == 0, ld_dst >= K, ld_src % 4 == 0).  thmr_op_gemm_split3: A / W split3 with row strides
 * lda / ldw in fp32-equivalents (multiples of 8), K % 32 == 0; bias / resid / C fp32; epi 0, 1, 2, 4, 5, 6 as thmr_op_gemm (6: per-tile variants, N % 4 == 0).  variant:
 *   -1 = the engine's rule; 0 = 128x256 tile, 8 waves; 2 = 128x128, 4 waves; 5 = the 128x256 grid with its ragged last round as 128x128
 *         half tiles (only shapes whose tile count leaves at most half a round: fc1 of a 64-crop batch; else an error) — all bit-identical;
 *   202 / 204 = split-K 2 / 4 on the big tiles (the engine's 5 ... 31 crops use 2, 3 and 4 crops 4);
 *   300 = 256 PERSISTENT workgroups over a tile stream (M % 128 == 0, N % 256 == 0, at least 256 tiles, a 256-CU device; a ragged last round
 *         is split along K with the accumulators handed from one workgroup to the next through memory — bit-identical to 0 / 2; the
 *         engine's fc2 at 32 crops and more);
 *   + 1000 (1000, 1002, 1202, 1204, 1300; epi 0 / 4): A is a ROW-BLOCKED split3 operand [rows / 32][K / 8][3][32][8] (rows padded to 32;
 *         chunk (r, k-group, piece) at (r / 32) K 192 + (k-group 3 + piece) 512 + (r % 32) 16 bytes) — the form the engine's fc1 hands fc2.
 * (Ids of kernels that lost their A/B exist in the experiments build only: tokenhmr_amd/ops.py lists them.) */
int thmr_op_split3(const float* src_dev, int64_t ld_src, void* dst_dev, int64_t ld_dst, int64_t rows, int32_t K, void* stream);
int thmr_op_gemm_split3(const void* A_split_dev, int64_t lda, const void* W_split_dev, int64_t ldw, const float* bias_dev,
                        const float* resid_dev, float* C_dev, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi,
                        float qscale, int32_t qcols, int32_t variant, void* stream);
/* the same product with the epilogue's result written as a split3 operand (the next GEMM's A; row stride 6 * ldcs bytes, N % 8 == 0,
 * ldcs % 8 == 0) instead of fp32: bit-identical to thmr_op_split3 of thmr_op_gemm_split3's output.  epi 0, 1, 2, 5; variant -1, 0, 2, 5;
 * 302 = the persistent kernel (epi 0 / 2); + 1000 = the result in the row-blocked form (Cs holds ceil(M / 32) * 32 rows). */
int thmr_op_gemm_split3_out_split3(const void* A_split_dev, int64_t lda, const void* W_split_dev, int64_t ldw, const float* bias_dev,
                                   void* C_split_dev, int64_t ldcs, int32_t M, int32_t N, int32_t K, int32_t epi, float qscale,
                                   int32_t qcols, int32_t variant, void* stream);
int thmr_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                      int32_t rows, int32_t D, float eps, int32_t relu, void* stream);
/* ViT global attention over 192 tokens, 16 heads x 80 (vit.py:113-122); qkv (B,192,3840) with q pre-scaled. */
int thmr_op_vit_attention(const float* qkv_dev, float* out_dev /*(B,192,1280)*/, int32_t B, void* stream);
/* the same with the kernel forced: 0 = the batch-size rule of thmr_op_vit_attention; 1 = three 64-query workgroups per (crop, head),
 * 3 = one 192-query workgroup, 5 = persistent workgroups (1 / 3 / 5 are bit-identical); 6 = key-split (16 queries per workgroup,
 * the 192 keys split over its 4 waves, partial softmaxes merged: what the engine uses up to six crops; equal to fp32 rounding;
 * 61 / 62 / 63 = the same with 16 / 32 / 48 queries per workgroup, bit-identical to each other) */
int thmr_op_vit_attention_variant(const float* qkv_dev, float* out_dev, int32_t B, int32_t variant, void* stream);
/* thmr_op_vit_attention with the output written as a split3 operand [B*192][1280/8][3][8] bf16 (see thmr_op_split3): for B >= 3
 * bit-identical to thmr_op_split3 of thmr_op_vit_attention's output; for B = 1 and 2 this operator runs the key-split kernel (another
 * order of the key sum), so there it is bit-identical to thmr_op_split3 of thmr_op_vit_attention_variant(..., 6, ...)'s output.
 * (The engine's split3 mode ran this up to round 4; it now runs thmr_op_vit_attention_b16 with out_split = 1.) */
int thmr_op_vit_attention_split3(const float* qkv_dev, void* out_split_dev, int32_t B, void* stream);
/* The same attention (vit.py:113-122) on the bf16 matrix pipe: q, k, v and the un-normalised probabilities enter v_mfma_f32_16x16x32_bf16 as
 * three bf16 pieces each, six products per pair, fp32 accumulate (csrc/attention_b16.hip) — the split3 mode's arithmetic applied to
 * q k^T and p v; the three 64-key blocks are combined with a running row maximum.  fp32-grade (error against fp64 at or below the fp32-MFMA
 * kernel's, tests/test_gpu_ops.py::test_vit_attention_b16), NOT bit-identical to thmr_op_vit_attention.  What the engine's split3 mode runs.
 * out_split = 0: out_dev is fp32 (B,192,1280); 1: the split3 operand [B*192][1280/8][3][8] bf16.  qt = 0: batch-size rule; 1: three
 * 64-query workgroups per (crop, head); 3: one workgroup of 192 queries (bit-identical to each other and for any B). */
int thmr_op_vit_attention_b16(const float* qkv_dev, void* out_dev, int32_t B, int32_t out_split, int32_t qt, void* stream);
/* rot6d_to_rotmat (geometry.py:64-84): (n,6) -> (n,3,3) */
int thmr_op_rot6d(const float* x_dev, float* R_dev, int32_t n, void* stream);
/* aa_to_rotmat (geometry.py:5-44; axis-angle -> quaternion -> rotation matrix, the reference's in-tree "Rodrigues" used for
 * ground-truth poses at tokenhmr.py:235,260,357): (n,3) -> (n,3,3) */
int thmr_op_aa_to_rotmat(const float* aa_dev, float* R_dev, int32_t n, void* stream);

/* Stand-alone SMPL model: the GT-side meshes the reference computes per sample on the CPU with
 * smplx.SMPL(gender) inside dataset workers (tokenhmr/lib/datasets/image_dataset.py:151-164,254-270, emdb_dataset.py:184-199)
 * reuse the LBS kernels with their own (male / female) constants.
 *   pose2rot = 0: pose_dev is (B,24,3,3) rotation matrices; 1: (B,72) axis-angle, converted by smplx batch_rodrigues. */
typedef struct thmr_smpl thmr_smpl;
int  thmr_smpl_create(const thmr_smpl_desc* desc, int32_t max_batch, int32_t device, thmr_smpl** out);
void thmr_smpl_destroy(thmr_smpl* m);
int  thmr_smpl_forward(thmr_smpl* m, const float* pose_dev, int32_t pose2rot, const float* betas_dev, int32_t B,
                       float* verts_dev /*(B,6890,3)*/, float* joints_dev /*(B,44,3) or NULL*/, void* stream);

/* Evaluation metrics right after the hot path — stateless, all buffers device-side.
 * Replaces compute_similarity_transform / eval_pose and the arithmetic of Evaluator.__call__
 * (tokenhmr/lib/utils/pose_utils.py:61-143, :201-275): pelvis alignment, MPJPE, PA-MPJPE (3x3 SVD Procrustes), PVE, in mm.
 *   pred_joints (B,n_joints,3); gt_joints (B,n_joints,gt_stride) (gt_stride = 4 for batch['keypoints_3d'] with its confidence column)
 *   pelvis_mode 0: joint pelvis_ind; 1: (joint1 + joint2)/2 (EMDB branch).  verts may be NULL (no PVE).
 *   pelvis_scratch (B,6) receives [pred_pelvis | gt_pelvis]. */
int thmr_eval_pose(const float* pred_joints_dev, const float* gt_joints_dev, int32_t n_joints, int32_t gt_stride,
                   const int32_t* kp_list_dev, int32_t n_kp, int32_t pelvis_ind, int32_t pelvis_mode,
                   const float* pred_verts_dev, const float* gt_verts_dev, int32_t n_verts, int32_t B,
                   float* mpjpe_mm_dev, float* re_mm_dev, float* pve_mm_dev, float* pelvis_scratch_dev, void* stream);
/* joints = J (n_joints,n_verts) @ verts (B,n_verts,3): J_regressor_24_SMPL of the EMDB branch (pose_utils.py:212,219) */
int thmr_regress_joints(const float* J_dev, const float* verts_dev, int32_t n_joints, int32_t n_verts, int32_t B,
                        float* out_dev, void* stream);

/* Crop preprocessing right before the hot path: decoded uint8 frame on the device + one affine per crop ->
 * normalised (n,3,patch,patch) fp32 = batch['img'].  Replaces, per crop, the reference's CPU-side
 *   skimage.filters.gaussian anti-alias of the whole frame   tokenhmr/lib/datasets/vitdet_dataset.py:62-68, utils.py:583-587
 *   cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT)            tokenhmr/lib/datasets/utils.py:351-356 (generate_image_patch_cv2)
 *   [:, :, ::-1], HWC->CHW float32, (x - mean)/std           vitdet_dataset.py:75-80, utils.py:599-617
 * with OpenCV's fixed-point bilinear and scipy's correlate1d arithmetic reproduced exactly (csrc/crop.hip).  The affine itself
 * (gen_trans_from_patch_cv + cv2.getAffineTransform, utils.py:81-128) is 3 points of host arithmetic and stays with the caller
 * (tokenhmr_amd/preprocess.py mirrors ViTDetDataset).
 *   M       forward 2x3 matrix exactly as passed to cv2.warpAffine (src -> dst), row-major
 *   sigma   gaussian sigma of the anti-alias blur applied before the warp, 0 = none; truncate: 4.0 (vitdet) / 3.0 (get_example)
 *   frame   (H, W, 3) uint8, row_stride bytes per row; swap_rb = 1 flips the channel order (BGR frame -> RGB planes)
 *   mean/std  in 0..255 units, indexed by OUTPUT channel.
 * The handle owns grow-only device scratch for the blurred regions; growing it synchronises the stream. */
typedef struct thmr_crop_desc {
    double M[6];
    double sigma;
    double truncate;
} thmr_crop_desc;
typedef struct thmr_cropper thmr_cropper;
int  thmr_cropper_create(int32_t device, thmr_cropper** out);
void thmr_cropper_destroy(thmr_cropper* c);
const char* thmr_cropper_last_error(const thmr_cropper* c);
int  thmr_cropper_run(thmr_cropper* c, const uint8_t* frame_dev, int32_t H, int32_t W, int64_t row_stride,
                      const thmr_crop_desc* crops_host, int32_t n, int32_t patch, int32_t swap_rb, const float* mean_host,
                      const float* std_host, float* out_dev, void* stream);

/* ---- data-parallel collectives for hosts without torch.distributed ----
 * The reference has no collective on this path (inference is single-device, tokenhmr/eval.py:52-54).  Crops shard with NO data-path
 * collective; two collectives surround the path: ONE broadcast of the packed weight arena at start-up (only rank `root` read the
 * checkpoint; follow it with thmr_finalize_weights(assume_all_loaded = 1) on the receivers) and ONE all-gather per batch of the packed
 * per-crop records.  `nccl_comm` is an ncclComm_t the caller created (ncclCommInitRank) with the RCCL already loaded in the process;
 * the library resolves ncclBroadcast / ncclAllGather from that copy at first use (it does not link its own).  Asynchronous on `stream`.
 *   record = [pred_vertices 20670 | pred_keypoints_3d 132 | pred_keypoints_2d 88 | rotmat 216 | betas 10 | pred_cam 3 | pred_cam_t 3 |
 *             token_idx 160 (int32 bits)] = THMR_RECORD_WORDS 32-bit words per crop (85,128 B). */
#define THMR_RECORD_WORDS 21282
int thmr_pack_records(const thmr_outputs* out, int32_t B, float* rec_dev /*(B, THMR_RECORD_WORDS)*/, void* stream);
int thmr_bcast_weights(thmr_engine* e, void* nccl_comm, int32_t root, void* stream);
/* every rank contributes `rows` records (pad to the largest shard); recv_dev is (world * rows, THMR_RECORD_WORDS) in rank order */
int thmr_allgather_records(void* nccl_comm, const float* rec_dev, int32_t rows, float* recv_dev, void* stream);
const char* thmr_collective_last_error(void);

/* Built-in profiler: HIP events recorded on the launch stream around each kernel class.
 * on = 0 off, 1 every class, 2 only the four ViT GEMM classes, 3 only fc1 (the dominant kernel), sampled.  An event pair costs ~2-3 us
 * of stream time: 128 pairs per call (on = 2) were 0.75 % of a B = 64 step and 20 % of a B = 1 call (round 3: the facade call
 * without events was FASTER than the timed loop), which is why bench.py times with on = 3, which samples every 4th fc1 launch (8 pairs per call; all 32 launches have one shape). */
/* How the four ViT GEMMs (qkv / proj / fc1 / fc2: 97 % of the path's arithmetic), the ViT attention and the decoder's to_kv GEMM multiply.
 * Both modes are fp32 in, fp32 accumulate, fp32 out.
 *   1 (DEFAULT, ABI 4): "split3" — each fp32 operand as three bf16 pieces, six bf16 MFMA products per element pair, fp32 accumulation
 *      (csrc/gemm_split16.hip, attention_b16.hip; v_mfma_f32_16x16x32_bf16).  fp32-GRADE, not bitwise fp32: the measured error against an
 *      fp64 product is no larger than the exact-fp32 kernel's (tests/test_gpu_ops.py::test_gemm_split3), at ~1.55x its rate end to end.
 *      It is what bench.py's `value`, the facade (tokenhmr_amd.model.load_tokenhmr) and every parity claim of the default path refer to:
 *      0 of 40,960 pose-token indices differ from the reference's on the four 64-crop fixtures, joints / vertices within
 *      max(0.1 mm, 2 x the reference's own fp32-vs-fp64 distance) (tests/test_gpu_model.py).  Applies to calls of at least 3 crops (one
 *      and two crops run the exact-fp32 kernels in either mode).  Four ranges, a crop's result is batch-independent within each: 3 and 4
 *      crops split the K sums of proj and fc2 four ways, 5 ... 15 two ways, 16 ... 31 only fc2's (two ways), 32 and more neither.
 *      LayerNorm, the epilogues and the rest of the head are fp32 arithmetic in both modes.
 *   0 (opt-out): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) everywhere — bitwise an fmaf chain.
 * Mode 1 needs the engine-owned split3 copy of the ViT weights (1.5x their fp32 bytes) and the operand buffers (+ the partial-sum planes
 * of its split-K ranges): built by thmr_finalize_weights while the mode is on (i.e. by default) or by thmr_set_vit_gemm(1) on finalized
 * weights, kept until thmr_destroy (setting 0 does not free them).  An engine created with max_batch < 3 never runs the mode and builds
 * nothing.  Like thmr_forward the switch allocates nothing per call, so a call in either mode can be captured in a hipGraph.
 * Returns 0 / negative; thmr_get_vit_gemm returns the mode. */
int thmr_set_vit_gemm(thmr_engine* e, int32_t mode, void* stream);
int thmr_get_vit_gemm(thmr_engine* e);
int thmr_prof_enable(thmr_engine* e, int32_t on);
int thmr_prof_collect(thmr_engine* e, thmr_prof_entry* entries /*[THMR_PROF_NUM]*/, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* TOKENHMR_HIP_H */
