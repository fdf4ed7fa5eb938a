/* mmr.h -- C ABI of the MI355X-native CLIP encode + cosine top-k hot path.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its scripts
 * call a Python object API (`clip.load(...)` -> model.encode_image / encode_text / model(...),
 * reference code/test_clip.py:6-16, code/search_image.py:132,156,335) and plain tensor
 * expressions for scoring/ranking (code/search_image.py:107, code/utils.py:17).  This header is
 * what a Python shim binds with ctypes to replace that path; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) unless named *_host;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no allocation, no synchronisation inside: the caller owns outputs and workspace, calls
 *     are asynchronous on `stream` and are hipGraph-capturable;
 *   - return 0 on success or a negative errno-style code; mmr_last_error() gives the text
 *     (thread-local).  Arguments are validated on the host BEFORE any launch.
 */
#ifndef MMR_H
#define MMR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MMR_F32 = 0, MMR_BF16 = 1 } mmr_dtype;

enum {
    MMR_OK = 0,
    MMR_EIO = -5,      /* a HIP runtime call failed */
    MMR_EINVAL = -22,  /* bad argument (shape, dtype, alignment, null pointer) */
    MMR_ENOSPC = -28,  /* workspace smaller than *_workspace_bytes() */
    MMR_ENOTSUP = -95  /* shape outside what the kernels are built for */
};

const char *mmr_last_error(void);
int mmr_version(void);

/* ------------------------------------------------------------------------------------------
 * Similarity / ranking  (replaces reference code/search_image.py:107 `100. * F @ r.t()`,
 * code/utils.py:17 `output.topk(k,1,True,True)`, code/search_image.py:133,157 `f /= f.norm()`).
 * ---------------------------------------------------------------------------------------- */

/* Bytes of scratch mmr_cosine_topk needs for this problem size. */
size_t mmr_search_workspace_bytes(int64_t N, int E, int Q, int k);

/* Top-k gallery rows per query by dot product, ordered by (-dot, +row index).
 *   q[Q,E], gallery[N,E] row-major, dtype fp32 or bf16 (both operands the same dtype).
 *   idx[Q,k] int32 row ids (-1 past N), score[Q,k] = (float)(dot64*scale),
 *   dot64[Q,k] (nullable) the exact fp64 dot products used for ranking,
 *   status[Q] (nullable): 0 = MFMA scan + certified exact re-rank, 1 = exhaustive exact path.
 *   scale must be > 0.
 *   gallery_norm_bound sizes the margin of the fast path's exactness certificate
 *   (8e-5 * ||q|| * bound, above the worst-case fp32 MFMA accumulation error for E <= 1024):
 *     <= 0 : measured by this call from the gallery itself (one extra streaming pass) -- always sound;
 *     >  0 : PRECONDITION: no gallery row has a larger L2 norm (1.0 for a normalised gallery).  An
 *            understated bound can let the certificate pass wrongly, i.e. return a top-k that is not exact;
 *            an overstated one only sends more queries down the exhaustive path.
 * Ranking is on fp64 dot products accumulated in the fixed order documented in
 * oracle/search_ref.c, so indices are bit-reproducible against the CPU oracle. */
int mmr_cosine_topk(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, int k,
                    float scale, float gallery_norm_bound, int32_t *idx, float *score, double *dot64,
                    int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* bound_out[0] (device float) = the largest row L2 norm of gallery[N,E] (0 for N = 0), rounded up.
 * One HBM-bound pass; an index that is searched many times measures once and hands the scalar to
 * mmr_cosine_topk_ex. */
int mmr_gallery_norm_bound(const void *gallery, mmr_dtype dtype, int64_t N, int E, float *bound_out, void *stream);

/* mmr_cosine_topk with the norm bound read on the device: the margin uses
 * max(gallery_norm_bound, *gallery_norm_bound_dev) (either may be absent: <= 0 / NULL; both absent =
 * measured in the call), so a caller-supplied number can widen the margin but never shrink it below
 * the measured one.  No host read of the scalar: the call stays asynchronous and graph-capturable. */
int mmr_cosine_topk_ex(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, int k,
                       float scale, float gallery_norm_bound, const float *gallery_norm_bound_dev, int32_t *idx,
                       float *score, double *dot64, int32_t *status, void *workspace, size_t workspace_bytes,
                       void *stream);

/* fp32 galleries that are searched many times: split the gallery ONCE into hi = bf16(x) and lo = bf16(x - hi) (two bf16
 * arrays [N,E], together the bytes of the fp32 gallery), then search with mmr_cosine_topk_split.  That search runs in tiers:
 *   1. the bf16 scan over `hi` alone with bf16-rounded queries (half the gallery bytes, one MFMA product), certified with a
 *      margin that adds the measured rounding residuals |q - bf16(q)| * max|g| + |q| * max_row|g - hi| and 32 candidate tiles;
 *   2. only for queries tier 1 leaves open: the three-product scan q_hi.g_hi + q_lo.g_hi + q_hi.g_lo over hi and lo (the scan
 *      mmr_cosine_topk runs on fp32 rows, minus its per-tile split) with the 8e-5 |q||g| margin;
 *   3. the exhaustive fp64 path for what is still open.
 * Every tier ranks on exact fp64 re-scores of the fp32 rows, so idx / score / dot64 are IDENTICAL to
 * mmr_cosine_topk_ex(..., MMR_F32, ...); status is 0 for tiers 1-2 and 1 for tier 3.
 * mmr_gallery_split_bf16 also writes max_row ||g - hi||_2 to *resid_bound_out (device float, may be NULL); pass that pointer as
 * split_resid_bound_dev (NULL = the worst case 2^-8 * norm bound is assumed).  The split arrays and the bound must come from
 * mmr_gallery_split_bf16 of the SAME gallery contents.  Workspace: mmr_search_workspace_bytes. */
int mmr_gallery_split_bf16(const float *gallery, int64_t N, int E, void *hi, void *lo, float *resid_bound_out, void *stream);
int mmr_cosine_topk_split(const void *q, const void *gallery, const void *gallery_hi, const void *gallery_lo,
                          const float *split_resid_bound_dev, int Q, int64_t N, int E, int k, float scale,
                          float gallery_norm_bound, const float *gallery_norm_bound_dev, int32_t *idx, float *score,
                          double *dot64, int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* out[Q,N] (fp32) = (float)(dot64 * scale): the materialised score matrix for small N. */
int mmr_similarity(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, float scale,
                   float *out, void *stream);

/* In-place row-wise x /= ||x||_2 (no epsilon, like the reference). */
int mmr_l2norm_rows(void *x, mmr_dtype dtype, int64_t rows, int E, void *stream);

/* Merge `parts` per-shard lists [parts,Q,k] (global int64 ids, fp64 dots; id < 0 = empty slot)
 * into one [Q,k] list with the same ordering.  Used after the RCCL all-gather. */
int mmr_topk_merge(const int64_t *idx_parts, const double *dot_parts, int parts, int Q, int k, float scale,
                   int64_t *idx, float *score, double *dot64, void *stream);

/* The same exchange with ONE message per rank: mmr_topk_pack writes packed[Q,k,2] int64 = (local id + row_offset or -1,
 * fp64 dot bits) from mmr_cosine_topk's idx/dot64; after the all-gather of those messages ([parts,Q,k,2])
 * mmr_topk_merge_packed ranks them.  Two launches around the collective instead of six tensor ops. */
int mmr_topk_pack(const int32_t *idx, const double *dot64, int Q, int k, int64_t row_offset, int64_t *packed, void *stream);
int mmr_topk_merge_packed(const int64_t *packed_parts, int parts, int Q, int k, float scale, int64_t *idx, float *score,
                          double *dot64, void *stream);

/* Multi-GPU leg for hosts without torch.distributed: one process per GPU; rank r searches its
 * gallery rows with mmr_cosine_topk(_ex), packs its list with mmr_topk_pack(row_offset = first global row of the shard),
 * then ONE RCCL all-gather over xGMI of the packed messages (mmr_allgather_topk_packed) and mmr_topk_merge_packed:
 *
 *     mmr_cosine_topk_ex(q, shard, ..., idx32, score, dot64, ...);          // local top-k, int32 local ids
 *     mmr_topk_pack(idx32, dot64, Q, k, row_offset, packed, stream);        // [Q,k,2] int64 = (global id | -1, dot bits)
 *     mmr_allgather_topk_packed(comm, packed, Q, k, parts, stream);         // [world,Q,k,2]   -- the collective
 *     mmr_topk_merge_packed(parts, world, Q, k, scale, idx64, score, dot64, stream);
 *
 * librccl is bound at run time (dlopen; MMR_RCCL_LIB overrides the name), so a single-GPU user needs no RCCL.  The Python
 * package performs the SAME exchange (one message of 2*Q*k int64 per rank) through torch.distributed's "nccl" (= RCCL)
 * backend (search.ShardedGalleryIndex.search_async). */
typedef struct mmr_comm mmr_comm;
#define MMR_COMM_ID_BYTES 128
/* rank 0: fill MMR_COMM_ID_BYTES bytes of HOST memory; the host program hands them to every rank (file, socket, MPI ...) */
int mmr_comm_unique_id(void *id_host);
/* collective over all ranks; binds the communicator to the calling thread's current HIP device */
int mmr_comm_init(int rank, int world, const void *id_host, mmr_comm **out);
void mmr_comm_destroy(mmr_comm *c);
/* packed_local [Q,k,2] int64 (device, mmr_topk_pack layout) -> packed_parts [world,Q,k,2] (device), asynchronous on
 * `stream`: ONE ncclAllGather.  Feed the result to mmr_topk_merge_packed(parts = world). */
int mmr_allgather_topk_packed(mmr_comm *c, const int64_t *packed_local, int Q, int k, int64_t *packed_parts, void *stream);
/* Unpacked form (kept for callers that hold separate arrays): idx_local / dot_local [Q,k] (device, GLOBAL int64 ids) ->
 * idx_parts / dot_parts [world,Q,k]; the two arrays travel as one grouped pair of all-gathers.  -> mmr_topk_merge. */
int mmr_allgather_topk(mmr_comm *c, const int64_t *idx_local, const double *dot_local, int Q, int k,
                       int64_t *idx_parts, double *dot_parts, void *stream);

/* Tip-Adapter logits, fused (replaces reference code/main_custom.py:111,124-127 and
 * code/utils.py:182-186):  tip = 100*F@W + alpha * (exp(-(beta - beta*(F@Kc))) @ V * 10).
 *   features[N,E]; clip_weights_t[C,E] = W^T; cache_keys_t[S,E] = Kc^T (all `dtype`);
 *   cache_values[S,C] fp32; tip_logits[N,C] fp32; clip_logits[N,C] fp32 (nullable) = 100*F@W.  C <= 64. */
int mmr_tip_adapter_logits(const void *features, const void *clip_weights_t, const void *cache_keys_t,
                           const float *cache_values, mmr_dtype dtype, int64_t N, int E, int C, int S, float alpha,
                           float beta, float *tip_logits, float *clip_logits, void *stream);

/* ------------------------------------------------------------------------------------------
 * Encoder towers  (replaces model.encode_image / encode_text of the `clip` package and
 * CLIPModel.get_image_features of transformers; arithmetic per SURVEY.md Appendix A).
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int kind;        /* 0 = CLIP vision, 1 = CLIP text, 2 = BERT-style text classifier (Taiyi) */
    int width;       /* d, multiple of 128 */
    int layers;
    int heads;       /* width / 64 */
    int mlp;         /* multiple of 128 */
    int tokens;      /* vision: 1 + (image_size/patch)^2; text: context length; kind 2: max positions */
    int embed_dim;   /* E, multiple of 128 */
    int image_size;  /* vision only */
    int patch;       /* vision only */
    int vocab;       /* text only */
    float ln_eps;
    int fold_ln;     /* kinds 0/1: 1 = the per-block LayerNorms are folded into the GEMMs that consume them.
                      * The blob then carries, per layer, QKV_W = bf16(W_qkv * ln1.w[None,:]),
                      * QKV_B = W_qkv @ ln1.b + b_qkv, QKV_C[n] = sum_k float(QKV_W[n,k]) and the same for
                      * FC1_{W,B,C} with ln2; LN1_x / LN2_x are not read.  The GEMM epilogue applies
                      * rstd*(acc - mean*C) + B from row statistics of the fp32 residual stream
                      * (same arithmetic as LN-then-GEMM up to bf16 rounding of W*gamma instead of LN(h)).
                      * 0 = separate LayerNorm launches, original tensors.  Must be 0 for kind 2. */
} mmr_tower_cfg;

/* Tensors of the weight blob.  Matrices are bf16 row-major [out,in]; vectors/embeddings fp32
 * except TOK_EMB (bf16 [vocab,d]).  PATCH_W is [d, Kpad] with K = (c,ky,kx) order zero-padded
 * to a multiple of 64.  `layer` is ignored for non-per-layer tensors. */
typedef enum {
    MMR_P_PATCH_W = 0, MMR_P_CLS, MMR_P_POS, MMR_P_LN_PRE_W, MMR_P_LN_PRE_B,
    MMR_P_LN1_W, MMR_P_LN1_B, MMR_P_QKV_W, MMR_P_QKV_B, MMR_P_OUT_W, MMR_P_OUT_B,
    MMR_P_LN2_W, MMR_P_LN2_B, MMR_P_FC1_W, MMR_P_FC1_B, MMR_P_FC2_W, MMR_P_FC2_B,
    MMR_P_LN_FINAL_W, MMR_P_LN_FINAL_B, MMR_P_PROJ, MMR_P_TOK_EMB,
    /* BERT-style text encoder only (kind 2): */
    MMR_P_TYPE_EMB, /* fp32 [2,d] token-type embeddings (row 0 is used) */
    MMR_P_POOL_W,   /* bf16 [d,d] pooler dense */
    MMR_P_POOL_B,   /* fp32 [d] */
    MMR_P_PROJ_B,   /* fp32 [E] classifier bias (MMR_P_PROJ is the classifier weight [E,d]) */
    /* fold_ln towers only, per layer: */
    MMR_P_QKV_C,    /* fp32 [3d] row sums of the gamma-folded QKV weight */
    MMR_P_FC1_C,    /* fp32 [mlp] row sums of the gamma-folded FC1 weight */
    MMR_P_COUNT
} mmr_param;

size_t mmr_tower_weights_bytes(const mmr_tower_cfg *cfg);
/* Byte offset/size of one tensor inside the blob; returns MMR_EINVAL if the tower has no such tensor. */
int mmr_tower_param_span(const mmr_tower_cfg *cfg, int param, int layer, size_t *offset, size_t *bytes);

typedef struct mmr_tower mmr_tower;

/* Workspace status word: every forward call (mmr_tower_forward / mmr_vit_encode_image / mmr_text_encode /
 * mmr_bert_forward*) zeroes the int32 at workspace offset 0 and the embedding kernels OR bits into it:
 *   bit 0 = a token id (or token type id) was out of range and was clamped.
 * The library never reads it back (no sync); a caller that feeds ids produced on the device reads it when it
 * chooses to synchronise.  Ids on the host are range-checked by the host shim before the call. */
#define MMR_STATUS_BAD_TOKEN_ID 1

/* `weights` is a device blob laid out per mmr_tower_param_span; it must outlive the tower.
 * Concurrency: a forward call only READS the tower and its weights; everything it writes lives in the caller's workspace and
 * output.  Forward calls that use DIFFERENT workspaces may therefore be in flight at once on different streams (two batches of
 * a gallery build: measured +10 % images/s on ViT-B/32 at batch 256, the second batch fills the CUs the first one's 150-200-tile
 * GEMM launches leave idle); calls that share a workspace must be stream-ordered.  The same holds for mmr_cosine_topk* and its
 * workspace (searches only read the gallery).  The Python shim calls such a workspace a "lane" (encode_image(lane=),
 * GalleryIndex.search(lane=)).
 * mmr_tower_set_shared_chip(t, 1) tells the tower that its forwards will run beside other work like that: the GEMM launches
 * then pick their tiles for efficiency per FLOP instead of for filling 256 CUs on their own (full 256x256 tiles where the
 * solo policy takes 256x192 ones, one workgroup per tile instead of 256 persistent ones for launches of a few long rounds:
 * +3-5 % images/s with two ViT-B/32 forwards in flight, -5 % with one).  Results do not
 * change (same K order per output element).  Set it while no forward of this tower is being issued. */
int mmr_tower_set_shared_chip(mmr_tower *t, int shared);
int mmr_tower_create(const mmr_tower_cfg *cfg, const void *weights, size_t weights_bytes, mmr_tower **out);
void mmr_tower_destroy(mmr_tower *t);
size_t mmr_tower_workspace_bytes(const mmr_tower *t, int batch);

/* pixels[B,3,S,S] NCHW contiguous (fp32 or bf16) -> out[B,E] (fp32 or bf16).
 * normalize != 0 applies the row L2 normalisation before the store. */
int mmr_vit_encode_image(mmr_tower *t, const void *pixels, mmr_dtype in_dtype, int B, void *out,
                         mmr_dtype out_dtype, int normalize, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ids[N,T] int32 (EOT = largest id per row) -> out[N,E]. */
int mmr_text_encode(mmr_tower *t, const int32_t *ids, int N, void *out, mmr_dtype out_dtype, int normalize,
                    void *workspace, size_t workspace_bytes, void *stream);

/* BERT-style text encoder (kind 2): replaces `text_encoder(text).logits` of
 * BertForSequenceClassification (Taiyi-CLIP Chinese text tower; reference code/test_taiyi.py:12,24,
 * CLIP-Chinese/lab_chinese.py:81-93).  Post-LN blocks, exact GELU, full attention over all T tokens (the
 * reference passes no attention_mask), pooler = tanh(dense(first token)), classifier -> out[N,E].
 * ids[N,T] int32 with 1 <= T <= cfg.tokens.  tap/tap_after as in mmr_tower_forward (after block i). */
size_t mmr_bert_workspace_bytes(const mmr_tower *t, int N, int T);
int mmr_bert_forward(mmr_tower *t, const int32_t *ids, int N, int T, void *out, mmr_dtype out_dtype, int normalize,
                     int tap_after, float *tap, void *workspace, size_t workspace_bytes, void *stream);
/* The `text_encoder(**inputs)` form (reference CLIP/union_dataset.py:312-314, CLIP-Chinese/lab_chinese.py:90-92):
 * the tokenizer's whole output.  token_type_ids[N,T] int32 (nullable = all 0; values 0/1 select the type
 * embedding row); attention_mask[N,T] int32 (nullable = all 1; 0 = padding key: it receives no attention weight,
 * like HF's additive -inf mask; padded QUERY rows are still computed, as in HF, and do not reach the pooled
 * output, which reads token 0). */
int mmr_bert_forward_masked(mmr_tower *t, const int32_t *ids, const int32_t *token_type_ids,
                            const int32_t *attention_mask, int N, int T, void *out, mmr_dtype out_dtype, int normalize,
                            int tap_after, float *tap, void *workspace, size_t workspace_bytes, void *stream);

/* Same forward with a tap on the fp32 residual stream for stage-level parity tests:
 * tap_after = -1 copies h after the embedding (+pre-LN for vision); i >= 0 after block i.
 * tap[B*T, d] fp32 (nullable).  `input` is pixels (vision) or ids (text). */
int mmr_tower_forward(mmr_tower *t, const void *input, mmr_dtype in_dtype, int B, void *out,
                      mmr_dtype out_dtype, int normalize, int tap_after, float *tap, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Preprocess  (replaces the PIL/torchvision `preprocess` applied per image at reference
 * code/search_image.py:127,155: Resize(BICUBIC) -> CenterCrop -> ToTensor -> Normalize).
 * ---------------------------------------------------------------------------------------- */

/* img: uint8 RGB [H,W,3].  The coefficient tables are Pillow's (precompute_coeffs +
 * normalize_coeffs_8bpc) for the S columns / rows of the centre-crop window, built on the host
 * (preprocess.py): bounds int32[S][2] = (first input index, taps), coeffs int32[S][k] in 22-bit
 * fixed point.  row0/rows = the input rows the vertical pass reads; tmp: uint8 [rows,S,3] scratch.
 * out: [3,S,S] fp32 or bf16 = ((u8/255) - mean) / std; out_u8 (nullable): the uint8 [S,S,3] crop. */
int mmr_preprocess_image(const uint8_t *img, int H, int W, int S, int row0, int rows, const int32_t *hbounds,
                         const int32_t *hcoeffs, int hk, const int32_t *vbounds, const int32_t *vcoeffs, int vk,
                         float mean0, float mean1, float mean2, float std0, float std1, float std2, uint8_t *tmp,
                         void *out, mmr_dtype out_dtype, uint8_t *out_u8, void *stream);

/* Batched form: B images of different sizes in one launch pair.  `desc` is a DEVICE array of B
 * descriptors; tables/pointers as in mmr_preprocess_image; out[B,3,S,S].  max_rows = max over images of
 * `rows` (sizes the grid). */
typedef struct {
    const uint8_t *img;
    const int32_t *hbounds, *hcoeffs, *vbounds, *vcoeffs;
    uint8_t *tmp;
    int32_t H, W, row0, rows, hk, vk;
} mmr_preprocess_desc;
int mmr_preprocess_batch(const void *desc, int B, int S, int max_rows, float mean0, float mean1, float mean2,
                         float std0, float std1, float std2, void *out, mmr_dtype out_dtype, void *stream);
/* Same, with max_width = the widest image of the batch (pixels): lets the horizontal pass stage whole input rows in LDS
 * (coalesced loads, ~1 global instruction per output pixel instead of 9).  The fast forms need every image's coefficient
 * rows padded with zero taps to a multiple of 4 (hk % 4 == 0, 16-byte aligned tables: what preprocess.py builds) and
 * S % 4 == 0; an image that does not qualify takes the byte-wise path inside the same launch.  Results are identical.
 * PRECONDITION of the fast forms: |coefficient| < 2^23 (they multiply on the 24-bit integer multiplier) -- true for every
 * table Pillow's normalize_coeffs_8bpc can produce (22 fractional bits, |weight| <= ~1.2); a caller with other tables
 * leaves hk unpadded (hk % 4 != 0) to stay on the 32-bit byte-wise path.  Nothing is read past the last byte of an image. */
int mmr_preprocess_batch_ex(const void *desc, int B, int S, int max_rows, int max_width, float mean0, float mean1, float mean2,
                            float std0, float std1, float std2, void *out, mmr_dtype out_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Launch profiler (measurement aid for bench.py): HIP event pairs around every kernel launch of a
 * class, recorded on the launch stream.  Off by default.
 * ---------------------------------------------------------------------------------------- */
enum {
    MMR_PROF_GEMM = 0,       /* the three GEMM kernels (256-row, 128x128, 128x32), all epilogues */
    MMR_PROF_ATTENTION = 1,  /* attention_kernel / attention_stream_kernel */
    MMR_PROF_ROWWISE = 2,    /* LayerNorm / embedding / pooling / patch gather / output cast */
    MMR_PROF_SCAN = 3,       /* scan_kernel (gallery stream, HBM-bound) */
    MMR_PROF_FINALIZE = 4,   /* select / rescore / rank kernels (selection + exact re-rank) */
    MMR_PROF_EXACT = 5,      /* exhaustive exact kernels */
    MMR_PROF_CLASSES = 6
};
/* on != 0: reset and start recording up to max_launches launches; on == 0: stop. */
int mmr_prof_enable(int on, int max_launches);
/* Sum of event-pair durations (ms) and launch count of one class; synchronises those events. */
int mmr_prof_read(int cls, double *total_ms, long long *launches, long long *dropped);

/* ------------------------------------------------------------------------------------------
 * Kernel-level test hooks (used by tests/ to localise a parity failure; not part of the drop-in).
 * ---------------------------------------------------------------------------------------- */

/* out[M,N] = epilogue(A[M,K] . W[N,K]^T): epi 0 = +bias -> bf16, 1 = +bias, QuickGELU -> bf16,
 * 2 = fp32 out += acc + bias, 3 = plain fp32, 4 = +bias -> fp32, 5 = +bias, exact GELU -> bf16,
 * 6 = +bias, tanh -> bf16.  M,N multiples of 128, K multiple of 64. */
int mmr_debug_gemm(int epi, const void *A, const void *W, int M, int N, int K, const float *bias, void *out,
                   void *stream);
/* x_bf16[rows,d] = LayerNorm(h_f32[rows,d]) */
int mmr_debug_layernorm(const float *h, const float *w, const float *b, void *x, int64_t rows, int d, float eps,
                        void *stream);
/* o_bf16[B*T, d] = softmax(QK^T/8 (+causal)) V per head, from packed qkv_bf16[B*T, 3d] */
int mmr_debug_attention(const void *qkv, void *o, int B, int T, int heads, int causal, void *stream);
/* the non-causal form with a key-padding mask key_mask[B,T] int32 (0 = masked key) */
int mmr_debug_attention_masked(const void *qkv, void *o, int B, int T, int heads, const int32_t *key_mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMR_H */
