// C ABI for the CLIP towers (include/mmr.h): weight-blob layout, handle, workspace plan and the
// forward pass as a fixed sequence of launches on the caller's stream (hipGraph-capturable: no
// allocation, no synchronisation, no host reads of device data).
//
// Replaces model.encode_image / model.encode_text of the `clip` package and
// CLIPModel.get_image_features of transformers, which the reference calls at
// code/search_image.py:132,156,335, code/test_clip.py:12-13, code/test_taiyi.py:23.
#include "mmr_common.h"

#include <new>

namespace mmr {
// gemm.hip
enum { EPI_BIAS_BF16 = 0, EPI_BIAS_GELU_BF16 = 1, EPI_BIAS_RESID_F32 = 2, EPI_STORE_F32 = 3,
       EPI_BIAS_F32 = 4, EPI_BIAS_GELU_ERF_BF16 = 5, EPI_BIAS_TANH_BF16 = 6, EPI_LNFOLD_BF16 = 7, EPI_LNFOLD_GELU_BF16 = 8,
       EPI_RESID_STATS_F32 = 9 };
int launch_gemm(int epi, const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out, hipStream_t st);
int launch_gemm_patch32(const bf16_t *pix, int B, int S, const bf16_t *W, int M, int N, float *out, hipStream_t st);
int launch_gemm_aux(int epi, const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out, const GemmAux &aux, hipStream_t st);
void gemm_set_shared_chip_hint(bool on);
// vit_ops.hip
int launch_im2col(const void *px, mmr_dtype dt, bf16_t *ap, int B, int S, int P, int G, int K, int Kpad, hipStream_t st);
int launch_embed_vision(const float *pe, const float *cls, const float *pos, const float *lw, const float *lb, float *h, int B, int T, int d, float eps, bf16_t *xb, float2 *stats, int32_t *status, hipStream_t st);
int launch_embed_text(const int32_t *ids, const bf16_t *tok, const float *pos, float *h, int Nb, int T, int d, int vocab, bf16_t *xb, float2 *stats, int32_t *status, hipStream_t st);
int launch_layernorm(const float *h, const float *w, const float *b, bf16_t *x, int64_t rows, int d, float eps, hipStream_t st);
int launch_pool_ln(const float *h, const int32_t *ids, const float *w, const float *b, bf16_t *xc, int Nb, int T, int d, float eps, hipStream_t st);
int launch_finish(const float *feat, void *out, mmr_dtype odt, int Nb, int E, int normalize, hipStream_t st);
int launch_embed_bert(const int32_t *ids, const bf16_t *tok, const float *pos, const float *type0, const float *lw, const float *lb, float *h, bf16_t *x, int Nb, int T, int d, int vocab, float eps, const int32_t *types, int32_t *status, hipStream_t st);
int launch_layernorm_inplace(float *h, const float *w, const float *b, bf16_t *x, int64_t rows, int d, float eps, hipStream_t st);
int launch_gather_first_rows(const bf16_t *x, bf16_t *xc, int Nb, int T, int d, hipStream_t st);
int launch_attention(const bf16_t *qkv, bf16_t *o, int Bn, int T, int heads, int d, int causal, const int32_t *kmask, hipStream_t st);

static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

static int patch_k(const mmr_tower_cfg &c) { return 3 * c.patch * c.patch; }
static int patch_kpad(const mmr_tower_cfg &c) { return round_up(patch_k(c), 64); }

static const char *validate_cfg(const mmr_tower_cfg *c)
{
    if (!c) return "null config";
    if (c->kind < 0 || c->kind > 2) return "kind must be 0 (vision), 1 (text) or 2 (BERT text)";
    if (c->width != 128 && c->width != 512 && c->width != 768 && c->width != 1024) return "width must be 128, 512, 768 or 1024";
    if (c->heads * 64 != c->width) return "heads must equal width / 64 (head dim 64)";
    if (c->layers < 1 || c->layers > 64) return "layers outside [1,64]";
    if (c->mlp < 128 || c->mlp % 128) return "mlp must be a positive multiple of 128";
    if (c->embed_dim < 128 || c->embed_dim % 128) return "embed_dim must be a positive multiple of 128";
    if (c->tokens < 1 || c->tokens > 608) return "tokens outside [1,608]";
    if (!(c->ln_eps > 0.f)) return "ln_eps must be > 0";
    if (c->fold_ln != 0 && c->fold_ln != 1) return "fold_ln must be 0 or 1";
    if (c->fold_ln && c->kind == 2) return "fold_ln is for pre-LN (CLIP) towers; BERT towers are post-LN";
    if (c->kind == 0) {
        if (c->patch < 1 || c->image_size < c->patch || c->image_size % c->patch) return "image_size must be a multiple of patch";
        const int g = c->image_size / c->patch;
        if (c->tokens != 1 + g * g) return "vision tokens must equal 1 + (image_size/patch)^2";
    } else {
        if (c->vocab < 2) return "text tower needs vocab >= 2";
    }
    return nullptr;
}

// Blob layout: a flat sequence of 256-byte aligned tensors in the order below.
struct Span { size_t off, bytes; };
struct Layout {
    Span global[MMR_P_COUNT];
    Span layer0[MMR_P_COUNT];
    size_t layer_stride, total;
};

static Layout make_layout(const mmr_tower_cfg &c)
{
    Layout L{};
    size_t off = 0;
    auto put = [&](Span &s, size_t bytes) { s.off = off; s.bytes = bytes; off += align_up(bytes, 256); };
    const size_t d = c.width, m = c.mlp, T = c.tokens, E = c.embed_dim;
    if (c.kind == 0) {
        put(L.global[MMR_P_PATCH_W], d * patch_kpad(c) * 2);
        put(L.global[MMR_P_CLS], d * 4);
        put(L.global[MMR_P_POS], T * d * 4);
        put(L.global[MMR_P_LN_PRE_W], d * 4);
        put(L.global[MMR_P_LN_PRE_B], d * 4);
    } else {
        put(L.global[MMR_P_TOK_EMB], (size_t)c.vocab * d * 2);
        put(L.global[MMR_P_POS], T * d * 4);
    }
    if (c.kind == 2) {
        put(L.global[MMR_P_TYPE_EMB], 2 * d * 4);
        put(L.global[MMR_P_LN_PRE_W], d * 4);      // embeddings LayerNorm
        put(L.global[MMR_P_LN_PRE_B], d * 4);
        put(L.global[MMR_P_POOL_W], d * d * 2);
        put(L.global[MMR_P_POOL_B], d * 4);
        put(L.global[MMR_P_PROJ_B], E * 4);
    } else {
        put(L.global[MMR_P_LN_FINAL_W], d * 4);
        put(L.global[MMR_P_LN_FINAL_B], d * 4);
    }
    put(L.global[MMR_P_PROJ], E * d * 2);
    const size_t l0 = off;
    put(L.layer0[MMR_P_LN1_W], d * 4);
    put(L.layer0[MMR_P_LN1_B], d * 4);
    put(L.layer0[MMR_P_QKV_W], 3 * d * d * 2);
    put(L.layer0[MMR_P_QKV_B], 3 * d * 4);
    put(L.layer0[MMR_P_OUT_W], d * d * 2);
    put(L.layer0[MMR_P_OUT_B], d * 4);
    put(L.layer0[MMR_P_LN2_W], d * 4);
    put(L.layer0[MMR_P_LN2_B], d * 4);
    put(L.layer0[MMR_P_FC1_W], m * d * 2);
    put(L.layer0[MMR_P_FC1_B], m * 4);
    put(L.layer0[MMR_P_FC2_W], d * m * 2);
    put(L.layer0[MMR_P_FC2_B], d * 4);
    if (c.fold_ln) {
        put(L.layer0[MMR_P_QKV_C], 3 * d * 4);
        put(L.layer0[MMR_P_FC1_C], m * 4);
    }
    L.layer_stride = off - l0;
    L.total = l0 + L.layer_stride * c.layers;
    return L;
}

static bool is_layer_param(int p) { return (p >= MMR_P_LN1_W && p <= MMR_P_FC2_B) || p == MMR_P_QKV_C || p == MMR_P_FC1_C; }

}  // namespace mmr

using namespace mmr;

struct mmr_tower {
    mmr_tower_cfg cfg;
    Layout lay;
    const char *w;  // device blob
    bool shared_chip = false;   // mmr_tower_set_shared_chip: forwards run beside other concurrent work (tile policy hint)
    template <typename T> const T *g(int p) const { return reinterpret_cast<const T *>(w + lay.global[p].off); }
    template <typename T> const T *l(int p, int layer) const {
        return reinterpret_cast<const T *>(w + lay.layer0[p].off + (size_t)layer * lay.layer_stride);
    }
};

extern "C" size_t mmr_tower_weights_bytes(const mmr_tower_cfg *cfg)
{
    if (validate_cfg(cfg)) return 0;
    return make_layout(*cfg).total;
}

extern "C" int mmr_tower_param_span(const mmr_tower_cfg *cfg, int param, int layer, size_t *offset, size_t *bytes)
{
    const char *why = validate_cfg(cfg);
    MMR_CHECK_ARG(!why, "mmr_tower_param_span: %s", why);
    MMR_CHECK_ARG(param >= 0 && param < MMR_P_COUNT && offset && bytes, "mmr_tower_param_span: bad param id %d", param);
    const Layout L = make_layout(*cfg);
    if (is_layer_param(param)) {
        MMR_CHECK_ARG(layer >= 0 && layer < cfg->layers, "mmr_tower_param_span: layer %d outside [0,%d)", layer, cfg->layers);
        MMR_CHECK_ARG(L.layer0[param].bytes != 0, "mmr_tower_param_span: this tower has no tensor %d", param);
        *offset = L.layer0[param].off + (size_t)layer * L.layer_stride;
        *bytes = L.layer0[param].bytes;
    } else {
        MMR_CHECK_ARG(L.global[param].bytes != 0, "mmr_tower_param_span: this tower has no tensor %d", param);
        *offset = L.global[param].off;
        *bytes = L.global[param].bytes;
    }
    return MMR_OK;
}

extern "C" int mmr_tower_create(const mmr_tower_cfg *cfg, const void *weights, size_t weights_bytes, mmr_tower **out)
{
    const char *why = validate_cfg(cfg);
    MMR_CHECK_ARG(!why, "mmr_tower_create: %s", why);
    MMR_CHECK_ARG(weights && out, "mmr_tower_create: null pointer");
    MMR_CHECK_ARG(((uintptr_t)weights & 255) == 0, "mmr_tower_create: weight blob must be 256-byte aligned");
    const Layout L = make_layout(*cfg);
    MMR_CHECK_ARG(weights_bytes >= L.total, "mmr_tower_create: blob %zu bytes < layout %zu", weights_bytes, L.total);
    mmr_tower *t = new (std::nothrow) mmr_tower;
    MMR_CHECK_ARG(t != nullptr, "mmr_tower_create: out of host memory");
    t->cfg = *cfg;
    t->lay = L;
    t->w = (const char *)weights;
    *out = t;
    return MMR_OK;
}

extern "C" void mmr_tower_destroy(mmr_tower *t) { delete t; }

namespace {
struct WsPlan {
    int M, Mpad, Mp, Mp_pad, Bpad;
    size_t off_status, off_h, off_x, off_big, off_pe, off_xc, off_feat, off_xo, off_stats, total;
};
WsPlan plan_ws(const mmr_tower_cfg &c, int B)
{
    WsPlan p{};
    p.M = B * c.tokens;
    p.Mpad = round_up(p.M, p.M >= 4096 ? 256 : 128);   // large batches: row count fits the 256x256 GEMM tile
    p.Bpad = round_up(B, 128);
    const size_t d = c.width;
    const size_t wide = (size_t)(c.mlp > 3 * c.width ? c.mlp : 3 * c.width);
    size_t big = (size_t)p.Mpad * wide * 2;
    size_t off = 0;
    auto put = [&](size_t &o, size_t bytes) { o = off; off += align_up(bytes, 256); };
    if (c.kind == 0) {
        p.Mp = B * (c.tokens - 1);
        p.Mp_pad = round_up(p.Mp, p.Mp >= 4096 ? 256 : 128);
        const size_t ap = (size_t)p.Mp_pad * patch_kpad(c) * 2;
        if (ap > big) big = ap;
    }
    put(p.off_status, 256);                                    // status word (include/mmr.h): always at offset 0
    put(p.off_h, (size_t)p.Mpad * d * 4);
    put(p.off_x, (size_t)p.Mpad * d * 2);
    put(p.off_big, big);
    put(p.off_pe, c.kind == 0 ? (size_t)p.Mp_pad * d * 4 : 0);
    put(p.off_xc, (size_t)p.Bpad * d * 2);
    put(p.off_feat, (size_t)p.Bpad * c.embed_dim * 4);
    if (c.fold_ln) {
        put(p.off_xo, (size_t)p.Mpad * d * 2);                 // attention output (x holds bf16(h))
        put(p.off_stats, (size_t)p.Mpad * LNFOLD_NP * 8);      // per-row (sum, sumsq) partial slots
    }
    p.total = off;
    return p;
}
}  // namespace

extern "C" int mmr_tower_set_shared_chip(mmr_tower *t, int shared)
{
    MMR_CHECK_ARG(t != nullptr, "mmr_tower_set_shared_chip: null tower");
    t->shared_chip = shared != 0;
    return MMR_OK;
}

extern "C" size_t mmr_tower_workspace_bytes(const mmr_tower *t, int batch)
{
    if (!t || batch < 1) return 0;
    return plan_ws(t->cfg, batch).total;
}

extern "C" int mmr_tower_forward(mmr_tower *t, const void *input, mmr_dtype in_dtype, int B, void *out, mmr_dtype out_dtype,
                                 int normalize, int tap_after, float *tap, void *workspace, size_t workspace_bytes,
                                 void *stream)
{
    MMR_CHECK_ARG(t != nullptr, "mmr_tower_forward: null tower");
    MMR_CHECK_ARG(t->cfg.kind != 2, "mmr_tower_forward: BERT towers run through mmr_bert_forward");
    const mmr_tower_cfg &c = t->cfg;
    MMR_CHECK_ARG(B >= 0 && B <= 65535, "mmr_tower_forward: batch %d outside [0,65535]", B);
    if (B == 0) return MMR_OK;
    MMR_CHECK_ARG(input && out && workspace, "mmr_tower_forward: null pointer");
    MMR_CHECK_ARG(out_dtype == MMR_F32 || out_dtype == MMR_BF16, "mmr_tower_forward: out dtype %d", (int)out_dtype);
    if (c.kind == 0) MMR_CHECK_ARG(in_dtype == MMR_F32 || in_dtype == MMR_BF16, "mmr_tower_forward: pixel dtype %d", (int)in_dtype);
    MMR_CHECK_ARG(((uintptr_t)input & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "mmr_tower_forward: input must be 16-byte and workspace 256-byte aligned");
    MMR_CHECK_ARG(tap_after >= -1 && tap_after < c.layers, "mmr_tower_forward: tap_after %d outside [-1,%d)", tap_after, c.layers);
    const WsPlan p = plan_ws(c, B);
    if (workspace_bytes < p.total) { set_error("mmr_tower_forward: workspace %zu < required %zu", workspace_bytes, p.total); return MMR_ENOSPC; }

    hipStream_t st = (hipStream_t)stream;
    struct SharedHint {                      // the GEMM launchers' tile policy for the launches of this call (host side, this thread)
        explicit SharedHint(bool on) { gemm_set_shared_chip_hint(on); }
        ~SharedHint() { gemm_set_shared_chip_hint(false); }
    } shared_hint(t->shared_chip);
    char *ws = (char *)workspace;
    float *h = (float *)(ws + p.off_h);
    bf16_t *x = (bf16_t *)(ws + p.off_x);
    bf16_t *big = (bf16_t *)(ws + p.off_big);
    float *pe = (float *)(ws + p.off_pe);
    bf16_t *xc = (bf16_t *)(ws + p.off_xc);
    float *feat = (float *)(ws + p.off_feat);
    const int d = c.width, T = c.tokens, E = c.embed_dim, m = c.mlp;
    const size_t hbytes = (size_t)p.M * d * sizeof(float);
    int rc;
    const bool fold = c.fold_ln != 0;
    bf16_t *xo = fold ? (bf16_t *)(ws + p.off_xo) : x;
    float2 *stats = fold ? (float2 *)(ws + p.off_stats) : nullptr;
    bf16_t *xb = fold ? x : nullptr;
    int32_t *status = (int32_t *)(ws + p.off_status);
    // text: the embedding kernel ORs bits into the word, so it is cleared ahead of it; vision: cleared BY the embedding kernel
    if (c.kind != 0) MMR_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));

    // ---- embeddings
    if (c.kind == 0) {
        const int G = c.image_size / c.patch, K = patch_k(c), Kp = patch_kpad(c);
        // patch 32 + bf16 pixels: the patch gather rides in the GEMM's A-tile loads (no im2col pass, no patch matrix)
        rc = MMR_ENOTSUP;
        if (c.patch == 32 && in_dtype == MMR_BF16 && K == Kp)
            rc = launch_gemm_patch32((const bf16_t *)input, B, c.image_size, t->g<bf16_t>(MMR_P_PATCH_W), p.Mp_pad, d, pe, st);
        if (rc == MMR_ENOTSUP) {
            if ((rc = launch_im2col(input, in_dtype, big, B, c.image_size, c.patch, G, K, Kp, st))) return rc;
            rc = launch_gemm(EPI_STORE_F32, big, t->g<bf16_t>(MMR_P_PATCH_W), p.Mp_pad, d, Kp, nullptr, pe, st);
        }
        if (rc) return rc;
        if ((rc = launch_embed_vision(pe, t->g<float>(MMR_P_CLS), t->g<float>(MMR_P_POS), t->g<float>(MMR_P_LN_PRE_W),
                                      t->g<float>(MMR_P_LN_PRE_B), h, B, T, d, c.ln_eps, xb, stats, status, st))) return rc;
    } else {
        if ((rc = launch_embed_text((const int32_t *)input, t->g<bf16_t>(MMR_P_TOK_EMB), t->g<float>(MMR_P_POS), h, B, T, d,
                                    c.vocab, xb, stats, status, st))) return rc;
    }
    if (tap && tap_after == -1) MMR_CHECK_HIP(hipMemcpyAsync(tap, h, hbytes, hipMemcpyDeviceToDevice, st));

    // ---- transformer blocks (pre-LN; modeling_clip.py:362-383)
    for (int i = 0; i < c.layers && fold; ++i) {
        // x = bf16(h) and `stats` always describe the current residual rows; LN1/LN2 live in the QKV/FC1 epilogues
        GemmAux a{};
        a.stats_in = stats; a.stats_out = stats; a.xout = x; a.inv_d = 1.f / (float)d; a.eps = c.ln_eps;
        a.colsum = t->l<float>(MMR_P_QKV_C, i);
        if ((rc = launch_gemm_aux(EPI_LNFOLD_BF16, x, t->l<bf16_t>(MMR_P_QKV_W, i), p.Mpad, 3 * d, d, t->l<float>(MMR_P_QKV_B, i), big, a, st))) return rc;
        if ((rc = launch_attention(big, xo, B, T, c.heads, d, c.kind == 1, nullptr, st))) return rc;
        if ((rc = launch_gemm_aux(EPI_RESID_STATS_F32, xo, t->l<bf16_t>(MMR_P_OUT_W, i), p.Mpad, d, d, t->l<float>(MMR_P_OUT_B, i), h, a, st))) return rc;
        a.colsum = t->l<float>(MMR_P_FC1_C, i);
        if ((rc = launch_gemm_aux(EPI_LNFOLD_GELU_BF16, x, t->l<bf16_t>(MMR_P_FC1_W, i), p.Mpad, m, d, t->l<float>(MMR_P_FC1_B, i), big, a, st))) return rc;
        if ((rc = launch_gemm_aux(EPI_RESID_STATS_F32, big, t->l<bf16_t>(MMR_P_FC2_W, i), p.Mpad, d, m, t->l<float>(MMR_P_FC2_B, i), h, a, st))) return rc;
        if (tap && tap_after == i) MMR_CHECK_HIP(hipMemcpyAsync(tap, h, hbytes, hipMemcpyDeviceToDevice, st));
    }
    for (int i = 0; i < c.layers && !fold; ++i) {
        if ((rc = launch_layernorm(h, t->l<float>(MMR_P_LN1_W, i), t->l<float>(MMR_P_LN1_B, i), x, p.M, d, c.ln_eps, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_BF16, x, t->l<bf16_t>(MMR_P_QKV_W, i), p.Mpad, 3 * d, d, t->l<float>(MMR_P_QKV_B, i), big, st))) return rc;
        if ((rc = launch_attention(big, x, B, T, c.heads, d, c.kind == 1, nullptr, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_RESID_F32, x, t->l<bf16_t>(MMR_P_OUT_W, i), p.Mpad, d, d, t->l<float>(MMR_P_OUT_B, i), h, st))) return rc;
        if ((rc = launch_layernorm(h, t->l<float>(MMR_P_LN2_W, i), t->l<float>(MMR_P_LN2_B, i), x, p.M, d, c.ln_eps, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_GELU_BF16, x, t->l<bf16_t>(MMR_P_FC1_W, i), p.Mpad, m, d, t->l<float>(MMR_P_FC1_B, i), big, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_RESID_F32, big, t->l<bf16_t>(MMR_P_FC2_W, i), p.Mpad, d, m, t->l<float>(MMR_P_FC2_B, i), h, st))) return rc;
        if (tap && tap_after == i) MMR_CHECK_HIP(hipMemcpyAsync(tap, h, hbytes, hipMemcpyDeviceToDevice, st));
    }

    // ---- pooled row -> LayerNorm -> projection (no bias) -> optional L2 normalise -> cast
    if ((rc = launch_pool_ln(h, c.kind == 1 ? (const int32_t *)input : nullptr, t->g<float>(MMR_P_LN_FINAL_W),
                             t->g<float>(MMR_P_LN_FINAL_B), xc, B, T, d, c.ln_eps, st))) return rc;
    if ((rc = launch_gemm(EPI_STORE_F32, xc, t->g<bf16_t>(MMR_P_PROJ), p.Bpad, E, d, nullptr, feat, st))) return rc;
    return launch_finish(feat, out, out_dtype, B, E, normalize, st);
}

extern "C" int mmr_vit_encode_image(mmr_tower *t, const void *pixels, mmr_dtype in_dtype, int B, void *out,
                                    mmr_dtype out_dtype, int normalize, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    MMR_CHECK_ARG(t && t->cfg.kind == 0, "mmr_vit_encode_image: not a vision tower");
    return mmr_tower_forward(t, pixels, in_dtype, B, out, out_dtype, normalize, -1, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mmr_text_encode(mmr_tower *t, const int32_t *ids, int N, void *out, mmr_dtype out_dtype, int normalize,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    MMR_CHECK_ARG(t && t->cfg.kind == 1, "mmr_text_encode: not a text tower");
    return mmr_tower_forward(t, ids, MMR_F32, N, out, out_dtype, normalize, -1, nullptr, workspace, workspace_bytes, stream);
}

// ---- BERT-style text classifier (Taiyi)
namespace {
struct BertWs {
    int M, Mpad, Npad;
    size_t off_status, off_h, off_x, off_big, off_xc, off_xp, off_feat, total;
};
BertWs plan_bert(const mmr_tower_cfg &c, int N, int T)
{
    BertWs p{};
    p.M = N * T;
    p.Mpad = round_up(p.M, p.M >= 4096 ? 256 : 128);
    p.Npad = round_up(N, 128);
    const size_t d = c.width;
    const size_t wide = (size_t)(c.mlp > 3 * c.width ? c.mlp : 3 * c.width);
    size_t off = 0;
    auto put = [&](size_t &o, size_t bytes) { o = off; off += align_up(bytes, 256); };
    put(p.off_status, 256);
    put(p.off_h, (size_t)p.Mpad * d * 4);
    put(p.off_x, (size_t)p.Mpad * d * 2);
    put(p.off_big, (size_t)p.Mpad * wide * 2);
    put(p.off_xc, (size_t)p.Npad * d * 2);
    put(p.off_xp, (size_t)p.Npad * d * 2);
    put(p.off_feat, (size_t)p.Npad * c.embed_dim * 4);
    p.total = off;
    return p;
}
}  // namespace

extern "C" size_t mmr_bert_workspace_bytes(const mmr_tower *t, int N, int T)
{
    if (!t || t->cfg.kind != 2 || N < 1 || T < 1 || T > t->cfg.tokens) return 0;
    return plan_bert(t->cfg, N, T).total;
}

extern "C" int mmr_bert_forward_masked(mmr_tower *t, const int32_t *ids, const int32_t *token_type_ids,
                                       const int32_t *attention_mask, int N, int T, void *out, mmr_dtype out_dtype,
                                       int normalize, int tap_after, float *tap, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    MMR_CHECK_ARG(t && t->cfg.kind == 2, "mmr_bert_forward: not a BERT tower");
    const mmr_tower_cfg &c = t->cfg;
    MMR_CHECK_ARG(N >= 0 && N <= 65535, "mmr_bert_forward: batch %d outside [0,65535]", N);
    if (N == 0) return MMR_OK;
    MMR_CHECK_ARG(T >= 1 && T <= c.tokens, "mmr_bert_forward: sequence length %d outside [1,%d]", T, c.tokens);
    MMR_CHECK_ARG(ids && out && workspace, "mmr_bert_forward: null pointer");
    MMR_CHECK_ARG(out_dtype == MMR_F32 || out_dtype == MMR_BF16, "mmr_bert_forward: out dtype %d", (int)out_dtype);
    MMR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "mmr_bert_forward: workspace must be 256-byte aligned");
    MMR_CHECK_ARG(tap_after >= -1 && tap_after < c.layers, "mmr_bert_forward: tap_after %d outside [-1,%d)", tap_after, c.layers);
    const BertWs p = plan_bert(c, N, T);
    if (workspace_bytes < p.total) { set_error("mmr_bert_forward: workspace %zu < required %zu", workspace_bytes, p.total); return MMR_ENOSPC; }

    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *h = (float *)(ws + p.off_h);
    bf16_t *x = (bf16_t *)(ws + p.off_x);
    bf16_t *big = (bf16_t *)(ws + p.off_big);
    bf16_t *xc = (bf16_t *)(ws + p.off_xc);
    bf16_t *xp = (bf16_t *)(ws + p.off_xp);
    float *feat = (float *)(ws + p.off_feat);
    const int d = c.width, E = c.embed_dim, m = c.mlp;
    const size_t hbytes = (size_t)p.M * d * sizeof(float);
    int rc;

    int32_t *status = (int32_t *)(ws + p.off_status);
    MMR_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if ((rc = launch_embed_bert(ids, t->g<bf16_t>(MMR_P_TOK_EMB), t->g<float>(MMR_P_POS), t->g<float>(MMR_P_TYPE_EMB),
                                t->g<float>(MMR_P_LN_PRE_W), t->g<float>(MMR_P_LN_PRE_B), h, x, N, T, d, c.vocab, c.ln_eps,
                                token_type_ids, status, st))) return rc;
    if (tap && tap_after == -1) MMR_CHECK_HIP(hipMemcpyAsync(tap, h, hbytes, hipMemcpyDeviceToDevice, st));
    // post-LN blocks (modeling_bert.py:139-204,282-352): x = LN(x + Attn(x)); x = LN(x + FFN(x))
    for (int i = 0; i < c.layers; ++i) {
        if ((rc = launch_gemm(EPI_BIAS_BF16, x, t->l<bf16_t>(MMR_P_QKV_W, i), p.Mpad, 3 * d, d, t->l<float>(MMR_P_QKV_B, i), big, st))) return rc;
        if ((rc = launch_attention(big, x, N, T, c.heads, d, 0, attention_mask, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_RESID_F32, x, t->l<bf16_t>(MMR_P_OUT_W, i), p.Mpad, d, d, t->l<float>(MMR_P_OUT_B, i), h, st))) return rc;
        if ((rc = launch_layernorm_inplace(h, t->l<float>(MMR_P_LN1_W, i), t->l<float>(MMR_P_LN1_B, i), x, p.M, d, c.ln_eps, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_GELU_ERF_BF16, x, t->l<bf16_t>(MMR_P_FC1_W, i), p.Mpad, m, d, t->l<float>(MMR_P_FC1_B, i), big, st))) return rc;
        if ((rc = launch_gemm(EPI_BIAS_RESID_F32, big, t->l<bf16_t>(MMR_P_FC2_W, i), p.Mpad, d, m, t->l<float>(MMR_P_FC2_B, i), h, st))) return rc;
        if ((rc = launch_layernorm_inplace(h, t->l<float>(MMR_P_LN2_W, i), t->l<float>(MMR_P_LN2_B, i), x, p.M, d, c.ln_eps, st))) return rc;
        if (tap && tap_after == i) MMR_CHECK_HIP(hipMemcpyAsync(tap, h, hbytes, hipMemcpyDeviceToDevice, st));
    }
    // pooler tanh(dense(first token)) -> classifier (modeling_bert.py:451-464; BertForSequenceClassification)
    if ((rc = launch_gather_first_rows(x, xc, N, T, d, st))) return rc;
    if ((rc = launch_gemm(EPI_BIAS_TANH_BF16, xc, t->g<bf16_t>(MMR_P_POOL_W), p.Npad, d, d, t->g<float>(MMR_P_POOL_B), xp, st))) return rc;
    if ((rc = launch_gemm(EPI_BIAS_F32, xp, t->g<bf16_t>(MMR_P_PROJ), p.Npad, E, d, t->g<float>(MMR_P_PROJ_B), feat, st))) return rc;
    return launch_finish(feat, out, out_dtype, N, E, normalize, st);
}

extern "C" int mmr_bert_forward(mmr_tower *t, const int32_t *ids, int N, int T, void *out, mmr_dtype out_dtype,
                                int normalize, int tap_after, float *tap, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    return mmr_bert_forward_masked(t, ids, nullptr, nullptr, N, T, out, out_dtype, normalize, tap_after, tap, workspace,
                                   workspace_bytes, stream);
}

// ---- kernel-level test hooks
extern "C" int mmr_debug_gemm(int epi, const void *A, const void *W, int M, int N, int K, const float *bias, void *out,
                              void *stream)
{
    MMR_CHECK_ARG(epi >= 0 && epi <= 6 && A && W && out && (bias || epi == 3), "mmr_debug_gemm: bad argument");
    return launch_gemm(epi, (const bf16_t *)A, (const bf16_t *)W, M, N, K, bias, out, (hipStream_t)stream);
}

extern "C" int mmr_debug_layernorm(const float *h, const float *w, const float *b, void *x, int64_t rows, int d, float eps,
                                   void *stream)
{
    MMR_CHECK_ARG(h && w && b && x && rows >= 0, "mmr_debug_layernorm: bad argument");
    if (rows == 0) return MMR_OK;
    return launch_layernorm(h, w, b, (bf16_t *)x, rows, d, eps, (hipStream_t)stream);
}

extern "C" int mmr_debug_attention(const void *qkv, void *o, int B, int T, int heads, int causal, void *stream)
{
    MMR_CHECK_ARG(qkv && o && B >= 1 && B <= 65535 && T >= 1 && heads >= 1, "mmr_debug_attention: bad argument");
    return launch_attention((const bf16_t *)qkv, (bf16_t *)o, B, T, heads, heads * 64, causal, nullptr, (hipStream_t)stream);
}

extern "C" int mmr_debug_attention_masked(const void *qkv, void *o, int B, int T, int heads, const int32_t *key_mask, void *stream)
{
    MMR_CHECK_ARG(qkv && o && key_mask && B >= 1 && B <= 65535 && T >= 1 && heads >= 1, "mmr_debug_attention_masked: bad argument");
    return launch_attention((const bf16_t *)qkv, (bf16_t *)o, B, T, heads, heads * 64, 0, key_mask, (hipStream_t)stream);
}
